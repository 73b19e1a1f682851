"""ttscube_amd — MI355X (gfx950) native waveform-synthesis hot path of TTS-Cube.

Mirrors the reference's Python surface for this path (cube.api, cube.networks.*, hifigan.models);
all compute runs in hand-written HIP kernels behind the C ABI of ``libttscube_hip.so``
(``include/ttscube_hip.h``).  There is NO CPU fallback: if the extension is missing or no GPU is
visible, the ops raise.
"""
__version__ = '0.1.0'
