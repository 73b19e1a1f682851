"""CPU: the C-ABI library loads and exports every symbol include/ttscube_hip.h declares (no compute calls)."""
import os
import re

import pytest

from tests.conftest import ROOT


def _declared_symbols():
    txt = open(os.path.join(ROOT, 'include', 'ttscube_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(ttsc_[a-z0-9_]+)\s*\(', txt)))


def test_header_declares_symbols():
    syms = _declared_symbols()
    assert 'ttsc_hifigan_forward' in syms and 'ttsc_conv1d_forward' in syms


def test_library_exports_every_declared_symbol():
    from ttscube_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    L = _lib.lib()
    for s in _declared_symbols():
        assert hasattr(L, s), 'libttscube_hip.so does not export %s' % s
    # the ctypes table binds exactly the header's symbols
    assert sorted(_lib.SIGNATURES) == _declared_symbols()
    assert b'gfx950' in L.ttsc_version()


def test_no_cpu_fallback():
    import torch
    from ttscube_amd import _lib
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(_lib.TTSCError):
        _lib.require_gpu()
