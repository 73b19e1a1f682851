"""Mirror of the reference's ``cube.networks`` for the waveform-synthesis hot path (same class names, constructor
arguments, batch-dict keys and ``state_dict`` layouts); compute runs in libttscube_hip.so."""
