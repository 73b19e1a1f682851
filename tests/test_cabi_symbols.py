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


def _declared_prototypes():
    """{symbol: (return type text, [parameter type texts])} parsed from the header's prototypes"""
    txt = open(os.path.join(ROOT, 'include', 'ttscube_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    txt = re.sub(r'//[^\n]*', '', txt)
    out = {}
    for m in re.finditer(r'([A-Za-z_][A-Za-z0-9_ \*]*?)\b(ttsc_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;', txt, flags=re.S):
        ret, name, params = m.group(1).strip(), m.group(2), ' '.join(m.group(3).split())
        plist = [] if params in ('', 'void') else [p.strip() for p in params.split(',')]
        out[name] = (ret, plist)
    return out


def _kind(ctype_text):
    """coarse class of a C parameter / return type: pointer, float, or integer of a byte width"""
    t = ctype_text.replace('const ', '').strip()
    if '*' in t:
        return 'ptr'
    base = t.split()[0] if t.split() else t
    return {'float': 'f32', 'double': 'f64', 'int': 'i32', 'int32_t': 'i32', 'uint32_t': 'i32', 'int64_t': 'i64', 'uint64_t': 'i64', 'size_t': 'i64',
            'void': 'void'}.get(base, base)


def test_ctypes_signatures_match_the_header_prototypes():
    """every entry of the ctypes table has the header's parameter COUNT and the same pointer / float / integer-width pattern (a stale argtypes list
    passes garbage in registers without any error from ctypes)"""
    import ctypes as C
    from ttscube_amd import _lib
    kinds = {C.c_void_p: 'ptr', C.c_char_p: 'ptr', C.c_float: 'f32', C.c_double: 'f64', C.c_int: 'i32', C.c_int32: 'i32', C.c_uint32: 'i32',
             C.c_int64: 'i64', C.c_uint64: 'i64', C.c_size_t: 'i64', None: 'void'}
    protos = _declared_prototypes()
    assert sorted(protos) == _declared_symbols()
    bad = []
    for name, (res, args) in _lib.SIGNATURES.items():
        ret, plist = protos[name]
        want = [_kind(p) for p in plist]
        got = ['ptr' if (isinstance(a, type) and issubclass(a, (C._Pointer,))) else kinds.get(a, str(a)) for a in args]
        if len(want) != len(got) or any(w != g for w, g in zip(want, got)):
            bad.append((name, want, got))
        if kinds.get(res, 'ptr' if res is not None else 'void') != _kind(ret):
            bad.append((name, 'return', _kind(ret), kinds.get(res, str(res))))
    assert not bad, bad[:5]


def test_no_cpu_fallback():
    import torch
    from ttscube_amd import _lib
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(_lib.TTSCError):
        _lib.require_gpu()
