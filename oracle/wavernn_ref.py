"""ORACLE wrapper (test infrastructure): ctypes binding of oracle/wavernn_ref.c + numpy restatements of the
µ-law codec (cube/networks/loss.py:236-269) and of CubenetVocoder's chunk fold/unfold (vocoder.py:109-131).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'libwavernn_ref.so')
MODE_ARGMAX, MODE_NOISE, MODE_PHILOX = 0, 1, 2
OUT_MULAW, OUT_RAW, OUT_MOL, OUT_GM, OUT_BETA = 0, 1, 2, 3, 4
OUT_KIND = {'mulaw': OUT_MULAW, 'raw': OUT_RAW, 'mol': OUT_MOL, 'gm': OUT_GM, 'beta': OUT_BETA}
SAMPLE_SIZE = {'mulaw': 256, 'raw': 256, 'mol': 30, 'gm': 2, 'beta': 2}      # loss.py sample_size
NOISE_WIDTH = {'mulaw': 256, 'raw': 256, 'mol': 11, 'gm': 1, 'beta': 18}     # injected noise scalars per step (wavernn_ref.c)


class Cfg(C.Structure):
    _fields_ = [('H', C.c_int32), ('num_layers', C.c_int32), ('use_lowres', C.c_int32), ('upsample', C.c_int32),
                ('upsample_low', C.c_int32), ('S', C.c_int32), ('n_mel', C.c_int32), ('out_kind', C.c_int32)]


class Weights(C.Structure):
    _fields_ = [('w_ih', C.c_void_p * 4), ('w_hh', C.c_void_p * 4), ('b_ih', C.c_void_p * 4), ('b_hh', C.c_void_p * 4),
                ('w_pre', C.c_void_p), ('b_pre', C.c_void_p), ('w_out', C.c_void_p), ('b_out', C.c_void_p),
                ('lc_w', C.c_void_p * 3), ('lc_b', C.c_void_p * 3), ('lut', C.c_void_p)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, 'wavernn_ref.c')
        if (not os.path.exists(_SO)) or os.path.getmtime(_SO) < os.path.getmtime(src):
            subprocess.run(['make', '-C', _HERE], check=True, capture_output=True)
        _lib = C.CDLL(_SO)
        _lib.wr_out_len.restype = C.c_int64
        for f in ('wr_expf', 'wr_logf', 'wr_tanhf', 'wr_sigmoidf'):
            getattr(_lib, f).restype = C.c_float
            getattr(_lib, f).argtypes = [C.c_float]
        _lib.wr_gumbel.restype = C.c_float
        _lib.wr_gumbel.argtypes = [C.c_uint32]
        _lib.wr_normal_icdf.restype = C.c_float
        _lib.wr_normal_icdf.argtypes = [C.c_float]
        _lib.wr_sample_beta.restype = C.c_float
        _lib.wr_sample_beta.argtypes = [C.c_void_p, C.c_void_p]
        _lib.wr_sample_mol.restype = C.c_float
        _lib.wr_sample_mol.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.wr_noise_beta.restype = None
        _lib.wr_noise_beta.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p]
    return _lib


def mulaw_lut(golden_dir=None):
    """256-entry decode table captured from the reference (tests/golden/mulaw_lut.npy)."""
    gd = golden_dir or os.path.join(os.path.dirname(_HERE), 'tests', 'golden')
    return np.load(os.path.join(gd, 'mulaw_lut.npy')).astype(np.float32)


def mulaw_encode(x):
    """loss.py:236-255 (torch branch, fp32): sign(x) log1p(255|x|)/log1p(255) -> ((.+1)/2*255+0.5) truncated, clipped."""
    x = np.asarray(x, dtype=np.float32)
    mu = np.float32(255.0)
    x_mu = np.sign(x) * np.log1p(mu * np.abs(x)) / np.log1p(mu)
    x_mu = ((x_mu + np.float32(1)) / np.float32(2) * mu + np.float32(0.5)).astype(np.int64)
    return np.clip(x_mu, 0, 255)


def raw_encode(x):
    x = np.asarray(x, dtype=np.float32)
    return np.clip(((x + np.float32(1.0)) / np.float32(2)) * np.float32(255), 0, 255).astype(np.int64)


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def decode(sd, mel, x_low=None, num_layers=1, H=512, use_lowres=True, upsample=240, upsample_low=10, output='mulaw',
           mode=MODE_ARGMAX, noise=None, seed=0, forced_x=None, want_logits=False, prefix='', b_offset=0):
    """sd: state_dict-like {name: array} in the reference WaveRNN key layout."""
    L_ = lib()
    mel = _f32(mel)
    B, T, NM = mel.shape
    S = SAMPLE_SIZE[output]
    NW = NOISE_WIDTH[output]
    cfg = Cfg(H, num_layers, int(use_lowres), upsample, upsample_low, S, NM, OUT_KIND[output])
    keep = []

    def P(name):
        a = _f32(sd[prefix + name])
        keep.append(a)
        return a.ctypes.data_as(C.c_void_p)

    w = Weights()
    for l in range(num_layers):
        w.w_ih[l] = P('_rnns.%d.weight_ih_l0' % l)
        w.w_hh[l] = P('_rnns.%d.weight_hh_l0' % l)
        w.b_ih[l] = P('_rnns.%d.bias_ih_l0' % l)
        w.b_hh[l] = P('_rnns.%d.bias_hh_l0' % l)
    w.w_pre, w.b_pre = P('_preoutput.linear_layer.weight'), P('_preoutput.linear_layer.bias')
    w.w_out, w.b_out = P('_output.linear_layer.weight'), P('_output.linear_layer.bias')
    Tl = 0
    xl = None
    if use_lowres:
        xl = _f32(x_low)
        Tl = xl.shape[1]
        for i in range(3):
            w.lc_w[i] = P('_lowres_conv.%d.conv.weight' % i)
            w.lc_b[i] = P('_lowres_conv.%d.conv.bias' % i)
    lut = mulaw_lut()
    w.lut = lut.ctypes.data_as(C.c_void_p)
    L = int(L_.wr_out_len(C.byref(cfg), T, Tl))
    idx = np.zeros((B, L), dtype=np.uint8)
    wav = np.zeros((B, L), dtype=np.float32)
    logits = np.zeros((B, L, S), dtype=np.float32) if want_logits else None
    nz = _f32(noise) if noise is not None else None
    fx = _f32(forced_x) if forced_x is not None else None
    if nz is not None:
        assert nz.shape == (B, L, NW), (nz.shape, (B, L, NW))
    if fx is not None:
        assert fx.shape[0] == B and fx.shape[1] >= L
        fx = _f32(fx[:, :L])
    rc = L_.wr_decode_at(C.byref(cfg), C.byref(w), mel.ctypes.data_as(C.c_void_p),
                      xl.ctypes.data_as(C.c_void_p) if xl is not None else None, B, T, Tl, mode,
                      nz.ctypes.data_as(C.c_void_p) if nz is not None else None, C.c_uint64(seed),
                      fx.ctypes.data_as(C.c_void_p) if fx is not None else None, C.c_int(b_offset),
                      idx.ctypes.data_as(C.c_void_p), wav.ctypes.data_as(C.c_void_p),
                      logits.ctypes.data_as(C.c_void_p) if logits is not None else None)
    assert rc == 0
    return idx, wav, logits


# ---- CubenetVocoder chunk folding (vocoder.py:109-131), numpy restatement -----------------------------
def inference_batch(mel, x_low, num_batches=20, upsample_low=10):
    """mel [1,T,80], x_low [1,Tl] -> folded {'mel': [nb, T/nb+1, 80], 'x_low': [nb, Tl/nb+10]}"""
    mel = np.asarray(mel, dtype=np.float32)
    x_low = np.asarray(x_low, dtype=np.float32)
    if mel.shape[1] < num_batches:
        num_batches = mel.shape[1]
    mel = mel[:, :mel.shape[1] // num_batches * num_batches]
    x_low = x_low[:, :x_low.shape[1] // num_batches * num_batches]
    mel_split = mel.reshape(num_batches, -1, mel.shape[2])
    x_low_split = x_low.reshape(num_batches, -1)
    m = np.ones((mel_split.shape[0], mel_split.shape[1] + 1, mel_split.shape[2]), dtype=np.float32) * -5
    m[:, 1:, :] = mel_split
    m[1:, 0, :] = mel_split[:-1, -1, :]
    xl = np.zeros((x_low_split.shape[0], x_low_split.shape[1] + upsample_low), dtype=np.float32)
    xl[:, upsample_low:] = x_low_split
    xl[1:, 0:upsample_low] = x_low_split[:-1, -upsample_low:]
    return {'mel': m, 'x_low': xl}


def compose_batched_inference(batched_x, upsample=240):
    batched_x = batched_x[:, upsample:]
    return batched_x.reshape(1, -1)


def synthetic_state_dict(H=512, num_layers=1, use_lowres=True, seed=1234, S=256, n_mel=80, prefix=''):
    """Seeded synthetic weights in the reference WaveRNN state_dict layout, torch-free so the
    same tensors can be rebuilt on the GPU box.  Includes the dead `_skip` Linear the reference always saves."""
    rng = np.random.RandomState(seed)
    ic = n_mel + 1 + (21 if use_lowres else 0)
    sd = {}

    def U(shape, bound):
        return rng.uniform(-bound, bound, size=shape).astype(np.float32)

    if use_lowres:
        cin = 1
        for i in range(3):
            sd['_lowres_conv.%d.conv.weight' % i] = U((20, cin, 7), 1.5 / np.sqrt(cin * 7))
            sd['_lowres_conv.%d.conv.bias' % i] = U((20,), 0.2)
            cin = 20
    sd['_skip.linear_layer.weight'] = U((H, ic), 1.0 / np.sqrt(ic))
    sd['_skip.linear_layer.bias'] = U((H,), 0.1)
    inp = ic
    for l in range(num_layers):
        k = 1.0 / np.sqrt(H)
        sd['_rnns.%d.weight_ih_l0' % l] = U((3 * H, inp), 1.5 * k)
        sd['_rnns.%d.weight_hh_l0' % l] = U((3 * H, H), 1.5 * k)
        sd['_rnns.%d.bias_ih_l0' % l] = U((3 * H,), k)
        sd['_rnns.%d.bias_hh_l0' % l] = U((3 * H,), k)
        inp = H
    sd['_preoutput.linear_layer.weight'] = U((256, H), 2.0 / np.sqrt(H))
    sd['_preoutput.linear_layer.bias'] = U((256,), 0.1)
    sd['_output.linear_layer.weight'] = U((S, 256), 4.0 / np.sqrt(256))
    sd['_output.linear_layer.bias'] = U((S,), 0.5)
    return {prefix + k: v for k, v in sd.items()}


def synthetic_inputs(B, T, seed=1234, upsample=240, upsample_low=10, n_mel=80):
    """mel = clip(N(-2,1),-5,1) [B,T,80]; x_low ~ U(-1,1) [B, T*upsample/upsample_low]  (SURVEY.md §8d C3)."""
    rng = np.random.RandomState(seed)
    mel = np.clip(rng.randn(B, T, n_mel) - 2.0, -5.0, 1.0).astype(np.float32)
    x_low = rng.uniform(-1, 1, size=(B, T * upsample // upsample_low)).astype(np.float32)
    return mel, x_low
