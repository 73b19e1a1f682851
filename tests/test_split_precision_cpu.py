"""CPU: the ARITHMETIC of the split-precision training kernels (csrc/conv_kernels.hpp `pow2_to`, csrc/conv_train.hip, conv_wgrad.hip), restated in
numpy — fp32 values carried as fp16 hi + lo after a power-of-two scale derived from the tensor's maximum, three products (hi*hi + hi*lo + lo*hi),
fp32 accumulation — against float64.  It pins the claims DESIGN §7 makes about the scheme (not about the kernels: those are compared with float64 on
the GPU in tests/test_conv_train_gpu.py): ~1e-6 of the output's magnitude for any input magnitude, graceful degradation for elements far below
the tensor's maximum, zero / non-finite maxima leave the scale at 1."""
import numpy as np
import pytest

X_TARGET, W_TARGET = 15, 10


def pow2_to(m, target):
    """the kernels' rule: m = f * 2^e (f in [0.5, 1)) -> 2^(target - e); 1 for zero, subnormal and non-finite m"""
    m = np.float32(m)
    ex = (m.view(np.uint32) >> 23) & 0xff
    if ex == 0 or ex == 255:
        return np.float32(1.0)
    s = int(np.clip(target - (int(ex) - 126), -100, 100))
    return np.float32(2.0) ** s


def split(v):
    hi = v.astype(np.float16)
    lo = (v - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float32), lo.astype(np.float32)


def split_matmul(w, x):
    """y = w @ x the way the kernels evaluate it (per-tensor scales, three fp16 x fp16 products — exact in fp32 —, fp32 accumulation)"""
    sx, sw = pow2_to(np.abs(x).max(), X_TARGET), pow2_to(np.abs(w).max(), W_TARGET)
    xh, xl = split(x * sx)
    wh, wl = split(w * sw)
    acc = (wl @ xh + wh @ xl + wh @ xh).astype(np.float32)
    return acc * (np.float32(1.0) / (sx * sw))


@pytest.mark.parametrize('mag', [1e-8, 1e-6, 1.0, 1e4, 3e7])
def test_scale_rule_keeps_any_magnitude_in_range(mag):
    rng = np.random.default_rng(0)
    w = (rng.standard_normal((64, 320)) / 18).astype(np.float32)
    x = (rng.standard_normal((320, 200)) * mag).astype(np.float32)
    ref = w.astype(np.float64) @ x.astype(np.float64)
    y = split_matmul(w, x)
    assert np.isfinite(y).all()
    assert np.abs(y - ref).max() <= 2e-6 * np.abs(ref).max()
    # the scaled maximum sits in [2^14, 2^15) / [2^9, 2^10): inside fp16 with the low halves of typical elements still normal
    assert 2.0 ** 14 <= np.abs(x).max() * pow2_to(np.abs(x).max(), X_TARGET) < 2.0 ** 15
    assert 2.0 ** 9 <= np.abs(w).max() * pow2_to(np.abs(w).max(), W_TARGET) < 2.0 ** 10


def test_elements_far_below_the_maximum_degrade_gracefully():
    """fp16 subnormals are kept: the representation error is 2^-25 in scaled units = 2^-40 of the tensor's maximum, whatever the element"""
    rng = np.random.default_rng(1)
    w = (rng.standard_normal((32, 192)) / 14).astype(np.float32)
    errs = {}
    for sh in (0, 12, 18, 24):
        x = (rng.standard_normal((192, 300)) * 2.0 ** -sh).astype(np.float32)
        x[0, 0] = 1.0                                   # one element pins the maximum
        ref = w.astype(np.float64) @ x.astype(np.float64)
        y = split_matmul(w, x)
        errs[sh] = np.abs(y - ref)[:, 1:].max() / np.abs(ref)[:, 1:].max()      # columns that only hold small elements
    assert errs[0] < 1e-6 and errs[12] < 1e-6 and errs[18] < 2e-6 and errs[24] < 1e-4, errs


def test_zero_and_non_finite_maxima_leave_scale_one():
    assert pow2_to(0.0, X_TARGET) == 1.0
    assert pow2_to(np.inf, X_TARGET) == 1.0
    assert pow2_to(np.nan, X_TARGET) == 1.0
    assert pow2_to(1e-45, X_TARGET) == 1.0            # subnormal fp32
    assert pow2_to(1.0, X_TARGET) == 2.0 ** 14        # 1.0 = 0.5 * 2^1
    assert pow2_to(0.75, W_TARGET) == 2.0 ** 10
