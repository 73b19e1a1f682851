"""GPU: the general fp32-MFMA GEMM (ttsc_gemm: NN / TN / NT / TT, split-K, row-shifted B) and the column sums against float64 —
the backward GEMMs of the Linears and of the LSTM / GRU projections (reference: autograd over nn.Linear / nn.GRU / nn.LSTM,
cube/networks/modules.py:505-563)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a.double() - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize('M,N,K', [(7, 5, 3), (130, 129, 17), (384, 102, 5000), (1536, 512, 24000), (256, 256, 16), (33, 1536, 4100)])
@pytest.mark.parametrize('ta,tb', [(False, False), (True, False), (False, True), (True, True)])
def test_gemm_variants_match_float64(M, N, K, ta, tb):
    from ttscube_amd.hip_layers import gemm_hip
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    a = torch.randn((K, M) if ta else (M, K), generator=g).cuda()
    b = torch.randn((N, K) if tb else (K, N), generator=g).cuda()
    ref = (a.double().t() if ta else a.double()) @ (b.double().t() if tb else b.double())
    out = gemm_hip(a, b, trans_a=ta, trans_b=tb)
    assert _rel(out, ref) < 2e-6
    # deterministic (fixed-order split-K reduction) and accumulating
    assert torch.equal(out, gemm_hip(a, b, trans_a=ta, trans_b=tb))
    acc = torch.ones_like(out)
    gemm_hip(a, b, trans_a=ta, trans_b=tb, out=acc, accumulate=True)
    assert _rel(acc, ref + 1) < 2e-6


def test_gemm_strided_operands_and_row_shift():
    """column slices of wider tensors as operands (leading dimensions), and dW_hh = dG^T . h_prev without materialising h_prev"""
    from ttscube_amd.hip_layers import gemm_hip
    B, T, H = 3, 700, 64
    g = torch.Generator().manual_seed(5)
    dG = torch.randn(B * T, 8 * H, generator=g).cuda()
    y = torch.randn(B, T, 2 * H, generator=g).cuda()
    for d, shift in ((0, -1), (1, 1)):
        hy = y[:, :, d * H:(d + 1) * H]
        hprev = torch.zeros_like(hy)
        if d == 0:
            hprev[:, 1:] = hy[:, :-1]
        else:
            hprev[:, :-1] = hy[:, 1:]
        sl = slice(d * 4 * H, (d + 1) * 4 * H)
        ref = dG[:, sl].double().t() @ hprev.reshape(B * T, H).double()
        out = gemm_hip(dG[:, sl], y.reshape(B * T, 2 * H)[:, d * H:(d + 1) * H], trans_a=True, b_row_shift=shift, b_period=T)
        assert _rel(out, ref) < 2e-6, d


@pytest.mark.parametrize('R,C', [(5, 3), (4096, 257), (384000, 256), (100, 1536)])
def test_colsum_matches_float64(R, C):
    from ttscube_amd.hip_layers import colsum_hip
    x = torch.randn(R, C, generator=torch.Generator().manual_seed(R + C)).cuda()
    out = colsum_hip(x)
    ref = x.double().sum(dim=0)
    assert float((out.double() - ref).abs().max()) < 1e-5 * (R ** 0.5 + 1)
    assert torch.equal(out, colsum_hip(x))
    wide = torch.randn(R, C + 9, generator=torch.Generator().manual_seed(1)).cuda()
    assert float((colsum_hip(wide[:, 4:4 + C]).double() - wide[:, 4:4 + C].double().sum(dim=0)).abs().max()) < 1e-5 * (R ** 0.5 + 1)


def test_gemm_rejects_cpu_tensors():
    from ttscube_amd._lib import TTSCError
    from ttscube_amd.hip_layers import gemm_hip
    with pytest.raises(TTSCError):
        gemm_hip(torch.zeros(2, 2), torch.zeros(2, 2))


@pytest.mark.parametrize('M,N,K,act', [(300, 2048, 256, None), (77, 80, 128, 'tanh'), (1000, 101, 512, None), (5, 2, 64, 'sigmoid'), (4480, 2048, 640, None),
                                       (129, 129, 36, None)])
def test_split_linear_matches_float64(M, N, K, act):
    """ttsc_linear_forward_split (fp16 hi / lo halves, three MFMA products): fp32-class accuracy against float64 — per element within
    2^-20 of sum |x||w| (the dropped lo.lo term and the operand roundings are ~2^-22 per product), as close to float64 as the exact fp32 kernel on
    average, bit-reproducible, accumulating."""
    from ttscube_amd import _lib
    from ttscube_amd.hip_layers import linear_hip
    g = torch.Generator().manual_seed(M + N + K)
    x = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))).cuda()      # rows of different scale
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    pre = x.double() @ w.double().t() + b.double()
    ref = {None: pre, 'tanh': torch.tanh(pre), 'sigmoid': torch.sigmoid(pre)}[act]
    bound = (x.double().abs() @ w.double().abs().t()) * 2.0 ** -20 + 1e-6
    y = linear_hip(x, w, b, act=act, split=True)
    ex = linear_hip(x, w, b, act=act)
    assert bool(((y.double() - ref).abs() <= bound).all())
    assert _rel(y, ref) < 3 * max(_rel(ex, ref), 1e-7)
    assert torch.equal(y, linear_hip(x, w, b, act=act, split=True))
    if act is None:
        acc = torch.ones_like(y)
        linear_hip(x, w, None, out=acc, accumulate=True, split=True)
        assert _rel(acc, pre - b.double() + 1) < 1e-6
    assert int(_lib.lib().ttsc_gemm_split_status()) == 0


def test_split_linear_skips_padding_tiles_and_guards_the_range():
    from ttscube_amd import _lib
    from ttscube_amd.hip_layers import linear_hip
    B, T, K, N = 5, 300, 96, 200
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, T, K, generator=g).cuda()
    w = torch.randn(N, K, generator=g).cuda()
    lens = [300, 10, 129, 0, 257]
    ld = torch.tensor(lens, dtype=torch.int32).cuda()
    full = linear_hip(x, w, None, split=True)
    out = torch.full((B * T, N), 7.0, device='cuda')
    linear_hip(x, w, None, out=out, split=True, lengths_dev=ld, period=T)
    out = out.view(B, T, N)
    for b_, n in enumerate(lens):
        assert torch.equal(out[b_, :n], full[b_, :n])                       # rows an utterance owns: the same bits
    # 128-row tiles: rows 640..767 = utterance 2's rows 40..167 hold live rows; rows 896..1023 = utterance 2 tail + utterance 3 (length 0) +
    # ... — check one tile that is padding only: utterance 3 occupies rows 900..1199; tile rows 1024..1151 lie wholly inside it -> untouched
    flat = out.view(B * T, N)
    assert bool((flat[1024:1152] == 7.0).all())
    # an operand beyond the fp16 range is reported (and the report clears)
    xb = x.clone()
    xb[0, 0, 0] = 1e5
    linear_hip(xb, w, None, split=True)
    with pytest.raises(_lib.TTSCError):
        _lib.check_split_status('test')
    _lib.check_split_status('test')


@pytest.mark.parametrize('R,Cc', [(128, 15), (1024, 5120), (1, 3072), (512, 656), (40, 7)])
def test_matvec_both_directions(R, Cc):
    import ctypes as C
    from ttscube_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(R + Cc)
    W = torch.randn(R, Cc, generator=g).cuda()
    x = torch.randn(Cc, generator=g).cuda()
    u = torch.randn(R, generator=g).cuda()
    o0 = torch.empty(R, device='cuda')
    o1 = torch.empty(Cc, device='cuda')
    ws = torch.empty(max(int(L.ttsc_matvec_workspace_bytes(R, Cc)) // 4, 1), device='cuda')
    P = _lib.dev_ptr
    _lib.check(L.ttsc_matvec(P(W), R, Cc, P(x), 0, P(o0), None, 0, _lib.current_stream()), 'matvec')
    _lib.check(L.ttsc_matvec(P(W), R, Cc, P(u), 1, P(o1), P(ws), ws.numel() * 4, _lib.current_stream()), 'matvec')
    assert _rel(o0, W.double() @ x.double()) < 2e-6 and _rel(o1, W.double().t() @ u.double()) < 2e-6
    o2 = torch.empty_like(o1)
    _lib.check(L.ttsc_matvec(P(W), R, Cc, P(u), 1, P(o2), P(ws), ws.numel() * 4, _lib.current_stream()), 'matvec')
    assert torch.equal(o1, o2)


def test_rows_segment_sum_equals_the_scatter_for_sorted_indices():
    """phoneme rows -> frame rows (training expand): the O(n C) adjoint for a non-decreasing index list gives the bits of the general kernel"""
    from ttscube_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(9)
    V, Cc = 700, 640
    counts = torch.randint(0, 12, (V,), generator=g)
    counts[5] = 0
    counts[V - 1] = 0
    idx = torch.repeat_interleave(torch.arange(V), counts).to(torch.int32).cuda()
    n = idx.numel()
    gout = torch.randn(n, Cc, generator=g).cuda()
    a = torch.empty(V, Cc, device='cuda')
    b = torch.full((V, Cc), 3.0, device='cuda')
    P = _lib.dev_ptr
    _lib.check(L.ttsc_rows_scatter_add(P(gout), P(idx), P(a), n, Cc, V, -1, _lib.current_stream()), 'scatter')
    _lib.check(L.ttsc_rows_segment_sum(P(gout), P(idx), P(b), n, Cc, V, _lib.current_stream()), 'segment')
    assert torch.equal(a, b) and bool((b[5] == 0).all()) and bool((b[V - 1] == 0).all())
    ref = torch.zeros(V, Cc, dtype=torch.float64, device='cuda').index_add_(0, idx.long(), gout.double())
    assert _rel(b, ref) < 1e-6
