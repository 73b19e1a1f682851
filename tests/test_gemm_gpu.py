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
