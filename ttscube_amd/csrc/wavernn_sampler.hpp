// Sampler stage of the continuous WaveRNN output distributions (MOL — the reference default —, Gaussian, Beta:
// cube/networks/loss.py:35-215), shared by the streaming kernel (wavernn.hip) and the tile kernel (wavernn_tile.hip) so that
// both stay bit-identical to oracle/wavernn_ref.c (one arithmetic definition: include/ttscube_math.h).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/ttscube_hip.h"
#include "../../include/ttscube_math.h"

namespace ttsc {

// One wave samples one utterance: y = its S output-layer values (LDS), o = b * L + t.  All 64 lanes call; wv / bi are valid
// on every lane.  The per-scalar noise terms are computed one per lane (Philox rounds + logs would otherwise sit serially on
// the per-step critical path).
__device__ __forceinline__ void wr_sample_continuous(int out_kind, int mode, const float* y, const float* noise, size_t o, int t, int b,
                                                     unsigned long long seed, int lane, float& wv, int& bi) {
    wv = 0.f;
    bi = 0;
    if (out_kind == TTSC_WR_OUT_MOL) {
        float gi = 0.f;
        if (lane <= TTSC_MOL_NMIX) {
            if (mode == 1) {
                gi = noise[o * TTSC_MOL_NOISE + lane];
            } else if (mode == 2) {
                const float uu = ttsc_u01_clip(ttsc_philox_word((uint32_t)lane, (uint32_t)t, (uint32_t)b, seed));
                gi = lane < TTSC_MOL_NMIX ? -ttsc_logf(-ttsc_logf(uu)) : ttsc_logistic(uu);
            }
        }
        float v = lane < TTSC_MOL_NMIX ? y[lane] + gi : -INFINITY;
        int k = lane;
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) {
            const float ov = __shfl_xor(v, off);
            const int ok = __shfl_xor(k, off);
            if (ov > v || (ov == v && ok < k)) {   // first maximum wins, like the oracle's sequential scan
                v = ov;
                k = ok;
            }
        }
        const float lg = __shfl(gi, TTSC_MOL_NMIX);
        k = __shfl(k, 0);
        const float mean = y[TTSC_MOL_NMIX + k];
        float ls = y[2 * TTSC_MOL_NMIX + k];
        ls = ls < TTSC_LOG_SCALE_MIN ? TTSC_LOG_SCALE_MIN : ls;
        wv = mean + ttsc_expf(ls) * lg;
        wv = wv < -1.0f ? -1.0f : wv;
        wv = wv > 1.0f ? 1.0f : wv;
        bi = k;
    } else if (out_kind == TTSC_WR_OUT_GM) {
        float z = 0.f;
        if (mode == 1)
            z = noise[o];
        else if (mode == 2)
            z = 0.8f * ttsc_normal_icdf(ttsc_u01(ttsc_philox_word(0u, (uint32_t)t, (uint32_t)b, seed)));
        wv = y[0] + z * ttsc_expf(y[1]);
    } else {   // beta: lanes 0 / 1 draw the two gamma variates
        constexpr int NV = 1 + 2 * TTSC_BETA_TRIES;
        const int v = lane & 1;
        float nz[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) nz[i] = 0.f;
        if (mode == 1) {
#pragma unroll
            for (int i = 0; i < NV; ++i) nz[i] = noise[o * TTSC_BETA_NOISE + v * NV + i];
        } else if (mode == 2) {
            nz[0] = ttsc_u01(ttsc_philox_word((uint32_t)(v * 16), (uint32_t)t, (uint32_t)b, seed));
#pragma unroll
            for (int i = 0; i < TTSC_BETA_TRIES; ++i) {
                nz[1 + 2 * i] = ttsc_normal_icdf(ttsc_u01(ttsc_philox_word((uint32_t)(v * 16 + 1 + 2 * i), (uint32_t)t, (uint32_t)b, seed)));
                nz[2 + 2 * i] = ttsc_u01(ttsc_philox_word((uint32_t)(v * 16 + 2 + 2 * i), (uint32_t)t, (uint32_t)b, seed));
            }
        } else {
            nz[0] = 0.5f;
            nz[2] = 0.5f;
        }
        const float gv = ttsc_gamma_mt(ttsc_expf(y[v]), nz);
        const float ga = __shfl(gv, 0), gb = __shfl(gv, 1);
        float sx = ga / (ga + gb);
        sx = sx < 1.17549435e-38f ? 1.17549435e-38f : sx;
        sx = sx > 0.99999994f ? 0.99999994f : sx;
        wv = (sx - 0.5f) * 2.0f;
    }
}

}  // namespace ttsc
