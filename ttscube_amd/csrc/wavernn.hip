// WaveRNN autoregressive decode as ONE persistent kernel per call (gfx950).
//
// Reference loop: cube/networks/modules.py:453-503 (WaveRNN._inference): per output sample ~10 ATen launches
// (cat, N x nn.GRU(seq=1), Linear+tanh, Linear, Categorical.sample, decode).  Here the whole L-step loop of an
// utterance tile runs inside one workgroup, with no inter-workgroup communication at all (utterances are
// independent), so the sequential dependency costs only workgroup barriers:
//
//   * thread j owns hidden unit j: it streams the three gate rows (r,z,n) of W_hh (pre-transposed to [k][3H],
//     so every load is one coalesced 256-B wave read from L2) and keeps the six gate accumulators of each of the
//     BT utterances of the tile in registers; the GRU state h lives in LDS (double-buffered) and is read as LDS
//     broadcasts; the point-wise GRU update is done by the same thread, so no exchange is needed between the
//     mat-vec and the state update.
//   * the input projection W_ih.x is an fmaf chain over [mel(80) | low-res feats(20) | interp(1) | last_x(1)] in
//     that (reference concat) order, so its prefix over the mel frame is constant for `upsample` (240) steps and
//     its prefix over the low-res features for 10 steps: both prefixes are cached in registers and only the last
//     two terms are evaluated per step — bit-identical to evaluating the whole chain every step.
//   * pre-output (tanh Linear H->256), output (Linear 256->S), Gumbel-max sampling, µ-law decode and the
//     feedback of the sample all stay inside the workgroup (LDS + 3 barriers).
//
// Arithmetic contract = oracle/wavernn_ref.c: every dot product is one k-ordered fp32 fmaf chain seeded with the
// bias, transcendental functions from include/ttscube_math.h, compiled with -ffp-contract=off.  That makes the
// uint8 sample indices bit-exact against the oracle (tests/test_wavernn_gpu.py).
#include <cstdlib>

#include "common.hpp"
#include "../../include/ttscube_math.h"
#include "../../include/ttscube_mulaw_lut.h"
#include "wavernn_sampler.hpp"
#include "wavernn_tile.hip"

namespace ttsc {

constexpr int WR_THREADS = 512;
constexpr int WR_MAXL = 4;

struct WrArgs {
    const float* mel;     // [B, T, n_mel]
    const float* interp;  // [B, Tl*up_low]   (hr only)
    const float* feats;   // [B, 20, Tl]      (hr only)
    const float* wt_ih[WR_MAXL];  // layer 0: [in_0][3H]; layers > 0: [H/4][3H][4]
    const float* wt_hh[WR_MAXL];  // [H/4][3H][4]
    const float* b_ih[WR_MAXL];
    const float* b_hh[WR_MAXL];
    const float* wt_pre;  // [H/4][256][4]
    const float* b_pre;
    const float* wt_out;  // [256/4][S][4]
    const float* b_out;
    const float* lut;
    const float* noise;     // [B, L, S] or null
    const float* forced_x;  // [B, L] or null
    uint8_t* out_idx;       // [B, L]
    float* out_wav;         // [B, L]
    float* out_logits;      // [B, L, S] or null
    int B, T, Tl, H, NL, use_lowres, up, up_low, S, n_mel, out_kind, mode;
    int L;
    unsigned long long seed;
};

// ---- conditioning: linear interpolation x10 and the three tanh(Conv1d k7) low-res layers --------------------
// (modules.py:353,416-420,458-461).  One workgroup per utterance; chain order = (ci outer, k inner), bias seed.
__global__ __launch_bounds__(256) void wr_cond_kernel(const float* __restrict__ x_low, const float* w0, const float* b0,
                                                      const float* w1, const float* b1, const float* w2, const float* b2,
                                                      float* interp, float* fa, float* fb, int Tl, int up_low) {
    const int b = blockIdx.x;
    const float* x = x_low + (size_t)b * Tl;
    const int n = Tl * up_low;
    const float scale = (float)Tl / (float)n;
    for (int t = threadIdx.x; t < n; t += blockDim.x) {
        float src = scale * ((float)t + 0.5f) - 0.5f;
        if (src < 0.f) src = 0.f;
        int i0 = (int)src;
        if (i0 > Tl - 1) i0 = Tl - 1;
        const int i1 = i0 + (i0 < Tl - 1 ? 1 : 0);
        const float l1 = src - (float)i0;
        const float l0 = 1.0f - l1;
        const float p0 = l0 * x[i0];
        const float p1 = l1 * x[i1];
        interp[(size_t)b * n + t] = p0 + p1;
    }
    float* A = fa + (size_t)b * 20 * Tl;
    float* Bf = fb + (size_t)b * 20 * Tl;
    for (int layer = 0; layer < 3; ++layer) {
        const float* in = layer == 0 ? x : (layer == 1 ? A : Bf);
        float* out = layer == 1 ? Bf : A;
        const float* w = layer == 0 ? w0 : (layer == 1 ? w1 : w2);
        const float* bb = layer == 0 ? b0 : (layer == 1 ? b1 : b2);
        const int cin = layer == 0 ? 1 : 20;
        for (int idx = threadIdx.x; idx < 20 * Tl; idx += blockDim.x) {
            const int co = idx / Tl, t = idx - co * Tl;
            float acc = bb[co];
            for (int ci = 0; ci < cin; ++ci)
                for (int k = 0; k < 7; ++k) {
                    const int p = t + k - 3;
                    const float v = (p >= 0 && p < Tl) ? in[ci * Tl + p] : 0.f;
                    acc = fmaf(w[(co * cin + ci) * 7 + k], v, acc);
                }
            out[idx] = ttsc_tanhf(acc);
        }
        __threadfence_block();
        __syncthreads();
    }
}

// acc[u][g] <- k-ordered fmaf chain over K inputs for NG rows (row g at rows-offset g*gstride + row) of a weight matrix
// packed as [K/4][rows][4] (four consecutive k of one row are one 16-byte load; consecutive threads = consecutive rows, so
// a wave reads 1 KiB contiguous per load).  v: LDS vector(s) [BT][vstride], read as 16-byte broadcasts.
template <int BT, int NG, int UN>
__device__ __forceinline__ void chain_matvec(float (&acc)[BT][NG], const float* __restrict__ wp, int rows, int gstride, int row,
                                             const float* v, int vstride, int K) {
    // Weight stream with EXPLICIT software pipelining: the 16-byte loads of a whole batch (UN k-blocks x NG rows) are
    // issued back to back into one register set while the fmaf chain consumes the other set.  Left to itself hipcc
    // places each load right before its use and waits vmcnt(0) per load, i.e. one L2 round trip per 16 bytes.
    // The chain order (k ascending, one fmaf per term) is unchanged.
    asm volatile("" : "+v"(row));   // (keeps the UN x NG load addresses from being hoisted out of the caller's step loop as 64-bit per-lane values: see rnn_chain.hpp)
    const float4* w4 = reinterpret_cast<const float4*>(wp) + row;
    const int KB = K >> 2;
    auto load = [&](float4 (&w)[UN][NG], int kb0) {
#pragma unroll
        for (int q = 0; q < UN; ++q)
#pragma unroll
            for (int g = 0; g < NG; ++g) w[q][g] = w4[(size_t)(kb0 + q) * rows + g * gstride];
    };
    auto fma_batch = [&](const float4 (&w)[UN][NG], int kb0) {
#pragma unroll
        for (int q = 0; q < UN; ++q) {
#pragma unroll
            for (int u = 0; u < BT; ++u) {
                const float4 hv = *reinterpret_cast<const float4*>(v + u * vstride + 4 * (kb0 + q));
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    float x = acc[u][g];
                    x = fmaf(w[q][g].x, hv.x, x);
                    x = fmaf(w[q][g].y, hv.y, x);
                    x = fmaf(w[q][g].z, hv.z, x);
                    x = fmaf(w[q][g].w, hv.w, x);
                    acc[u][g] = x;
                }
            }
        }
    };
    if (KB % UN == 0) {
        float4 wa[UN][NG], wb[UN][NG];
        const int NB = KB / UN;
        load(wa, 0);
        for (int bi = 0; bi < NB; bi += 2) {
            if (bi + 1 < NB) load(wb, (bi + 1) * UN);
            fma_batch(wa, bi * UN);
            if (bi + 2 < NB) load(wa, (bi + 2) * UN);
            if (bi + 1 < NB) fma_batch(wb, (bi + 1) * UN);
        }
    } else {
        for (int kb = 0; kb < KB; ++kb) {
            float4 w[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g) w[g] = w4[(size_t)kb * rows + g * gstride];
#pragma unroll
            for (int u = 0; u < BT; ++u) {
                const float4 hv = *reinterpret_cast<const float4*>(v + u * vstride + 4 * kb);
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    float x = acc[u][g];
                    x = fmaf(w[g].x, hv.x, x);
                    x = fmaf(w[g].y, hv.y, x);
                    x = fmaf(w[g].z, hv.z, x);
                    x = fmaf(w[g].w, hv.w, x);
                    acc[u][g] = x;
                }
            }
        }
    }
}

// The two output Linears (pre-output H -> 256, output 256 -> S): FOUR k-ordered chains over consecutive quarters of the inputs (in blocks of 4:
// quarter q = blocks [q KB / 4, (q + 1) KB / 4)), the first seeded with the bias, added as ((p0 + p1) + p2) + p3 — the contract of
// oracle/wavernn_ref.c::matvec_chain4, which the tile kernel (wavernn_tile.hip) evaluates with the four quarters on four waves.
__device__ __forceinline__ float chain_matvec_quarters(const float* __restrict__ wp, int rows, int row, const float* v, int K, float bias) {
    const int KB = K >> 2;
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int b0 = (q * KB) >> 2, b1 = ((q + 1) * KB) >> 2;
        float acc[1][1] = {{q == 0 ? bias : 0.f}};
        chain_matvec<1, 1, 4>(acc, wp + (size_t)b0 * rows * 4, rows, 0, row, v + 4 * b0, 0, 4 * (b1 - b0));
        tot = q == 0 ? acc[0][0] : tot + acc[0][0];
    }
    return tot;
}

template <int BT>
__global__ __launch_bounds__(WR_THREADS) void wr_decode_kernel(WrArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int H = a.H, S = a.S, NL = a.NL, NM = a.n_mel;
    float* hbuf = sm;                          // [2][NL][BT][H]
    float* pre = hbuf + 2 * NL * BT * H;       // [BT][256]
    float* score = pre + BT * 256;             // [BT][S]
    float* xin = score + BT * S;               // [BT][128]: mel frame | low-res feats
    float* lastx = xin + BT * 128;             // [BT]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int j = tid;                          // hidden unit owned by this thread
    const bool unit = j < H;
    const int H3 = 3 * H;
    const int I0 = NM + (a.use_lowres ? 21 : 0) + 1;

    // utterance index of tile slot u (clamped for reads; writes are guarded by BOK).  Computed, not stored in an
    // array: a runtime-indexed register array would be demoted to scratch.
#define BIDX(u) (min((int)blockIdx.x * BT + (u), a.B - 1))
#define BOK(u) ((int)blockIdx.x * BT + (u) < a.B)
    for (int i = tid; i < 2 * NL * BT * H; i += WR_THREADS) hbuf[i] = 0.f;
    if (tid < BT) lastx[tid] = 0.f;

    // per-thread constants of layer 0: biases and the weights of the two per-step input features
    float bih0[3] = {0, 0, 0}, w_int[3] = {0, 0, 0}, w_lx[3] = {0, 0, 0};
    if (unit) {
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            bih0[g] = a.b_ih[0][g * H + j];
            w_lx[g] = a.wt_ih[0][(size_t)(I0 - 1) * H3 + g * H + j];
            if (a.use_lowres) w_int[g] = a.wt_ih[0][(size_t)(I0 - 2) * H3 + g * H + j];
        }
    }
    float pmel[BT][3], plow[BT][3];
#pragma unroll
    for (int u = 0; u < BT; ++u)
#pragma unroll
        for (int g = 0; g < 3; ++g) pmel[u][g] = plow[u][g] = 0.f;
    __syncthreads();

    int cur = 0;
    int fr = 0, fr_phase = 0;  // t / up, t % up   (incremental: no integer division in the step loop)
    int lo = 0, lo_phase = 0;  // t / up_low, t % up_low
    for (int t = 0; t < a.L; ++t) {
        // ---- refresh the cached prefixes of the layer-0 input chain --------------------------------------
        const bool new_frame = fr_phase == 0;
        const bool new_low = a.use_lowres && lo_phase == 0;
        if (new_frame || new_low) {
            if (new_frame) {
                for (int i = tid; i < BT * NM; i += WR_THREADS) {
                    const int u = i / NM, k = i - u * NM;
                    xin[u * 128 + k] = a.mel[((size_t)BIDX(u) * a.T + fr) * NM + k];
                }
            }
            if (new_low) {
                for (int i = tid; i < BT * 20; i += WR_THREADS) {
                    const int u = i / 20, q = i - u * 20;
                    xin[u * 128 + NM + q] = a.feats[((size_t)BIDX(u) * 20 + q) * a.Tl + lo];
                }
            }
            __syncthreads();
            if (unit) {
                if (new_frame) {
#pragma unroll
                    for (int u = 0; u < BT; ++u)
#pragma unroll
                        for (int g = 0; g < 3; ++g) pmel[u][g] = bih0[g];
                    const float* w = a.wt_ih[0] + j;
                    for (int k = 0; k < NM; ++k) {
                        const float w0 = w[(size_t)k * H3], w1 = w[(size_t)k * H3 + H], w2 = w[(size_t)k * H3 + 2 * H];
#pragma unroll
                        for (int u = 0; u < BT; ++u) {
                            const float v = xin[u * 128 + k];
                            pmel[u][0] = fmaf(w0, v, pmel[u][0]);
                            pmel[u][1] = fmaf(w1, v, pmel[u][1]);
                            pmel[u][2] = fmaf(w2, v, pmel[u][2]);
                        }
                    }
                }
                if (new_low) {
#pragma unroll
                    for (int u = 0; u < BT; ++u)
#pragma unroll
                        for (int g = 0; g < 3; ++g) plow[u][g] = pmel[u][g];
                    const float* w = a.wt_ih[0] + (size_t)NM * H3 + j;
                    for (int k = 0; k < 20; ++k) {
                        const float w0 = w[(size_t)k * H3], w1 = w[(size_t)k * H3 + H], w2 = w[(size_t)k * H3 + 2 * H];
#pragma unroll
                        for (int u = 0; u < BT; ++u) {
                            const float v = xin[u * 128 + NM + k];
                            plow[u][0] = fmaf(w0, v, plow[u][0]);
                            plow[u][1] = fmaf(w1, v, plow[u][1]);
                            plow[u][2] = fmaf(w2, v, plow[u][2]);
                        }
                    }
                }
            }
        }

        // ---- GRU layers ----------------------------------------------------------------------------------
        const int nxt = cur ^ 1;
        for (int l = 0; l < NL; ++l) {
            const float* hc = hbuf + ((size_t)(cur * NL + l) * BT) * H;   // h_{t-1} of layer l
            float* hn = hbuf + ((size_t)(nxt * NL + l) * BT) * H;         // h_t of layer l
            if (unit) {
                float gi[BT][3], gh[BT][3];
                if (l == 0) {
#pragma unroll
                    for (int u = 0; u < BT; ++u) {
                        const float lx = lastx[u];
#pragma unroll
                        for (int g = 0; g < 3; ++g) {
                            float acc = a.use_lowres ? plow[u][g] : pmel[u][g];
                            if (a.use_lowres) acc = fmaf(w_int[g], a.interp[(size_t)BIDX(u) * ((size_t)a.Tl * a.up_low) + t], acc);
                            gi[u][g] = fmaf(w_lx[g], lx, acc);
                        }
                    }
                } else {
                    const float* hp = hbuf + ((size_t)(nxt * NL + (l - 1)) * BT) * H;  // fresh output of layer l-1
#pragma unroll
                    for (int u = 0; u < BT; ++u)
#pragma unroll
                        for (int g = 0; g < 3; ++g) gi[u][g] = a.b_ih[l][g * H + j];
                    chain_matvec<BT, 3, 2>(gi, a.wt_ih[l], H3, H, j, hp, H, H);
                }
#pragma unroll
                for (int u = 0; u < BT; ++u)
#pragma unroll
                    for (int g = 0; g < 3; ++g) gh[u][g] = a.b_hh[l][g * H + j];
                chain_matvec<BT, 3, 2>(gh, a.wt_hh[l], H3, H, j, hc, H, H);
#pragma unroll
                for (int u = 0; u < BT; ++u) {
                    const float r = ttsc_sigmoidf(gi[u][0] + gh[u][0]);
                    const float z = ttsc_sigmoidf(gi[u][1] + gh[u][1]);
                    const float rg = r * gh[u][2];
                    const float nn = ttsc_tanhf(gi[u][2] + rg);
                    const float d = hc[u * H + j] - nn;
                    hn[u * H + j] = fmaf(z, d, nn);
                }
            }
            __syncthreads();
        }

        // ---- pre-output: tanh(Linear H -> 256) -------------------------------------------------------------
        {
            const float* ht = hbuf + ((size_t)(nxt * NL + (NL - 1)) * BT) * H;
            const int row = tid & 255;
            for (int u = tid >> 8; u < BT; u += WR_THREADS / 256) {
                pre[u * 256 + row] = ttsc_tanhf(chain_matvec_quarters(a.wt_pre, 256, row, ht + u * H, H, a.b_pre[row]));
            }
        }
        __syncthreads();
        // ---- output logits + noise ------------------------------------------------------------------------
        {
            const int row = tid & 255;
            for (int u = tid >> 8; u < BT; u += WR_THREADS / 256) {
                if (row < S) {
                    const float acc = chain_matvec_quarters(a.wt_out, S, row, pre + u * 256, 256, a.b_out[row]);
                    const size_t o = ((size_t)BIDX(u) * a.L + t) * S + row;
                    if (a.out_logits && BOK(u)) a.out_logits[o] = acc;
                    float g = 0.f;
                    if (a.out_kind >= 2) {
                        // continuous outputs: the noise enters in the sampler stage below
                    } else if (a.mode == 1) {
                        g = a.noise[o];
                    } else if (a.mode == 2) {
                        uint32_t r4[4];
                        ttsc_philox4x32((uint32_t)(row >> 2), (uint32_t)t, (uint32_t)BIDX(u), 0u,
                                        (uint32_t)a.seed, (uint32_t)(a.seed >> 32), r4);
                        g = ttsc_gumbel(r4[row & 3]);
                    }
                    score[u * S + row] = acc + g;
                }
            }
        }
        __syncthreads();
        // ---- continuous output distributions (MOL is the reference default): a few lanes of wave u sample utterance u ----
        // Same arithmetic as oracle/wavernn_ref.c through include/ttscube_math.h; the per-scalar noise terms are computed
        // one per lane (Philox rounds + logs would otherwise sit serially on the per-step critical path).
        if (a.out_kind >= 2) {
            for (int u = wave; u < BT; u += WR_THREADS / 64) {
                const float* y = score + u * S;
                const size_t o = (size_t)BIDX(u) * a.L + t;
                float wv;
                int bi;
                wr_sample_continuous(a.out_kind, a.mode, y, a.noise, o, t, BIDX(u), a.seed, lane, wv, bi);
                if (lane == 0) {
                    if (BOK(u)) {
                        a.out_idx[o] = (uint8_t)bi;
                        a.out_wav[o] = wv;
                    }
                    lastx[u] = a.forced_x ? a.forced_x[o] : wv;
                }
            }
        }
        // ---- Gumbel-max: first maximum wins; wave u reduces utterance u ------------------------------------
        for (int u = wave; a.out_kind < 2 && u < BT; u += WR_THREADS / 64) {
            float bs = score[u * S + lane];
            int bi = lane;
            for (int s = lane + 64; s < S; s += 64) {
                const float v = score[u * S + s];
                if (v > bs) {
                    bs = v;
                    bi = s;
                }
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const float os = __shfl_xor(bs, off);
                const int oi = __shfl_xor(bi, off);
                if (os > bs || (os == bs && oi < bi)) {
                    bs = os;
                    bi = oi;
                }
            }
            if (lane == 0) {
                float wv;
                if (a.out_kind == 0)
                    wv = a.lut[bi];
                else
                    wv = (((float)bi / 255.0f) - 0.5f) * 2.0f;
                const size_t o = (size_t)BIDX(u) * a.L + t;
                if (BOK(u)) {
                    a.out_idx[o] = (uint8_t)bi;
                    a.out_wav[o] = wv;
                }
                lastx[u] = a.forced_x ? a.forced_x[o] : wv;
            }
        }
        __syncthreads();
        cur = nxt;
        if (++fr_phase == a.up) { fr_phase = 0; ++fr; }
        if (++lo_phase == a.up_low) { lo_phase = 0; ++lo; }
    }
}

#undef BIDX
#undef BOK

}  // namespace ttsc

using namespace ttsc;

struct ttsc_wavernn {
    ttsc_wavernn_cfg cfg;
    int in0 = 0;
    float* wt_ih[WR_MAXL] = {nullptr, nullptr, nullptr, nullptr};
    float* wt_hh[WR_MAXL] = {nullptr, nullptr, nullptr, nullptr};
    float* b_ih[WR_MAXL] = {nullptr, nullptr, nullptr, nullptr};
    float* b_hh[WR_MAXL] = {nullptr, nullptr, nullptr, nullptr};
    float *wt_pre = nullptr, *b_pre = nullptr, *wt_out = nullptr, *b_out = nullptr, *lut = nullptr;
    float* lc_w[3] = {nullptr, nullptr, nullptr};
    float* lc_b[3] = {nullptr, nullptr, nullptr};
    // host copies (torch layout) kept for the tile kernel's per-member packing
    std::vector<float> h_wih0, h_whh0, h_bih0, h_bhh0, h_wpre, h_bpre, h_wout, h_bout;
    std::vector<float> h_wih1, h_whh1, h_bih1, h_bhh1;   // second GRU layer (the reference class default is num_layers = 2)
    float *c_whh = nullptr, *c_wih = nullptr, *c_bih = nullptr, *c_bhh = nullptr, *c_wpre = nullptr, *c_bpre = nullptr, *c_wout = nullptr,
          *c_bout = nullptr;
    float *q_whh = nullptr, *q_wih = nullptr, *q_bih = nullptr, *q_bhh = nullptr, *q_wpre = nullptr, *q_bpre = nullptr, *q_wout = nullptr,
          *q_bout = nullptr;   // tile kernel (wavernn_tile.hip): 8 row slices of every matrix
    float *q_whh2 = nullptr, *q_wih2 = nullptr, *q_bih2 = nullptr, *q_bhh2 = nullptr;   // ... and of the second GRU layer
    bool tile_dirty = true;
    int last_kind = 0;         // 0 streaming kernel, 2 tile kernel
    unsigned* last_abort_word = nullptr;   // device word set by the tile kernel when a hand-off timed out
    std::vector<std::string> have;
    bool has(const std::string& n) const {
        for (auto& s : have)
            if (s == n) return true;
        return false;
    }
};

static int upload(float** dst, const float* host, size_t n) {
    if (*dst) (void)hipFree(*dst);
    *dst = nullptr;
    TTSC_HIP_CHECK(hipMalloc((void**)dst, n * sizeof(float)));
    TTSC_HIP_CHECK(hipMemcpy(*dst, host, n * sizeof(float), hipMemcpyHostToDevice));
    return TTSC_OK;
}

// torch [rows, K] -> device [K/4][rows][4]: thread `row` reads four consecutive k as one 16-byte load (K % 4 == 0)
static int upload_packed4(float** dst, const float* host, int64_t rows, int64_t K) {
    std::vector<float> t((size_t)rows * K);
    for (int64_t r = 0; r < rows; ++r)
        for (int64_t k = 0; k < K; ++k) t[((size_t)(k >> 2) * rows + r) * 4 + (k & 3)] = host[(size_t)r * K + k];
    return upload(dst, t.data(), t.size());
}

// torch [rows, cols] -> device [cols][rows] so that consecutive threads (rows) read consecutive addresses
static int upload_transposed(float** dst, const float* host, int64_t rows, int64_t cols) {
    std::vector<float> t((size_t)rows * cols);
    for (int64_t r = 0; r < rows; ++r)
        for (int64_t c = 0; c < cols; ++c) t[(size_t)c * rows + r] = host[(size_t)r * cols + c];
    return upload(dst, t.data(), t.size());
}

extern "C" int ttsc_wavernn_create(const ttsc_wavernn_cfg* cfg, ttsc_wavernn** out) {
    TTSC_REQUIRE(cfg && out, "ttsc_wavernn_create: null argument");
    TTSC_REQUIRE(cfg->H >= 8 && cfg->H <= WR_THREADS && cfg->H % 8 == 0, "ttsc_wavernn_create: layer_size must be a multiple of 8 in [8, %d], got %d",
                 WR_THREADS, cfg->H);
    TTSC_REQUIRE(cfg->num_layers >= 1 && cfg->num_layers <= WR_MAXL, "ttsc_wavernn_create: num_layers must be in [1,%d]", WR_MAXL);
    TTSC_REQUIRE(cfg->S >= 1 && cfg->S <= 256, "ttsc_wavernn_create: sample_size must be in [1,256] (mulaw/raw = 256)");
    TTSC_REQUIRE(cfg->n_mel >= 1 && cfg->n_mel <= 100, "ttsc_wavernn_create: n_mel must be in [1,100]");
    TTSC_REQUIRE(cfg->upsample >= 1 && cfg->upsample_low >= 1, "ttsc_wavernn_create: bad upsample factors");
    TTSC_REQUIRE(cfg->out_kind >= TTSC_WR_OUT_MULAW && cfg->out_kind <= TTSC_WR_OUT_BETA, "ttsc_wavernn_create: unknown output kind %d", cfg->out_kind);
    {
        const int want = cfg->out_kind == TTSC_WR_OUT_MOL ? 3 * TTSC_MOL_NMIX : (cfg->out_kind >= TTSC_WR_OUT_GM ? 2 : 256);
        TTSC_REQUIRE(cfg->S == want, "ttsc_wavernn_create: output kind %d has sample_size %d (cube/networks/loss.py), got %d", cfg->out_kind, want, cfg->S);
    }
    ttsc_wavernn* w = new ttsc_wavernn();
    w->cfg = *cfg;
    w->in0 = cfg->n_mel + 1 + (cfg->use_lowres ? 21 : 0);
    int rc = upload(&w->lut, TTSC_MULAW_LUT, 256);
    if (rc) {
        delete w;
        return rc;
    }
    *out = w;
    return TTSC_OK;
}

extern "C" void ttsc_wavernn_destroy(ttsc_wavernn* w) {
    if (!w) return;
    for (int l = 0; l < WR_MAXL; ++l) {
        if (w->wt_ih[l]) (void)hipFree(w->wt_ih[l]);
        if (w->wt_hh[l]) (void)hipFree(w->wt_hh[l]);
        if (w->b_ih[l]) (void)hipFree(w->b_ih[l]);
        if (w->b_hh[l]) (void)hipFree(w->b_hh[l]);
    }
    for (float* p : {w->wt_pre, w->b_pre, w->wt_out, w->b_out, w->lut, w->lc_w[0], w->lc_w[1], w->lc_w[2], w->lc_b[0], w->lc_b[1], w->lc_b[2],
                     w->c_whh, w->c_wih, w->c_bih, w->c_bhh, w->c_wpre, w->c_bpre, w->c_wout, w->c_bout, w->q_whh2, w->q_wih2, w->q_bih2, w->q_bhh2,
                     w->q_whh, w->q_wih, w->q_bih, w->q_bhh, w->q_wpre, w->q_bpre, w->q_wout, w->q_bout})
        if (p) (void)hipFree(p);
    delete w;
}

static bool shape_is(const int64_t* shape, int nd, std::initializer_list<int64_t> want) {
    if (nd != (int)want.size()) return false;
    int i = 0;
    for (int64_t v : want)
        if (shape[i++] != v) return false;
    return true;
}

extern "C" int ttsc_wavernn_set_weight(ttsc_wavernn* w, const char* name, const float* host, const int64_t* shape, int32_t nd) {
    TTSC_REQUIRE(w && name && host && shape, "ttsc_wavernn_set_weight: null argument");
    const std::string n(name);
    const int H = w->cfg.H, S = w->cfg.S;
    int rc = TTSC_OK;
    int l = -1, idx = -1;
    char kind[32] = {0};
    if (n.compare(0, 6, "_skip.") == 0) return TTSC_OK;  // dead layer in every reference checkpoint (modules.py:424)
    if (sscanf(name, "_rnns.%d.%31s", &l, kind) == 2) {
        TTSC_REQUIRE(l >= 0 && l < w->cfg.num_layers, "ttsc_wavernn_set_weight: '%s': layer out of range", name);
        const int in_l = l == 0 ? w->in0 : H;
        const std::string k(kind);
        if (k == "weight_ih_l0") {
            TTSC_REQUIRE(shape_is(shape, nd, {3 * H, in_l}), "ttsc_wavernn_set_weight: '%s' expects [%d,%d]", name, 3 * H, in_l);
            rc = (l == 0) ? upload_transposed(&w->wt_ih[l], host, 3 * H, in_l) : upload_packed4(&w->wt_ih[l], host, 3 * H, in_l);
            if (l == 0) w->h_wih0.assign(host, host + (size_t)3 * H * in_l);
            if (l == 1) w->h_wih1.assign(host, host + (size_t)3 * H * in_l);
        } else if (k == "weight_hh_l0") {
            TTSC_REQUIRE(shape_is(shape, nd, {3 * H, H}), "ttsc_wavernn_set_weight: '%s' expects [%d,%d]", name, 3 * H, H);
            rc = upload_packed4(&w->wt_hh[l], host, 3 * H, H);
            if (l == 0) w->h_whh0.assign(host, host + (size_t)3 * H * H);
            if (l == 1) w->h_whh1.assign(host, host + (size_t)3 * H * H);
        } else if (k == "bias_ih_l0") {
            TTSC_REQUIRE(shape_is(shape, nd, {3 * H}), "ttsc_wavernn_set_weight: '%s' expects [%d]", name, 3 * H);
            rc = upload(&w->b_ih[l], host, 3 * H);
            if (l == 0) w->h_bih0.assign(host, host + 3 * H);
            if (l == 1) w->h_bih1.assign(host, host + 3 * H);
        } else if (k == "bias_hh_l0") {
            TTSC_REQUIRE(shape_is(shape, nd, {3 * H}), "ttsc_wavernn_set_weight: '%s' expects [%d]", name, 3 * H);
            rc = upload(&w->b_hh[l], host, 3 * H);
            if (l == 0) w->h_bhh0.assign(host, host + 3 * H);
            if (l == 1) w->h_bhh1.assign(host, host + 3 * H);
        } else {
            TTSC_REQUIRE(false, "ttsc_wavernn_set_weight: unknown key '%s'", name);
        }
    } else if (sscanf(name, "_lowres_conv.%d.conv.%31s", &idx, kind) == 2) {
        TTSC_REQUIRE(w->cfg.use_lowres && idx >= 0 && idx < 3, "ttsc_wavernn_set_weight: '%s' not part of this network", name);
        const int cin = idx == 0 ? 1 : 20;
        if (std::string(kind) == "weight") {
            TTSC_REQUIRE(shape_is(shape, nd, {20, cin, 7}), "ttsc_wavernn_set_weight: '%s' expects [20,%d,7]", name, cin);
            rc = upload(&w->lc_w[idx], host, 20 * cin * 7);
        } else {
            TTSC_REQUIRE(shape_is(shape, nd, {20}), "ttsc_wavernn_set_weight: '%s' expects [20]", name);
            rc = upload(&w->lc_b[idx], host, 20);
        }
    } else if (n == "_preoutput.linear_layer.weight") {
        TTSC_REQUIRE(shape_is(shape, nd, {256, H}), "ttsc_wavernn_set_weight: '%s' expects [256,%d]", name, H);
        rc = upload_packed4(&w->wt_pre, host, 256, H);
        w->h_wpre.assign(host, host + (size_t)256 * H);
    } else if (n == "_preoutput.linear_layer.bias") {
        TTSC_REQUIRE(shape_is(shape, nd, {256}), "ttsc_wavernn_set_weight: '%s' expects [256]", name);
        rc = upload(&w->b_pre, host, 256);
        w->h_bpre.assign(host, host + 256);
    } else if (n == "_output.linear_layer.weight") {
        TTSC_REQUIRE(shape_is(shape, nd, {S, 256}), "ttsc_wavernn_set_weight: '%s' expects [%d,256]", name, S);
        rc = upload_packed4(&w->wt_out, host, S, 256);
        w->h_wout.assign(host, host + (size_t)S * 256);
    } else if (n == "_output.linear_layer.bias") {
        TTSC_REQUIRE(shape_is(shape, nd, {S}), "ttsc_wavernn_set_weight: '%s' expects [%d]", name, S);
        rc = upload(&w->b_out, host, S);
        w->h_bout.assign(host, host + S);
    } else {
        TTSC_REQUIRE(false, "ttsc_wavernn_set_weight: unknown key '%s'", name);
    }
    if (rc == TTSC_OK && !w->has(n)) w->have.push_back(n);
    w->tile_dirty = true;
    return rc;
}

// ---- tile path (wavernn_tile.hip): 8 workgroups step 8 utterances, each owning 1/8 of the rows -------------
static size_t tile_lds_bytes(const ttsc_wavernn* w) {
    const auto& c = w->cfg;
    size_t n = (size_t)c.H * 32 + (size_t)256 * 32 + (size_t)WT_NC * (c.H + 4) + (size_t)WT_NC * 260 + (size_t)3 * (c.H / WT_NC) * WT_NC + 64 + (size_t)3 * (c.H / WT_NC) + 64;
    n += (size_t)2 * 4 * 32 * WT_NC;   // partial sums of the two output Linears: [quarter][32 rows][8 utterances] each
    if (c.num_layers == 2) n += (size_t)WT_NC * (c.H + 4) + (size_t)3 * (c.H / WT_NC) * WT_NC + (size_t)6 * (c.H / WT_NC);   // h2 vector, W_hh2 h2 products, b_ih2 | b_hh2
    return n * sizeof(float);
}

static bool tile_supported(const ttsc_wavernn* w, int B) {
    const auto& c = w->cfg;
    // Default for one-layer networks whenever every member can be resident (B <= 256 on MI355X);
    // env TTSC_WR_TILE=0 forces the streaming kernel.  Bit-exact either way.  Measurements: DESIGN.md.
    const char* ev = getenv("TTSC_WR_TILE");
    if (ev && atoi(ev) == 0) return false;
    if (c.num_layers > 2 || c.H % (4 * WT_NC) != 0 || c.H > 512 || c.S > 256) return false;
    if (c.num_layers == 2) {
        const char* e2 = getenv("TTSC_WR_TILE2");   // A/B switch: two-layer nets back on the streaming kernel
        if (e2 && atoi(e2) == 0) return false;
    }
    if (c.out_kind < 2 && c.S % WT_NC != 0) return false;   // (continuous heads: S = 30 / 2, padded to 32 / 8 rows)
    if (tile_lds_bytes(w) > 160 * 1024) return false;
    const int G = (int)ceil_div(B, WT_NC);
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
    return G * WT_NC <= cus;   // every member must be resident, otherwise the exchange cannot complete
}

// granules (8 bytes) of the exchange area of G tiles + the abort word
static size_t tile_exchange_granules(const ttsc_wavernn* w, int G) {
    const auto& c = w->cfg;
    return (size_t)G * 2 * ((size_t)c.num_layers * WT_NC * c.H + (size_t)WT_NC * 256 + (size_t)WT_NC * round_up(c.S, WT_NC) + WT_NC);
}
static size_t tile_exchange_bytes(const ttsc_wavernn* w, int B) {
    return tile_exchange_granules(w, (int)ceil_div(B, WT_NC)) * 8 + 256;
}

static int tile_pack(ttsc_wavernn* w) {
    const auto& c = w->cfg;
    const int NC = WT_NC, H = c.H, UPW = H / NC, R3 = 3 * UPW, I0 = w->in0, I0P = (int)round_up(I0, 4), S = c.S, SP = (int)round_up(S, NC), SR = SP / NC, PR = 256 / NC;
    std::vector<float> whh((size_t)NC * H * R3, 0.f), wih((size_t)NC * I0P * R3, 0.f), bih((size_t)NC * R3), bhh((size_t)NC * R3);
    // the output slice is padded to PR = 32 rows (zero rows beyond SR), so that both resident slices share one layout
    std::vector<float> wpre((size_t)NC * H * PR), bpre((size_t)NC * PR), wout((size_t)NC * 256 * PR, 0.f), bout((size_t)NC * PR, 0.f);
    for (int m = 0; m < NC; ++m) {
        for (int q = 0; q < 3; ++q)
            for (int j = 0; j < UPW; ++j) {
                const int row = q * H + m * UPW + j, lr = q * UPW + j;
                for (int k = 0; k < H; ++k) whh[(size_t)m * H * R3 + ((size_t)(k >> 2) * R3 + lr) * 4 + (k & 3)] = w->h_whh0[(size_t)row * H + k];
                for (int k = 0; k < I0; ++k) wih[(size_t)m * I0P * R3 + ((size_t)(k >> 2) * R3 + lr) * 4 + (k & 3)] = w->h_wih0[(size_t)row * I0 + k];
                bih[(size_t)m * R3 + lr] = w->h_bih0[row];
                bhh[(size_t)m * R3 + lr] = w->h_bhh0[row];
            }
        for (int r = 0; r < PR; ++r) {
            const int row = m * PR + r;
            for (int k = 0; k < H; ++k) wpre[(size_t)m * H * PR + ((size_t)(k >> 2) * PR + r) * 4 + (k & 3)] = w->h_wpre[(size_t)row * H + k];
            bpre[(size_t)m * PR + r] = w->h_bpre[row];
        }
        for (int r = 0; r < SR; ++r) {
            const int row = m * SR + r;
            if (row >= S) continue;   // padding rows of a continuous head stay zero
            for (int k = 0; k < 256; ++k) wout[(size_t)m * 256 * PR + ((size_t)(k >> 2) * PR + r) * 4 + (k & 3)] = w->h_wout[(size_t)row * 256 + k];
            bout[(size_t)m * PR + r] = w->h_bout[row];
        }
    }
    int rc;
    if ((rc = upload(&w->q_whh, whh.data(), whh.size()))) return rc;
    if ((rc = upload(&w->q_wih, wih.data(), wih.size()))) return rc;
    if ((rc = upload(&w->q_bih, bih.data(), bih.size()))) return rc;
    if ((rc = upload(&w->q_bhh, bhh.data(), bhh.size()))) return rc;
    if ((rc = upload(&w->q_wpre, wpre.data(), wpre.size()))) return rc;
    if ((rc = upload(&w->q_bpre, bpre.data(), bpre.size()))) return rc;
    if ((rc = upload(&w->q_wout, wout.data(), wout.size()))) return rc;
    if ((rc = upload(&w->q_bout, bout.data(), bout.size()))) return rc;
    if (c.num_layers == 2) {   // second GRU layer: input = h1 (K = H), same member slices and packing
        std::vector<float> whh2((size_t)NC * H * R3), wih2((size_t)NC * H * R3), bih2((size_t)NC * R3), bhh2((size_t)NC * R3);
        for (int m = 0; m < NC; ++m)
            for (int q = 0; q < 3; ++q)
                for (int j = 0; j < UPW; ++j) {
                    const int row = q * H + m * UPW + j, lr = q * UPW + j;
                    for (int k = 0; k < H; ++k) {
                        whh2[(size_t)m * H * R3 + ((size_t)(k >> 2) * R3 + lr) * 4 + (k & 3)] = w->h_whh1[(size_t)row * H + k];
                        wih2[(size_t)m * H * R3 + ((size_t)(k >> 2) * R3 + lr) * 4 + (k & 3)] = w->h_wih1[(size_t)row * H + k];
                    }
                    bih2[(size_t)m * R3 + lr] = w->h_bih1[row];
                    bhh2[(size_t)m * R3 + lr] = w->h_bhh1[row];
                }
        if ((rc = upload(&w->q_whh2, whh2.data(), whh2.size()))) return rc;
        if ((rc = upload(&w->q_wih2, wih2.data(), wih2.size()))) return rc;
        if ((rc = upload(&w->q_bih2, bih2.data(), bih2.size()))) return rc;
        if ((rc = upload(&w->q_bhh2, bhh2.data(), bhh2.size()))) return rc;
    }
    w->tile_dirty = false;
    return TTSC_OK;
}

extern "C" int64_t ttsc_wavernn_out_len(const ttsc_wavernn* w, int64_t T, int64_t Tl) {
    if (!w) return TTSC_EINVAL;
    int64_t L = T * w->cfg.upsample;
    if (w->cfg.use_lowres) {
        const int64_t l2 = Tl * w->cfg.upsample_low;
        if (l2 < L) L = l2;
    }
    return L;
}

static size_t cond_bytes(const ttsc_wavernn* w, int32_t B, int64_t Tl) {
    if (!w->cfg.use_lowres) return 256;
    return (size_t)round_up((int64_t)(((size_t)B * Tl * w->cfg.upsample_low + 2 * (size_t)B * 20 * Tl) * sizeof(float)) + 256, 256);
}

extern "C" size_t ttsc_wavernn_workspace_bytes(const ttsc_wavernn* w, int32_t B, int64_t T, int64_t Tl) {
    if (!w) return 0;
    return cond_bytes(w, B, Tl) + tile_exchange_bytes(w, B);
}

extern "C" int ttsc_wavernn_decode(ttsc_wavernn* w, const float* mel, const float* x_low, int32_t B, int64_t T, int64_t Tl,
                                   int32_t mode, const float* noise, uint64_t seed, const float* forced_x, uint8_t* idx,
                                   float* wav, float* logits, void* ws, size_t ws_bytes, void* stream) {
    TTSC_REQUIRE(w && mel && idx && wav, "ttsc_wavernn_decode: null argument");
    TTSC_REQUIRE(B > 0 && T > 0, "ttsc_wavernn_decode: bad B/T");
    TTSC_REQUIRE(mode >= 0 && mode <= 2, "ttsc_wavernn_decode: mode must be 0 (argmax), 1 (injected noise) or 2 (philox)");
    TTSC_REQUIRE(mode != TTSC_WR_MODE_NOISE || noise, "ttsc_wavernn_decode: mode=noise needs a noise buffer");
    const auto& c = w->cfg;
    for (int l = 0; l < c.num_layers; ++l)
        if (!(w->wt_ih[l] && w->wt_hh[l] && w->b_ih[l] && w->b_hh[l])) {
            set_error("ttsc_wavernn_decode: weights of _rnns.%d missing", l);
            return TTSC_ESTATE;
        }
    if (!(w->wt_pre && w->b_pre && w->wt_out && w->b_out)) {
        set_error("ttsc_wavernn_decode: _preoutput/_output weights missing");
        return TTSC_ESTATE;
    }
    hipStream_t s = (hipStream_t)stream;
    WrArgs a;
    memset(&a, 0, sizeof(a));
    if (c.use_lowres) {
        TTSC_REQUIRE(x_low && Tl > 0, "ttsc_wavernn_decode: the high-res network needs x_low");
        for (int i = 0; i < 3; ++i)
            if (!(w->lc_w[i] && w->lc_b[i])) {
                set_error("ttsc_wavernn_decode: _lowres_conv.%d weights missing", i);
                return TTSC_ESTATE;
            }
        const size_t need = ttsc_wavernn_workspace_bytes(w, B, T, Tl);
        if (!ws || ws_bytes < need) {
            set_error("ttsc_wavernn_decode: workspace %zu < required %zu bytes", ws_bytes, need);
            return TTSC_ENOMEM;
        }
        float* interp = (float*)ws;
        float* fa = interp + (size_t)B * Tl * c.upsample_low;
        float* fb = fa + (size_t)B * 20 * Tl;
        hipLaunchKernelGGL(wr_cond_kernel, dim3(B), dim3(256), 0, s, x_low, w->lc_w[0], w->lc_b[0], w->lc_w[1], w->lc_b[1],
                           w->lc_w[2], w->lc_b[2], interp, fa, fb, (int)Tl, c.upsample_low);
        a.interp = interp;
        a.feats = fa;
    }
    a.mel = mel;
    for (int l = 0; l < c.num_layers; ++l) {
        a.wt_ih[l] = w->wt_ih[l];
        a.wt_hh[l] = w->wt_hh[l];
        a.b_ih[l] = w->b_ih[l];
        a.b_hh[l] = w->b_hh[l];
    }
    a.wt_pre = w->wt_pre;
    a.b_pre = w->b_pre;
    a.wt_out = w->wt_out;
    a.b_out = w->b_out;
    a.lut = w->lut;
    a.noise = noise;
    a.forced_x = forced_x;
    a.out_idx = idx;
    a.out_wav = wav;
    a.out_logits = logits;
    a.B = B;
    a.T = (int)T;
    a.Tl = (int)Tl;
    a.H = c.H;
    a.NL = c.num_layers;
    a.use_lowres = c.use_lowres;
    a.up = c.upsample;
    a.up_low = c.upsample_low;
    a.S = c.S;
    a.n_mel = c.n_mel;
    a.out_kind = c.out_kind;
    a.mode = mode;
    {
        const int64_t L64 = ttsc_wavernn_out_len(w, T, Tl);
        TTSC_REQUIRE(L64 < (1ll << 31), "ttsc_wavernn_decode: more than 2^31 samples per utterance");
        a.L = (int)L64;
    }
    a.seed = seed;
    TTSC_REQUIRE(a.L > 0, "ttsc_wavernn_decode: nothing to decode (L=%d)", a.L);
    if (tile_supported(w, B) && a.L < (1 << 24)) {   // (candidate granules carry a 24-bit step tag)
        const int NC = WT_NC, BU = WT_NC;
        if (w->tile_dirty) {
            int prc = tile_pack(w);
            if (prc) return prc;
        }
        const size_t need_all = ttsc_wavernn_workspace_bytes(w, B, T, Tl);
        if (!ws || ws_bytes < need_all) {
            set_error("ttsc_wavernn_decode: workspace %zu < required %zu bytes", ws_bytes, need_all);
            return TTSC_ENOMEM;
        }
        const int G = (int)ceil_div(B, BU);
        char* xbase = (char*)ws + cond_bytes(w, B, Tl);
        WtArgs qa;
        memset(&qa, 0, sizeof(qa));
        qa.mel = mel; qa.interp = a.interp; qa.feats = a.feats;
        qa.whh = w->q_whh; qa.wih = w->q_wih; qa.bih = w->q_bih; qa.bhh = w->q_bhh;
        qa.wpre = w->q_wpre; qa.bpre = w->q_bpre; qa.wout = w->q_wout; qa.bout = w->q_bout;
        qa.whh2 = w->q_whh2; qa.wih2 = w->q_wih2; qa.bih2 = w->q_bih2; qa.bhh2 = w->q_bhh2; qa.NL = c.num_layers;
        qa.lut = w->lut; qa.noise = noise; qa.forced_x = forced_x; qa.out_idx = idx; qa.out_wav = wav; qa.out_logits = logits;
        u64* f = (u64*)xbase;
        qa.xh = f; f += (size_t)G * 2 * BU * c.H;
        if (c.num_layers == 2) { qa.xh2 = f; f += (size_t)G * 2 * BU * c.H; }
        qa.xpre = f; f += (size_t)G * 2 * BU * 256;
        qa.xlog = f; f += (size_t)G * 2 * BU * round_up(c.S, NC);
        qa.xlx = f; f += (size_t)G * 2 * BU;
        qa.abort_word = (unsigned*)f;
        qa.B = B; qa.T = (int)T; qa.Tl = (int)Tl; qa.H = c.H; qa.UPW = c.H / NC; qa.I0 = w->in0; qa.I0P = (int)round_up(w->in0, 4);
        qa.use_lowres = c.use_lowres; qa.up = c.upsample; qa.up_low = c.upsample_low; qa.S = c.S; qa.SP = (int)round_up(c.S, NC); qa.SR = qa.SP / NC;
        qa.n_mel = c.n_mel; qa.out_kind = c.out_kind; qa.mode = mode; qa.L = a.L; qa.G = G; qa.seed = seed;
        qa.GP = (int)round_up(G, WT_XCDS);
        // all tags (and the abort word) start at zero; the first step carries tag 1
        TTSC_HIP_CHECK(hipMemsetAsync(xbase, 0, tile_exchange_granules(w, G) * 8 + 64, s));
        const size_t lds = tile_lds_bytes(w);
        if (lds > 64 * 1024) {
            // full 160 KiB once per (device, kernel): a later model with a larger H, or a second device, needs no re-arming
            if (int rc = ensure_full_lds((const void*)wr_tile_kernel<false, false>)) return rc;
            if (int rc = ensure_full_lds((const void*)wr_tile_kernel<true, false>)) return rc;
            if (int rc = ensure_full_lds((const void*)wr_tile_kernel<false, true>)) return rc;
            if (int rc = ensure_full_lds((const void*)wr_tile_kernel<true, true>)) return rc;
        }
#ifdef TTSC_ABLATE
        unsigned long long* prof_dev = nullptr;
        if (getenv("TTSC_WT_PROF")) {
            TTSC_HIP_CHECK(hipMalloc((void**)&prof_dev, (size_t)qa.GP * NC * 16 * sizeof(unsigned long long)));
            TTSC_HIP_CHECK(hipMemsetAsync(prof_dev, 0, (size_t)qa.GP * NC * 16 * sizeof(unsigned long long), s));
            qa.prof = prof_dev;
        }
#endif
        const dim3 tg(qa.GP * NC), tb(WT_THREADS);
        if (c.num_layers == 2) {
            if (c.out_kind >= 2)
                hipLaunchKernelGGL((wr_tile_kernel<true, true>), tg, tb, lds, s, qa);
            else
                hipLaunchKernelGGL((wr_tile_kernel<false, true>), tg, tb, lds, s, qa);
        } else if (c.out_kind >= 2) {
            hipLaunchKernelGGL((wr_tile_kernel<true, false>), tg, tb, lds, s, qa);
        } else {
            hipLaunchKernelGGL((wr_tile_kernel<false, false>), tg, tb, lds, s, qa);
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) {
            set_error("wr_tile_kernel launch failed: %s", hipGetErrorString(e));
            return TTSC_EHIP;
        }
#ifdef TTSC_ABLATE
        if (prof_dev) {   // where a step of the tile kernel goes: thread 0 of every workgroup, mean, microseconds per step
            TTSC_HIP_CHECK(hipStreamSynchronize(s));
            std::vector<unsigned long long> hp((size_t)qa.GP * NC * 16);
            TTSC_HIP_CHECK(hipMemcpy(hp.data(), prof_dev, hp.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            hipFree(prof_dev);
            static const char* nm[14] = {"loop", "ih-prefix+join", "lastx-handoff+gate", "h-handoff+stage", "tail:noise", "tail:wait for the output quarters",
                                         "tail:add quarters+candidates", "tail:candidate-handoff+sample",
                                         "w4: rest of the step (join, gates, h hand-off)", "w4: pre-output quarter chain", "w4: wait for the 4 partials",
                                         "w4: add+tanh+publish", "w4: gather its quarter of pre", "w4: output quarter chain+signal"};
            double tot = 0;
            for (int i = 0; i < 14; ++i) {
                if (i == 8) {
                    fprintf(stderr, "wt-prof %-28s %7.2f us/step (B=%d, L=%d)\n", "total (wave 0)", tot, B, qa.L);
                    tot = 0;
                }
                double sum = 0;
                int n = 0;
                for (size_t wg = 0; wg < (size_t)qa.GP * NC; ++wg)
                    if (hp[wg * 16 + 3]) { sum += (double)hp[wg * 16 + i]; ++n; }
                const double us = n ? sum / n / 100.0 / qa.L : 0.0;
                tot += us;
                fprintf(stderr, "wt-prof %-46s %7.2f us/step\n", nm[i], us);
            }
            fprintf(stderr, "wt-prof %-28s %7.2f us/step (B=%d, L=%d)\n", "total (wave 4)", tot, B, qa.L);
        }
#endif
        w->last_abort_word = qa.abort_word;
        w->last_kind = 2;
        return TTSC_OK;
    }
    w->last_abort_word = nullptr;
    w->last_kind = 0;
    // Utterances per workgroup (BT).  Measured on MI355X (H=512, 1 layer): one utterance per workgroup is fastest
    // per step (60 us) while the batch fits the 256 CUs; beyond that a tile of 2/4 utterances shares one weight
    // stream (74 / 98 us per step) and raises throughput (B=1024, BT=4: 9.6 M samples/s).
    int bt = 1;
    if (B > 512) bt = 4; else if (B > 256) bt = 2;
    if (const char* ev = getenv("TTSC_WR_BT")) {
        const int v = atoi(ev);
        if (v == 1 || v == 2 || v == 4 || v == 8) bt = v;
    }
    const size_t lds = ((size_t)2 * c.num_layers * bt * c.H + (size_t)bt * 256 + (size_t)bt * c.S + (size_t)bt * 128 + 16) * sizeof(float);
    dim3 grid((unsigned)ceil_div(B, bt));
    switch (bt) {
        case 1: hipLaunchKernelGGL(wr_decode_kernel<1>, grid, dim3(WR_THREADS), lds, s, a); break;
        case 2: hipLaunchKernelGGL(wr_decode_kernel<2>, grid, dim3(WR_THREADS), lds, s, a); break;
        case 4: hipLaunchKernelGGL(wr_decode_kernel<4>, grid, dim3(WR_THREADS), lds, s, a); break;
        default: hipLaunchKernelGGL(wr_decode_kernel<8>, grid, dim3(WR_THREADS), lds, s, a); break;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("wr_decode_kernel launch failed: %s", hipGetErrorString(e));
        return TTSC_EHIP;
    }
    return TTSC_OK;
}


// After the stream has executed the last decode: -1 = streaming kernel (nothing to check), 2 = tile
// kernel ok, 1 = a multi-workgroup kernel aborted on a hand-off timeout (results invalid).  Synchronises the stream.
extern "C" int ttsc_wavernn_last_status(ttsc_wavernn* w, void* stream) {
    if (!w) return TTSC_EINVAL;
    if (!w->last_abort_word) return -1;
    unsigned v = 0;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return TTSC_EHIP;
    if (hipMemcpy(&v, w->last_abort_word, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return TTSC_EHIP;
    if (v) set_error("WaveRNN multi-workgroup kernel: inter-workgroup hand-off timed out (not all members resident?)");
    return v ? 1 : (w->last_kind == 2 ? 2 : 0);
}
