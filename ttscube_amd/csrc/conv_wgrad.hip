// Weight gradient of Conv1d / ConvTranspose1d (training) as an fp32-MFMA correlation — gfx950.
//
//   G[a, b, j] += sum_n sum_t  P[n, a, t] * lrelu(q_scale * Q[n, b, t + base + j*step], q_slope)
//
// Conv1d (weight [Co,Ci,K], dilation d, padding p):  P = dL/dy [N,Co,Lout], Q = the layer input x [N,Ci,Lin] read through
// the same leaky-relu prologue as the forward, base = -p, step = d  ->  G = dL/dW.  ConvTranspose1d is expressed in the
// same form on the phase-de-interleaved output gradient (see hifigan/autograd.py).
//
// GEMM view: M = a (32 rows per wave), N = b (32 columns per wave), K = the N*L positions — a reduction that is 10^5..10^6
// long while M x N is at most 512 x 256, so the parallelism is in K: every WAVE is an independent worker that owns one
// 32 x 32 (a, b) tile with ALL taps (J accumulators; the P fragment is shared by the J taps) over a contiguous run of
// 64-position chunks of one batch item.  The four waves of a workgroup are summed through LDS and the workgroup's partial
// tile goes to a workspace [split][tap][A][B] with coalesced stores; a second kernel adds the splits in a fixed order
// (deterministic; fp32 atomics into G measured 10x slower: 17 M device-scope atomics per layer).  Chunks are staged
// through a per-wave LDS tile (row pitch = 2 mod 64 words: the 32-row x 2-position fragment reads are conflict-free) with
// the global loads of the next chunk issued before the MFMA loop of the current one (explicit register double buffer).
#include <algorithm>

#include "common.hpp"
#include "conv_kernels.hpp"

namespace ttsc {

struct WgArgs {
    const float* P;
    const float* Q;
    float* part;       // workspace [splits = gridDim.x][Jtot][A][Bc]
    int N, A, Bc, LP, LQ;
    int J, Jtot, j0;   // taps handled by this launch: j0 .. j0+J-1 of Jtot
    int base, step;
    float q_scale, q_slope;
    int chunks;        // 64-position chunks per batch item = ceil(LP / 64)
    int CH;            // chunks per wave
    int groups;        // position groups per batch item = ceil(chunks / CH)
    int minoff, span;  // Q window of one chunk: positions t0 + minoff .. t0 + minoff + 64 + span
    int cg_n, Ag, Btot; // grouped convolution: cg_n conv groups of Ag rows (A = cg_n * Ag) x Bc columns each; Q has Btot = cg_n * Bc channels
};

constexpr int WG_TK = 64;
constexpr int WG_PP = 66;    // P tile pitch (words)
constexpr int WG_QP = 130;   // Q tile pitch: 64 + span (<= 64) + pad

template <int JT>
__global__ __launch_bounds__(256, 1) void conv_wgrad_kernel(WgArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    float* pl = sm + wave * (32 * WG_PP + 32 * WG_QP);
    float* ql = pl + 32 * WG_PP;
    // tile -> (conv group, row tile, column tile): rows a0.. of the group's Ag rows, columns b0.. of ITS Bc input channels
    const int tb_n = (a.Bc + 31) >> 5, ta_n = (a.Ag + 31) >> 5;
    const int cg = blockIdx.y / (ta_n * tb_n), ti = blockIdx.y % (ta_n * tb_n);
    const int a0 = cg * a.Ag + (ti / tb_n) * 32, a_end = (cg + 1) * a.Ag, b0 = (ti % tb_n) * 32;
    const int grp = blockIdx.x * 4 + wave;            // position group of this wave
    const bool live = grp < a.N * a.groups;
    const int n = live ? grp / a.groups : 0;
    const int c_beg = live ? (grp % a.groups) * a.CH : 0;
    const int c_end = live ? min(c_beg + a.CH, a.chunks) : 0;
    const bool two = a.span > 0;                      // Q window wider than 64 positions

    f32x16 acc[JT];
#pragma unroll
    for (int j = 0; j < JT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    const float* Pn = a.P + (size_t)n * a.A * a.LP;
    const float* Qn = a.Q + ((size_t)n * a.Btot + (size_t)cg * a.Bc) * a.LQ;
    // `load` only ISSUES the global loads (clamped addresses, no use of the values): any arithmetic on a loaded value placed
    // here makes hipcc wait for that load before issuing the next ones.  Masking and the leaky-relu prologue happen in
    // `commit`, one MFMA phase later.
    float pr[32], q0r[32], q1r[32];
    auto load = [&](int c) __attribute__((always_inline)) {
        const int t0 = c * WG_TK;
        const int tp = min(t0 + lane, a.LP - 1);
#pragma unroll
        for (int r = 0; r < 32; ++r) pr[r] = Pn[(unsigned)(min(a0 + r, a_end - 1) * a.LP + tp)];
        const int q_a = min(max(t0 + a.minoff + lane, 0), a.LQ - 1);
#pragma unroll
        for (int r = 0; r < 32; ++r) q0r[r] = Qn[(unsigned)(min(b0 + r, a.Bc - 1) * a.LQ + q_a)];
        if (two) {
            const int q_b = min(max(t0 + a.minoff + lane + 64, 0), a.LQ - 1);
#pragma unroll
            for (int r = 0; r < 32; ++r) q1r[r] = Qn[(unsigned)(min(b0 + r, a.Bc - 1) * a.LQ + q_b)];
        }
    };
    auto commit = [&](int c) __attribute__((always_inline)) {
        const int t0 = c * WG_TK;
        const bool pok = t0 + lane < a.LP;
#pragma unroll
        for (int r = 0; r < 32; ++r) pl[r * WG_PP + lane] = (pok && a0 + r < a_end) ? pr[r] : 0.f;
        const int q_a = t0 + a.minoff + lane;
        const bool qa_ok = q_a >= 0 && q_a < a.LQ;
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            float v = q0r[r] * a.q_scale;
            v = v > 0.f ? v : v * a.q_slope;
            ql[r * WG_QP + lane] = (qa_ok && b0 + r < a.Bc) ? v : 0.f;
        }
        if (two) {
            const int q_b = q_a + 64;
            const bool qb_ok = q_b >= 0 && q_b < a.LQ && lane < a.span;
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                float v = q1r[r] * a.q_scale;
                v = v > 0.f ? v : v * a.q_slope;
                ql[r * WG_QP + 64 + lane] = (qb_ok && b0 + r < a.Bc) ? v : 0.f;
            }
        }
    };

    const float* qj[JT];   // per-tap fragment base inside the Q tile
#pragma unroll
    for (int j = 0; j < JT; ++j) qj[j] = ql + l31 * WG_QP + half + (a.base + (a.j0 + j) * a.step - a.minoff);

    // every wave of the workgroup runs the same number of iterations (workgroup barriers inside); idle ones add zeros
    if (c_beg < c_end) load(c_beg);
    for (int it = 0; it < a.CH; ++it) {
        const int c = c_beg + it;
        const bool on = c < c_end;
        if (on) commit(c);
        __syncthreads();
        if (c + 1 < c_end) load(c + 1);
        if (on) {
            const float* pa = pl + l31 * WG_PP + half;
#pragma unroll 2
            for (int kk0 = 0; kk0 < WG_TK / 2; kk0 += 4) {
                // all fragment reads of four k-steps first (the LDS returns them in order, the MFMAs start as they land);
                // a read placed right before its MFMA costs one LDS round trip per MFMA
                float af[4], bf[4][JT];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    af[u] = pa[2 * (kk0 + u)];
#pragma unroll
                    for (int j = 0; j < JT; ++j) bf[u][j] = qj[j][2 * (kk0 + u)];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < JT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[u], bf[u][j], acc[j], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // ---- sum the four waves through LDS (staging tiles are dead), one tap at a time; coalesced partial stores ---------
    // C/D layout: column (= b) = lane & 31, row (= a) = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    float* red = sm;   // [4 waves][16 r][64 lanes]
    const int tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < JT; ++j) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave * 1024 + r * 64 + lane] = acc[j][r];
        __syncthreads();
        float* dst = a.part + ((size_t)blockIdx.x * a.Jtot + a.j0 + j) * a.A * a.Bc;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i;
            const float v = (red[e] + red[1024 + e]) + (red[2048 + e] + red[3072 + e]);
            const int r = e >> 6, ln = e & 63;
            const int arow = a0 + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5), bcol = b0 + (ln & 31);
            if (arow < a_end && bcol < a.Bc) dst[(size_t)arow * a.Bc + bcol] = v;
        }
    }
}

// G[a][b][j] = sum over splits of part[sp][j][a][b]   (fixed order).  Block = 32 consecutive elements x 8 split lanes: lane q adds
// the splits q, q+8, ... with four loads in flight, then the 8 lanes are added in index order through LDS.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ G, int splits, int J, long AB,
                                                           const float* __restrict__ bias_part = nullptr, float* __restrict__ db = nullptr, int A = 0,
                                                           int bias_S = 0, unsigned bias_b0 = 0) {
    __shared__ float red[8][32];
    if (bias_part && blockIdx.x >= bias_b0) {   // surplus workgroups: db[c] = the bias_S slices of row c in index order (bias_grad_kernel's second stage)
        const int c = (int)(blockIdx.x - bias_b0) * 256 + threadIdx.x;
        if (c < A) {
            float tot = 0.f;
            for (int k = 0; k < bias_S; ++k) tot += bias_part[(size_t)c * bias_S + k];
            db[c] = tot;
        }
        return;
    }
    const long total = (long)J * AB;
    const int ex = threadIdx.x & 31, q = threadIdx.x >> 5;
    const long i = (long)blockIdx.x * 32 + ex;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < total) {
        const float* p = part + i;
        int sp = q;
        for (; sp + 24 < splits; sp += 32) {
            s0 += p[(size_t)sp * total];
            s1 += p[(size_t)(sp + 8) * total];
            s2 += p[(size_t)(sp + 16) * total];
            s3 += p[(size_t)(sp + 24) * total];
        }
        for (; sp < splits; sp += 8) s0 += p[(size_t)sp * total];
    }
    red[q][ex] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (q == 0 && i < total) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k][ex];
        const long j = i / AB, ab = i - j * AB;
        G[ab * J + j] = t;
    }
}

template <int JT>
static int launch_wgrad(const WgArgs& a, hipStream_t s) {
    const int tiles = a.cg_n * ((a.Ag + 31) / 32) * ((a.Bc + 31) / 32);
    const int wgs_x = (a.N * a.groups + 3) / 4;
    const size_t lds = (size_t)4 * (32 * WG_PP + 32 * WG_QP) * sizeof(float);
    if (int rc = ensure_full_lds(reinterpret_cast<const void*>(conv_wgrad_kernel<JT>))) return rc;   // once per (device, kernel)
    hipLaunchKernelGGL(conv_wgrad_kernel<JT>, dim3(wgs_x, tiles), dim3(256), lds, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("conv_wgrad_kernel launch failed: %s", hipGetErrorString(e));
        return TTSC_EHIP;
    }
    return TTSC_OK;
}


// ---------------------------------------------------------------------------------------------------------------
// Split-precision weight gradient (the dense layers of the training step: discriminator 512..1024-channel layers, generator ResBlocks).
// Same correlation as above, but as a proper GEMM tile on v_mfma_f32_32x32x16_f16: a workgroup owns 128 rows (a) x 64 columns (b) with all
// taps of the launch (2 x 2 waves, 64 x 32 each: 2 x JT accumulator tiles) and walks 64-position chunks of (batch item, time); both
// operands are ACTIVATIONS here, so both are scaled by a device-side maximum (amax[0] = max |Q|, amax[1] = max |P|), split into fp16
// hi + lo while they are staged, and multiplied as hi*hi + hi*lo + lo*hi.  LDS rows are [channel][position] in fp16: the P fragment of a lane
// is 8 consecutive positions at a 16-byte aligned address; the Q fragment of tap j is the same 8 positions shifted by the tap offset — a
// 2-byte aligned ds_read_b128 (gfx950 reads LDS unaligned).  The fp32 kernel gives every WAVE its own 32 x 32 tile, i.e. re-reads P and Q
// from L2 32 times each for a 1024 x 1024 layer (467 us); here the operands are read 16 / 8 times and the products run at the fp16 rate.
struct WgsArgs {
    const float* P;
    const float* Q;
    float* part;         // [splits = gridDim.x][Jtot][A][Bc]
    const float* amax_q;   // *amax_q = max |Q|, *amax_p = max |P|
    const float* amax_p;
    int N, A, Bc, LP, LQ;
    int Jtot, j0, base, step;
    float q_scale, q_slope;
    int chunks, CH, items;   // 64-position chunks per batch item; work items (batch item, chunk) per workgroup; N * chunks
    int minoff, span;
    // bias gradient riding along (round 6): the workgroups with blockIdx.y >= bias_y0 are not tiles of the weight gradient — workgroup
    // k = (blockIdx.y - bias_y0) * gridDim.x + blockIdx.x < A * bias_S sums slice k % bias_S of row k / bias_S of P over all batch items into
    // bias_part[row][slice] (train_ops.hip::bias_grad_kernel's first stage, same order, same bits); wgrad_reduce_kernel's surplus workgroups add the slices.
    float* bias_part;
    int bias_y0, bias_S;
};
constexpr int WS_PP = 72, WS_QP = 136;   // row pitches in halves: 144 / 272 bytes = 16 mod 128 (eight rows cover all banks), 16-byte multiples
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
struct __attribute__((packed, aligned(2))) half8_u {
    half8 v;
};
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) f4u {   // four consecutive floats at a 4-byte aligned address (one global_load_dwordx4)
    float v[4];
};

template <int JT>
__global__ __launch_bounds__(256, 2) void wgrad_f16x3_kernel(WgsArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smraw[];
    _Float16* Ph = reinterpret_cast<_Float16*>(smraw);   // [128][WS_PP]
    _Float16* Pl = Ph + 128 * WS_PP;
    _Float16* Qh = Pl + 128 * WS_PP;                     // [64][WS_QP]
    _Float16* Ql = Qh + 64 * WS_QP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, half = lane >> 5, l31 = lane & 31;
    if (a.bias_part && (int)blockIdx.y >= a.bias_y0) {   // bias-gradient slice (see WgsArgs::bias_part)
        const long k = (long)((int)blockIdx.y - a.bias_y0) * gridDim.x + blockIdx.x;
        if (k >= (long)a.A * a.bias_S) return;
        const int c = (int)(k / a.bias_S), sl = (int)(k - (long)c * a.bias_S);
        const int per = (a.LP + a.bias_S - 1) / a.bias_S;
        const int t0 = sl * per, t1 = min(t0 + per, a.LP);
        const int w = t1 - t0, n = a.N * w;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        auto at = [&](int i) {
            const int b = i / w, t = t0 + (i - b * w);
            return a.P[((size_t)b * a.A + c) * a.LP + t];
        };
        int i = tid;
        for (; i + 768 < n; i += 1024) {
            const float v0 = at(i), v1 = at(i + 256), v2 = at(i + 512), v3 = at(i + 768);
            s0 += v0;
            s1 += v1;
            s2 += v2;
            s3 += v3;
        }
        for (; i < n; i += 256) s0 += at(i);
        float v = (s0 + s1) + (s2 + s3);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        float* red = reinterpret_cast<float*>(smraw);
        if (lane == 0) red[wave] = v;
        __syncthreads();
        if (tid == 0) a.bias_part[(size_t)c * a.bias_S + sl] = (((0.f + red[0]) + red[1]) + red[2]) + red[3];
        return;
    }
    const int tb_n = (a.Bc + 63) >> 6;
    const int a0 = (blockIdx.y / tb_n) * 128, b0 = (blockIdx.y % tb_n) * 64;
    const int it_beg = blockIdx.x * a.CH, it_end = min(it_beg + a.CH, a.items);
    // power-of-two ranges from the device-side maxima (same rule as the convolution: conv_kernels.hpp)
    const float sq = pow2_to(*a.amax_q * a.q_scale, SPLIT_X_TARGET), sp = pow2_to(*a.amax_p, SPLIT_X_TARGET);
    const float qs = a.q_scale * sq;

    f32x16 acc[2][JT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < JT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int it = it_beg; it < it_end; ++it) {
        const int n = it / a.chunks, t0 = (it - n * a.chunks) * 64;
        // (opaque copies of the row lengths: otherwise hipcc hoists the 48 clamped row offsets out of this loop and keeps them in VGPRs
        // next to the 160 accumulator registers)
        int lp = a.LP, lq = a.LQ, a0v = a0 + wave * 32, b0v = b0 + wave * 16;
        asm volatile("" : "+s"(lp), "+s"(lq), "+v"(a0v), "+v"(b0v));
        __syncthreads();   // the fragment reads of the previous chunk are done
        // A chunk whose positions and Q window lie inside the rows (every chunk but the first / last of a sequence) is staged FOUR positions per lane:
        // 16-byte loads (a quarter-wave covers the 64 positions of a row, the wave four rows per instruction) and 8-byte LDS writes — per thread
        // 16 loads + 32 writes instead of 64 + 128.  Measured (round 6, tools/probes/wgrad_time.py, profiles/r06_wgrad_staging.log): with only a
        // quarter of the 2-byte staging writes issued the launches ran 11 - 36 % faster — the LDS instruction count, not the matrix pipe, bounded them.
        const int qlo = t0 + a.minoff;
        const bool interior = t0 + 64 <= a.LP && qlo >= 0 && qlo + 64 + ((a.span + 3) & ~3) <= a.LQ;
        if (interior) {
            const int sub = lane >> 4, q4 = 4 * (lane & 15);
            {
                const float* Pn = a.P + (size_t)n * a.A * a.LP + t0 + q4;
                f4u pv[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int ar = a0v + 4 * k + sub;
                    pv[k] = *reinterpret_cast<const f4u*>(Pn + (unsigned)((ar < a.A ? ar : a.A - 1) * lp));
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int row = wave * 32 + 4 * k + sub;
                    const bool rok = a0 + row < a.A;
                    half4 h, l;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = rok ? pv[k].v[e] * sp : 0.f;
                        h[e] = (_Float16)v;
                        l[e] = (_Float16)(v - (float)h[e]);
                    }
                    *reinterpret_cast<half4*>(Ph + row * WS_PP + q4) = h;
                    *reinterpret_cast<half4*>(Pl + row * WS_PP + q4) = l;
                }
            }
            {
                const float* Qn = a.Q + (size_t)n * a.Bc * a.LQ + qlo + q4;
                const bool two = a.span > 0;
#pragma unroll
                for (int ph = 0; ph < 2; ++ph) {
                    if (ph == 1 && !two) break;
                    const bool qok = ph == 0 || q4 < a.span;   // (second part: only the quads the taps reach)
                    f4u qv[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int br = b0v + 4 * k + sub;
                        qv[k] = *reinterpret_cast<const f4u*>(Qn + (unsigned)((br < a.Bc ? br : a.Bc - 1) * lq) + (qok ? ph * 64 : 0));
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int row = wave * 16 + 4 * k + sub;
                        const bool rok = b0 + row < a.Bc;
                        half4 h, l;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float v = qv[k].v[e] * qs;
                            v = fmaxf(v, v * a.q_slope);
                            v = rok ? v : 0.f;
                            h[e] = (_Float16)v;
                            l[e] = (_Float16)(v - (float)h[e]);
                        }
                        if (qok) {
                            *reinterpret_cast<half4*>(Qh + row * WS_QP + ph * 64 + q4) = h;
                            *reinterpret_cast<half4*>(Ql + row * WS_QP + ph * 64 + q4) = l;
                        }
                    }
                }
            }
        } else {
        // Staging of an edge chunk: lane = position (coalesced 256-byte row segments).  Loads are ISSUED in batches of 16 into registers with clamped
        // addresses and no use of the values (hipcc waits for a load right before its first use: load-convert-store per element is one L2 round trip
        // each), then masked, scaled, split and written as fp16.
        {   // P: wave w stages rows 32 w .. 32 w + 31
            const float* Pn = a.P + (size_t)n * a.A * a.LP;
            const int t = t0 + lane;
            const bool tok = t < a.LP;
            const unsigned tc = (unsigned)(tok ? t : a.LP - 1);
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) {
                float pr[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ar = a0v + pb * 16 + r;
                    pr[r] = Pn[(unsigned)((ar < a.A ? ar : a.A - 1) * lp) + tc];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wave * 32 + pb * 16 + r;
                    const float v = (tok && a0 + row < a.A) ? pr[r] * sp : 0.f;
                    const _Float16 h = (_Float16)v;
                    Ph[row * WS_PP + lane] = h;
                    Pl[row * WS_PP + lane] = (_Float16)(v - (float)h);
                }
            }
        }
        {   // Q: wave w stages rows 16 w .. 16 w + 15: columns 0..63 and, when the taps reach further, 64 .. 64 + span - 1
            const float* Qn = a.Q + (size_t)n * a.Bc * a.LQ;
            const int q1 = t0 + a.minoff + lane, q2 = q1 + 64;
            const bool ok1 = q1 >= 0 && q1 < a.LQ, ok2 = lane < a.span && q2 >= 0 && q2 < a.LQ;
            const unsigned c1 = (unsigned)min(max(q1, 0), a.LQ - 1), c2 = (unsigned)min(max(q2, 0), a.LQ - 1);
            const bool two = a.span > 0;
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {   // columns 0..63, then 64..64+span-1 (16 staging registers live at a time)
                if (ph == 1 && !two) break;
                float qr[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int br = b0v + r;
                    qr[r] = Qn[(unsigned)((br < a.Bc ? br : a.Bc - 1) * lq) + (ph ? c2 : c1)];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wave * 16 + r;
                    float v = qr[r] * qs;
                    v = fmaxf(v, v * a.q_slope);   // leaky-relu for slopes in [0, 1]
                    v = ((ph ? ok2 : ok1) && b0 + row < a.Bc) ? v : 0.f;
                    const _Float16 h = (_Float16)v;
                    Qh[row * WS_QP + ph * 64 + lane] = h;
                    Ql[row * WS_QP + ph * 64 + lane] = (_Float16)(v - (float)h);
                }
            }
        }
        }   // (edge chunk)
        __syncthreads();
        const _Float16* pa = Ph + (wm * 64 + l31) * WS_PP + half * 8;
        const _Float16* qb = Qh + (wn * 32 + l31) * WS_QP + half * 8 + (a.base + a.j0 * a.step - a.minoff);
#pragma unroll 1
        for (int ks = 0; ks < 4; ++ks) {
            half8 ah[2], al[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = *reinterpret_cast<const half8*>(pa + i * 32 * WS_PP + ks * 16);
                al[i] = *reinterpret_cast<const half8*>(pa + 128 * WS_PP + i * 32 * WS_PP + ks * 16);
            }
#pragma unroll
            for (int j = 0; j < JT; ++j) {
                const half8 bh = reinterpret_cast<const half8_u*>(qb + ks * 16 + j * a.step)->v;
                const half8 bl = reinterpret_cast<const half8_u*>(qb + 64 * WS_QP + ks * 16 + j * a.step)->v;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh, acc[i][j], 0, 0, 0);
                }
            }
        }
    }
    // partial tile -> workspace (C/D layout: column b = lane & 31, row a = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)); 128-byte runs per row
    const float un = 1.f / (sp * sq);
    const int bcol = b0 + wn * 32 + l31;
#pragma unroll
    for (int j = 0; j < JT; ++j) {
        float* dst = a.part + ((size_t)blockIdx.x * a.Jtot + a.j0 + j) * a.A * a.Bc;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int arow = a0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (arow < a.A && bcol < a.Bc) dst[(size_t)arow * a.Bc + bcol] = acc[i][j][r] * un;
            }
    }
}

template <int JT>
static int launch_wgrad_split(const WgsArgs& a, int splits, hipStream_t s) {
    const int tiles = ((a.A + 127) / 128) * ((a.Bc + 63) / 64);
    const size_t lds = (size_t)(2 * 128 * WS_PP + 2 * 64 * WS_QP) * sizeof(_Float16);
    if (int rc = ensure_full_lds(reinterpret_cast<const void*>(wgrad_f16x3_kernel<JT>))) return rc;
    const int extra = a.bias_part ? (int)(((long)a.A * a.bias_S + splits - 1) / splits) : 0;   // rows of bias-gradient workgroups behind the tiles
    hipLaunchKernelGGL(wgrad_f16x3_kernel<JT>, dim3(splits, tiles + extra), dim3(256), lds, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("wgrad_f16x3_kernel launch failed: %s", hipGetErrorString(e));
        return TTSC_EHIP;
    }
    return TTSC_OK;
}


// ---- grouped layers on the same scheme (round 6) ------------------------------------------------------------------------------------------
// The multi-scale discriminator's k = 41 layers are grouped (4 / 16 groups of 16 .. 64 output channels over 16 .. 128 de-interleaved input channels):
// their weight gradients were the last convolution work of the step on the exact fp32 matrix instruction (conv_wgrad_kernel: 8.3 ms of a 72 ms
// step at b = 16, profiles/r06_train_kernel_stats.csv).  A group is a small dense problem with a long contraction, so here a workgroup owns ONE
// group's rows (RT row tiles of 32: Ag <= 64) x 64 of its columns, and its four waves are (tap half) x (column half): wave (wt, wn) multiplies every
// 64-position chunk into RT x JT accumulator tiles for the taps j0 + wt JT .. of the launch — 2 JT taps per launch, 120 matrix instructions per wave and
// staged chunk as in the dense kernel.  (First version: the waves split the POSITIONS of a chunk instead — 42 instructions per staged chunk and three
// launches for 21 taps: 153 us per launch, slower than the exact kernel it was to replace, profiles/r06_grouped_wgrad_ab.log.)  Operand staging, fp16
// split, ranges and the unaligned tap reads are wgrad_f16x3_kernel's.
struct WgsgArgs {
    const float* P;        // [N][A][LP],  A = G * Ag
    const float* Q;        // [N][G * Bg][LQ]
    float* part;           // [gridDim.x][Jtot][A][Bg]
    const float* amax_q;
    const float* amax_p;
    int N, A, Bg, LP, LQ, G, Ag;
    int Jtot, j0, base, step;
    float q_scale, q_slope;
    int chunks, CH, items;
    int minoff, span;
};

template <int JT, int RT>
__global__ __launch_bounds__(256, 2) void wgrad_f16x3_grouped_kernel(WgsgArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smraw[];
    _Float16* Ph = reinterpret_cast<_Float16*>(smraw);   // [32 RT][WS_PP]
    _Float16* Pl = Ph + 32 * RT * WS_PP;
    _Float16* Qh = Pl + 32 * RT * WS_PP;                 // [64][WS_QP]
    _Float16* Ql = Qh + 64 * WS_QP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wt = wave >> 1, wn = wave & 1, half = lane >> 5, l31 = lane & 31;
    const int ctn = (a.Bg + 63) >> 6;
    const int g = blockIdx.y / ctn, b0 = (blockIdx.y % ctn) * 64;
    const int a0 = g * a.Ag;
    const int it_beg = blockIdx.x * a.CH, it_end = min(it_beg + a.CH, a.items);
    const float sq = pow2_to(*a.amax_q * a.q_scale, SPLIT_X_TARGET), sp = pow2_to(*a.amax_p, SPLIT_X_TARGET);
    const float qs = a.q_scale * sq;
    const int Btot = a.G * a.Bg;

    f32x16 acc[RT][JT];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < JT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Operand loads of chunk it + 1 are in flight while chunk it is multiplied (explicit register double buffer, as in conv_wgrad_kernel): the first
    // version staged load -> convert -> barrier -> multiply per chunk and spent ~7.5 us per 64 positions waiting for HBM (181 us per launch of the
    // 128 -> 128, k = 41 layer against 154 + 174 us on the exact kernel).
    f4u pv[2 * RT], qv[2][4];   // staging registers: pv[r >> 2].v[r & 3] = row r of the wave's share (edge chunks), or four positions of row 4 k + sub (interior chunks)
    const bool two = a.span > 0;
    const int sub = lane >> 4, q4 = 4 * (lane & 15);
    // interior chunk (positions and Q window inside the rows): four positions per lane, 16-byte loads and 8-byte LDS writes — see wgrad_f16x3_kernel
    auto is_interior = [&](int it) __attribute__((always_inline)) -> bool {
        const int n = it / a.chunks, t0 = (it - n * a.chunks) * 64, qlo = t0 + a.minoff;
        return t0 + 64 <= a.LP && qlo >= 0 && qlo + 64 + ((a.span + 3) & ~3) <= a.LQ;
    };
    auto issue = [&](int it) __attribute__((always_inline)) {
        const int n = it / a.chunks, t0 = (it - n * a.chunks) * 64;
        int lp = a.LP, lq = a.LQ, rowv = wave * (8 * RT), b0v = b0 + wave * 16;
        asm volatile("" : "+s"(lp), "+s"(lq), "+v"(rowv), "+v"(b0v));
        const float* Pn = a.P + ((size_t)n * a.A + a0) * a.LP;
        const float* Qn = a.Q + ((size_t)n * Btot + (size_t)g * a.Bg) * a.LQ;
        if (is_interior(it)) {
#pragma unroll
            for (int k = 0; k < 2 * RT; ++k) {
                const int row = rowv + 4 * k + sub;
                pv[k] = *reinterpret_cast<const f4u*>(Pn + (unsigned)((row < a.Ag ? row : a.Ag - 1) * lp) + t0 + q4);
            }
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
                if (ph == 1 && !two) break;
                const bool qok = ph == 0 || q4 < a.span;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int br = b0v + 4 * k + sub;
                    qv[ph][k] = *reinterpret_cast<const f4u*>(Qn + (unsigned)((br < a.Bg ? br : a.Bg - 1) * lq) + t0 + a.minoff + q4 + (qok ? ph * 64 : 0));
                }
            }
            return;
        }
        const int t = t0 + lane;
        const unsigned tc = (unsigned)(t < a.LP ? t : a.LP - 1);
#pragma unroll
        for (int r = 0; r < 8 * RT; ++r) {
            const int row = rowv + r;
            pv[r >> 2].v[r & 3] = Pn[(unsigned)((row < a.Ag ? row : a.Ag - 1) * lp) + tc];
        }
        const int q1 = t0 + a.minoff + lane, q2 = q1 + 64;
        const unsigned c1 = (unsigned)min(max(q1, 0), a.LQ - 1), c2 = (unsigned)min(max(q2, 0), a.LQ - 1);
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            if (ph == 1 && !two) break;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int br = b0v + r;
                qv[ph][r >> 2].v[r & 3] = Qn[(unsigned)((br < a.Bg ? br : a.Bg - 1) * lq) + (ph ? c2 : c1)];
            }
        }
    };
    auto commit = [&](int it) __attribute__((always_inline)) {
        const int n = it / a.chunks, t0 = (it - n * a.chunks) * 64;
        if (is_interior(it)) {
#pragma unroll
            for (int k = 0; k < 2 * RT; ++k) {
                const int row = wave * (8 * RT) + 4 * k + sub;
                const bool rok = row < a.Ag;
                half4 h, l;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = rok ? pv[k].v[e] * sp : 0.f;
                    h[e] = (_Float16)v;
                    l[e] = (_Float16)(v - (float)h[e]);
                }
                *reinterpret_cast<half4*>(Ph + row * WS_PP + q4) = h;
                *reinterpret_cast<half4*>(Pl + row * WS_PP + q4) = l;
            }
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
                if (ph == 1 && !two) break;
                const bool qok = ph == 0 || q4 < a.span;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int row = wave * 16 + 4 * k + sub;
                    const bool rok = b0 + row < a.Bg;
                    half4 h, l;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = qv[ph][k].v[e] * qs;
                        v = fmaxf(v, v * a.q_slope);
                        v = rok ? v : 0.f;
                        h[e] = (_Float16)v;
                        l[e] = (_Float16)(v - (float)h[e]);
                    }
                    if (qok) {
                        *reinterpret_cast<half4*>(Qh + row * WS_QP + ph * 64 + q4) = h;
                        *reinterpret_cast<half4*>(Ql + row * WS_QP + ph * 64 + q4) = l;
                    }
                }
            }
            return;
        }
        const bool tok = t0 + lane < a.LP;
#pragma unroll
        for (int r = 0; r < 8 * RT; ++r) {
            const int row = wave * (8 * RT) + r;
            const float v = (tok && row < a.Ag) ? pv[r >> 2].v[r & 3] * sp : 0.f;
            const _Float16 h = (_Float16)v;
            Ph[row * WS_PP + lane] = h;
            Pl[row * WS_PP + lane] = (_Float16)(v - (float)h);
        }
        const int q1 = t0 + a.minoff + lane, q2 = q1 + 64;
        const bool ok1 = q1 >= 0 && q1 < a.LQ, ok2 = lane < a.span && q2 >= 0 && q2 < a.LQ;
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            if (ph == 1 && !two) break;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wave * 16 + r;
                float v = qv[ph][r >> 2].v[r & 3] * qs;
                v = fmaxf(v, v * a.q_slope);
                v = ((ph ? ok2 : ok1) && b0 + row < a.Bg) ? v : 0.f;
                const _Float16 h = (_Float16)v;
                Qh[row * WS_QP + ph * 64 + lane] = h;
                Ql[row * WS_QP + ph * 64 + lane] = (_Float16)(v - (float)h);
            }
        }
    };
    if (it_beg < it_end) issue(it_beg);
    for (int it = it_beg; it < it_end; ++it) {
        __syncthreads();   // the fragment reads of the previous chunk are done
        commit(it);
        __syncthreads();
        if (it + 1 < it_end) issue(it + 1);
        __builtin_amdgcn_sched_barrier(0);   // (the loads leave before the multiplications, not at their first use)
        const _Float16* pa = Ph + l31 * WS_PP + half * 8;
        const _Float16* qb = Qh + (wn * 32 + l31) * WS_QP + half * 8 + (a.base + (a.j0 + wt * JT) * a.step - a.minoff);
#pragma unroll 1
        for (int ks = 0; ks < 4; ++ks) {
            half8 ah[RT], al[RT];
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                ah[i] = *reinterpret_cast<const half8*>(pa + i * 32 * WS_PP + ks * 16);
                al[i] = *reinterpret_cast<const half8*>(pa + 32 * RT * WS_PP + i * 32 * WS_PP + ks * 16);
            }
#pragma unroll
            for (int j = 0; j < JT; ++j) {
                const half8 bh = reinterpret_cast<const half8_u*>(qb + ks * 16 + j * a.step)->v;
                const half8 bl = reinterpret_cast<const half8_u*>(qb + 64 * WS_QP + ks * 16 + j * a.step)->v;
#pragma unroll
                for (int i = 0; i < RT; ++i) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh, acc[i][j], 0, 0, 0);
                }
            }
        }
    }
    const float un = 1.f / (sp * sq);
    const int bcol = b0 + wn * 32 + l31;
#pragma unroll
    for (int j = 0; j < JT; ++j) {
        const int jj = a.j0 + wt * JT + j;
        if (jj >= a.Jtot) break;   // (the last launch's second tap half may run past the kernel: computed on whatever the window holds, never stored)
        float* dst = a.part + ((size_t)blockIdx.x * a.Jtot + jj) * a.A * a.Bg;
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lrow = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (lrow < a.Ag && bcol < a.Bg) dst[(size_t)(a0 + lrow) * a.Bg + bcol] = acc[i][j][r] * un;
            }
    }
}

template <int JT, int RT>
static int launch_wgrad_split_grouped(const WgsgArgs& a, int splits, hipStream_t s) {
    const int tiles = a.G * ((a.Bg + 63) / 64);
    const size_t lds = (size_t)(2 * 32 * RT * WS_PP + 2 * 64 * WS_QP) * sizeof(_Float16);
    hipLaunchKernelGGL((wgrad_f16x3_grouped_kernel<JT, RT>), dim3(splits, tiles), dim3(256), lds, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("wgrad_f16x3_grouped_kernel launch failed: %s", hipGetErrorString(e));
        return TTSC_EHIP;
    }
    return TTSC_OK;
}
template <int RT>
static int launch_wgrad_split_grouped_jt(const WgsgArgs& a, int jt, int splits, hipStream_t s) {
    switch (jt) {
        case 1: return launch_wgrad_split_grouped<1, RT>(a, splits, s);
        case 2: return launch_wgrad_split_grouped<2, RT>(a, splits, s);
        case 3: return launch_wgrad_split_grouped<3, RT>(a, splits, s);
        case 4: return launch_wgrad_split_grouped<4, RT>(a, splits, s);
        default: break;
    }
    if constexpr (RT == 1) {
        switch (jt) {
            case 6: return launch_wgrad_split_grouped<6, 1>(a, splits, s);
            case 7: return launch_wgrad_split_grouped<7, 1>(a, splits, s);
            case 8: return launch_wgrad_split_grouped<8, 1>(a, splits, s);
            case 9: return launch_wgrad_split_grouped<9, 1>(a, splits, s);
            case 10: return launch_wgrad_split_grouped<10, 1>(a, splits, s);
            default: break;
        }
    }
    return launch_wgrad_split_grouped<5, RT>(a, splits, s);
}

}  // namespace ttsc

using namespace ttsc;

// split of the position axis: ~2048 waves (two rounds of one wave per SIMD)
static void wgrad_split(int N, int A, int Bc, int64_t LP, int* chunks, int* CH, int* groups, int* splits) {
    *chunks = (int)ceil_div(LP, WG_TK);
    const int tiles = ((A + 31) / 32) * ((Bc + 31) / 32);
    long ch = ((long)N * *chunks * tiles + 2047) / 2048;
    if (ch < 1) ch = 1;
    if (ch > *chunks) ch = *chunks;
    *CH = (int)ch;
    *groups = (int)ceil_div(*chunks, *CH);
    *splits = (N * *groups + 3) / 4;
}

extern "C" size_t ttsc_conv_wgrad_workspace_bytes(int32_t N, int32_t A, int32_t Bc, int64_t LP, int32_t J) {
    if (N <= 0 || A <= 0 || Bc <= 0 || LP <= 0 || J <= 0) return 0;
    int chunks, CH, groups, splits;
    wgrad_split(N, A, Bc, LP, &chunks, &CH, &groups, &splits);
    return (size_t)splits * J * A * Bc * sizeof(float);
}

extern "C" int ttsc_conv_wgrad(const float* p_dev, const float* q_dev, float* g_dev, int32_t N, int32_t A, int32_t Bc, int64_t LP,
                               int64_t LQ, int32_t J, int32_t base, int32_t step, float q_scale, float q_slope, void* ws_dev,
                               size_t ws_bytes, void* stream) {
    return ttsc_conv_wgrad_grouped(p_dev, q_dev, g_dev, N, A, Bc, 1, LP, LQ, J, base, step, q_scale, q_slope, ws_dev, ws_bytes, stream);
}

extern "C" int ttsc_conv_wgrad_grouped(const float* p_dev, const float* q_dev, float* g_dev, int32_t N, int32_t A, int32_t Bc, int32_t cgroups,
                                       int64_t LP, int64_t LQ, int32_t J, int32_t base, int32_t step, float q_scale, float q_slope, void* ws_dev,
                                       size_t ws_bytes, void* stream) {
    TTSC_REQUIRE(p_dev && q_dev && g_dev, "ttsc_conv_wgrad: null argument");
    TTSC_REQUIRE(cgroups >= 1 && A % cgroups == 0, "ttsc_conv_wgrad: %d rows do not split into %d groups", A, cgroups);
    TTSC_REQUIRE(N > 0 && A > 0 && Bc > 0 && LP > 0 && LQ > 0 && J > 0, "ttsc_conv_wgrad: bad shape N=%d A=%d B=%d LP=%lld LQ=%lld J=%d", N, A, Bc,
                 (long long)LP, (long long)LQ, J);
    TTSC_REQUIRE(LP < (1ll << 30) && LQ < (1ll << 30), "ttsc_conv_wgrad: length too large");
    TTSC_REQUIRE(q_slope >= 0.f && q_slope <= 1.f, "ttsc_conv_wgrad: q_slope must be in [0,1]");
    hipStream_t s = (hipStream_t)stream;
    WgArgs a;
    a.P = p_dev;
    a.Q = q_dev;
    a.part = (float*)ws_dev;
    TTSC_REQUIRE(ws_dev && ws_bytes >= ttsc_conv_wgrad_workspace_bytes(N, A, Bc, LP, J), "ttsc_conv_wgrad: workspace too small (%zu < %zu)",
                 ws_bytes, ttsc_conv_wgrad_workspace_bytes(N, A, Bc, LP, J));
    a.N = N;
    a.A = A;
    a.Bc = Bc;
    a.LP = (int)LP;
    a.LQ = (int)LQ;
    a.Jtot = J;
    a.base = base;
    a.step = step;
    a.q_scale = q_scale;
    a.q_slope = q_slope;
    a.cg_n = cgroups;
    a.Ag = A / cgroups;
    a.Btot = cgroups * Bc;
    int splits;
    wgrad_split(N, A, Bc, LP, &a.chunks, &a.CH, &a.groups, &splits);
    for (int j0 = 0; j0 < J; j0 += 12) {
        a.j0 = j0;
        a.J = J - j0 < 12 ? J - j0 : 12;
        const int off_first = base + j0 * step, off_last = base + (j0 + a.J - 1) * step;
        a.minoff = off_first < off_last ? off_first : off_last;
        a.span = (off_first < off_last ? off_last : off_first) - a.minoff;
        TTSC_REQUIRE(a.span <= 64, "ttsc_conv_wgrad: tap window (%d positions) exceeds 64", a.span);
        int rc;
        switch (a.J) {   // one instance per tap count: no per-tap branch in the MFMA loop
            case 1: rc = launch_wgrad<1>(a, s); break;
            case 2: rc = launch_wgrad<2>(a, s); break;
            case 3: rc = launch_wgrad<3>(a, s); break;
            case 4: rc = launch_wgrad<4>(a, s); break;
            case 5: rc = launch_wgrad<5>(a, s); break;
            case 6: rc = launch_wgrad<6>(a, s); break;
            case 7: rc = launch_wgrad<7>(a, s); break;
            case 8: rc = launch_wgrad<8>(a, s); break;
            case 9: rc = launch_wgrad<9>(a, s); break;
            case 10: rc = launch_wgrad<10>(a, s); break;
            case 11: rc = launch_wgrad<11>(a, s); break;
            default: rc = launch_wgrad<12>(a, s); break;
        }
        if (rc) return rc;
    }
    const long AB = (long)A * Bc;
    const long blocks = (AB * J + 31) / 32;
    TTSC_REQUIRE(blocks < (1l << 31), "ttsc_conv_wgrad: weight tensor too large");
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)ws_dev, g_dev, splits, (int)J, AB);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("wgrad_reduce_kernel launch failed: %s", hipGetErrorString(e));
        return TTSC_EHIP;
    }
    return TTSC_OK;
}


// ---- split-precision variant (wgrad_f16x3_kernel) ---------------------------------------------------------------------------------
static void wgrad_split_plan(int N, int A, int Bc, int64_t LP, int* chunks, int* CH, int* items, int* splits) {
    *chunks = (int)ceil_div(LP, 64);
    *items = N * *chunks;
    const int tiles = ((A + 127) / 128) * ((Bc + 63) / 64);
    long want = (512 + tiles - 1) / tiles;   // ~two workgroups per CU
    if (want < 1) want = 1;
    if (want > *items) want = *items;
    *CH = (int)ceil_div(*items, want);
    *splits = (int)ceil_div(*items, *CH);    // every split owns at least one item
}

extern "C" int32_t ttsc_conv_wgrad_split_supported(int32_t A, int32_t Bc, int32_t J, int32_t step) {
    const int st = step < 0 ? -step : step;
    return A >= 64 && Bc >= 32 && J >= 1 && J <= 16 && st >= 1 && st <= 64;
}

// slices per row of the bias gradient that rides along: train_ops.hip::bias_grad_splits (>= 1024 positions of every batch item per workgroup, ~1024 in all)
static int wgrad_bias_slices(int32_t A, int64_t LP) {
    long s = (LP + 1023) / 1024;
    const long cap = (1024 + A - 1) / A;
    if (s > cap) s = cap;
    return (int)(s < 1 ? 1 : s);
}

extern "C" size_t ttsc_conv_wgrad_split_workspace_bytes(int32_t N, int32_t A, int32_t Bc, int64_t LP, int32_t J) {
    if (N <= 0 || A <= 0 || Bc <= 0 || LP <= 0 || J <= 0) return 0;
    int chunks, CH, items, splits;
    wgrad_split_plan(N, A, Bc, LP, &chunks, &CH, &items, &splits);
    return 256 + (size_t)splits * J * A * Bc * sizeof(float) + (size_t)A * wgrad_bias_slices(A, LP) * sizeof(float);
}

extern "C" int ttsc_conv_wgrad_split(const float* p_dev, const float* q_dev, float* g_dev, int32_t N, int32_t A, int32_t Bc, int64_t LP, int64_t LQ,
                                     int32_t J, int32_t base, int32_t step, float q_scale, float q_slope, float* amax_q, float* amax_p, int32_t measure, void* ws_dev,
                                     size_t ws_bytes, void* stream) {
    return ttsc_conv_wgrad_split_bias(p_dev, q_dev, g_dev, nullptr, N, A, Bc, LP, LQ, J, base, step, q_scale, q_slope, amax_q, amax_p, measure, ws_dev, ws_bytes,
                                      stream);
}

extern "C" int ttsc_conv_wgrad_split_bias(const float* p_dev, const float* q_dev, float* g_dev, float* db_dev, int32_t N, int32_t A, int32_t Bc, int64_t LP,
                                          int64_t LQ, int32_t J, int32_t base, int32_t step, float q_scale, float q_slope, float* amax_q, float* amax_p,
                                          int32_t measure, void* ws_dev, size_t ws_bytes, void* stream) {
    TTSC_REQUIRE(p_dev && q_dev && g_dev && ws_dev, "ttsc_conv_wgrad_split: null argument");
    TTSC_REQUIRE(N > 0 && LP > 0 && LQ > 0 && ttsc_conv_wgrad_split_supported(A, Bc, J, step), "ttsc_conv_wgrad_split: shape not supported (N=%d A=%d B=%d J=%d step=%d)",
                 N, A, Bc, J, step);
    TTSC_REQUIRE((int64_t)N * A * LP < (1ll << 31) && (int64_t)N * Bc * LQ < (1ll << 31), "ttsc_conv_wgrad_split: tensor too large");
    TTSC_REQUIRE(q_slope >= 0.f && q_slope <= 1.f && q_scale > 0.f, "ttsc_conv_wgrad_split: q_slope must be in [0,1], q_scale positive");
    TTSC_REQUIRE(ws_bytes >= ttsc_conv_wgrad_split_workspace_bytes(N, A, Bc, LP, J) && ((uintptr_t)ws_dev & 15) == 0, "ttsc_conv_wgrad_split: workspace too small or misaligned");
    hipStream_t s = (hipStream_t)stream;
    // range words as in ttsc_conv_train: measure bit 0 = max |Q| now, bit 1 = max |P| now; null pointers = workspace words, measured here
    float* ws_words = reinterpret_cast<float*>(ws_dev);
    if (!amax_q) { amax_q = ws_words; measure |= 1; }
    if (!amax_p) { amax_p = ws_words + 1; measure |= 2; }
    // (bit 2: the caller's words come from a pool it zeroed with one launch for the whole step — no memset here)
    if (int rc = launch_amax2((measure & 1) ? q_dev : nullptr, (long)N * Bc * LQ, amax_q, (measure & 2) ? p_dev : nullptr, (long)N * A * LP, amax_p, s, (measure & 4) != 0)) return rc;
    WgsArgs a;
    a.P = p_dev;
    a.Q = q_dev;
    a.part = reinterpret_cast<float*>(reinterpret_cast<char*>(ws_dev) + 256);
    a.amax_q = amax_q;
    a.amax_p = amax_p;
    a.N = N;
    a.A = A;
    a.Bc = Bc;
    a.LP = (int)LP;
    a.LQ = (int)LQ;
    a.Jtot = J;
    a.base = base;
    a.step = step;
    a.q_scale = q_scale;
    a.q_slope = q_slope;
    int splits;
    wgrad_split_plan(N, A, Bc, LP, &a.chunks, &a.CH, &a.items, &splits);
    // the layer's bias gradient db[a] = sum_{n,t} P[n,a,t] in the same two launches (its slices behind the tiles of the first tap group, their sum
    // behind the reduction): 146 launches of a Cubegan step that were a bias_grad_kernel each (profiles/r06_train_wbank_kernel_stats.csv)
    const int bias_S = wgrad_bias_slices(A, LP);
    float* bias_part = a.part + (size_t)splits * J * A * Bc;
    a.bias_part = db_dev ? bias_part : nullptr;
    a.bias_S = bias_S;
    a.bias_y0 = ((A + 127) / 128) * ((Bc + 63) / 64);
    const int st = step < 0 ? -step : step;
    const int jt_max = std::min(5, 64 / st + 1);   // taps per launch: five accumulator pairs, and a Q window of at most 64 extra positions
    for (int j0 = 0; j0 < J;) {
        const int jt = std::min(jt_max, J - j0);
        a.j0 = j0;
        const int off_first = base + j0 * step, off_last = base + (j0 + jt - 1) * step;
        a.minoff = std::min(off_first, off_last);
        a.span = std::max(off_first, off_last) - a.minoff;
        int rc;
        switch (jt) {
            case 1: rc = launch_wgrad_split<1>(a, splits, s); break;
            case 2: rc = launch_wgrad_split<2>(a, splits, s); break;
            case 3: rc = launch_wgrad_split<3>(a, splits, s); break;
            case 4: rc = launch_wgrad_split<4>(a, splits, s); break;
            default: rc = launch_wgrad_split<5>(a, splits, s); break;
        }
        if (rc) return rc;
        j0 += jt;
        a.bias_part = nullptr;   // (the first tap group's launch carried the slices)
    }
    const long AB = (long)A * Bc;
    const long blocks = (AB * J + 31) / 32;
    TTSC_REQUIRE(blocks < (1l << 31) - 4096, "ttsc_conv_wgrad_split: weight tensor too large");
    const unsigned bias_blocks = db_dev ? (unsigned)((A + 255) / 256) : 0u;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)blocks + bias_blocks), dim3(256), 0, s, (const float*)a.part, g_dev, splits, (int)J, AB,
                       db_dev ? (const float*)bias_part : nullptr, db_dev, (int)A, bias_S, (unsigned)blocks);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("wgrad_reduce_kernel launch failed: %s", hipGetErrorString(e));
        return TTSC_EHIP;
    }
    return TTSC_OK;
}

// ---- grouped split-precision variant (wgrad_f16x3_grouped_kernel) --------------------------------------------------------------------------
static void wgrad_split_grouped_plan(int N, int G, int Bg, int64_t LP, int* chunks, int* CH, int* items, int* splits) {
    *chunks = (int)ceil_div(LP, 64);
    *items = N * *chunks;
    const int tiles = G * ((Bg + 63) / 64);
    long want = (512 + tiles - 1) / tiles;   // ~two workgroups per CU
    if (want < 1) want = 1;
    if (want > *items) want = *items;
    *CH = (int)ceil_div(*items, want);
    *splits = (int)ceil_div(*items, *CH);
}

extern "C" int32_t ttsc_conv_wgrad_split_grouped_supported(int32_t A, int32_t Bg, int32_t groups, int32_t J, int32_t step) {
    const int st = step < 0 ? -step : step;
    if (groups < 2 || A % groups) return 0;
    const int Ag = A / groups;
    return Ag >= 16 && Ag <= 64 && Bg >= 16 && J >= 1 && J <= 64 && step >= 1 && step <= 3;   // (the taps of a launch's two halves share one staged window: 19 steps <= 64)
}

extern "C" size_t ttsc_conv_wgrad_split_grouped_workspace_bytes(int32_t N, int32_t A, int32_t Bg, int32_t groups, int64_t LP, int32_t J) {
    if (N <= 0 || A <= 0 || Bg <= 0 || groups <= 0 || LP <= 0 || J <= 0) return 0;
    int chunks, CH, items, splits;
    wgrad_split_grouped_plan(N, groups, Bg, LP, &chunks, &CH, &items, &splits);
    return 256 + (size_t)splits * J * A * Bg * sizeof(float);
}

extern "C" int ttsc_conv_wgrad_split_grouped(const float* p_dev, const float* q_dev, float* g_dev, int32_t N, int32_t A, int32_t Bg, int32_t groups, int64_t LP,
                                             int64_t LQ, int32_t J, int32_t base, int32_t step, float q_scale, float q_slope, float* amax_q, float* amax_p,
                                             int32_t measure, void* ws_dev, size_t ws_bytes, void* stream) {
    TTSC_REQUIRE(p_dev && q_dev && g_dev && ws_dev, "ttsc_conv_wgrad_split_grouped: null argument");
    TTSC_REQUIRE(N > 0 && LP > 0 && LQ > 0 && ttsc_conv_wgrad_split_grouped_supported(A, Bg, groups, J, step),
                 "ttsc_conv_wgrad_split_grouped: shape not supported (N=%d A=%d Bg=%d groups=%d J=%d step=%d)", N, A, Bg, groups, J, step);
    TTSC_REQUIRE((int64_t)N * A * LP < (1ll << 31) && (int64_t)N * Bg * groups * LQ < (1ll << 31), "ttsc_conv_wgrad_split_grouped: tensor too large");
    TTSC_REQUIRE(q_slope >= 0.f && q_slope <= 1.f && q_scale > 0.f, "ttsc_conv_wgrad_split_grouped: q_slope must be in [0,1], q_scale positive");
    TTSC_REQUIRE(ws_bytes >= ttsc_conv_wgrad_split_grouped_workspace_bytes(N, A, Bg, groups, LP, J) && ((uintptr_t)ws_dev & 15) == 0,
                 "ttsc_conv_wgrad_split_grouped: workspace too small or misaligned");
    hipStream_t s = (hipStream_t)stream;
    float* ws_words = reinterpret_cast<float*>(ws_dev);
    if (!amax_q) { amax_q = ws_words; measure |= 1; }
    if (!amax_p) { amax_p = ws_words + 1; measure |= 2; }
    if (int rc = launch_amax2((measure & 1) ? q_dev : nullptr, (long)N * Bg * groups * LQ, amax_q, (measure & 2) ? p_dev : nullptr, (long)N * A * LP, amax_p, s,
                              (measure & 4) != 0))
        return rc;
    WgsgArgs a;
    a.P = p_dev;
    a.Q = q_dev;
    a.part = reinterpret_cast<float*>(reinterpret_cast<char*>(ws_dev) + 256);
    a.amax_q = amax_q;
    a.amax_p = amax_p;
    a.N = N;
    a.A = A;
    a.Bg = Bg;
    a.G = groups;
    a.Ag = A / groups;
    a.LP = (int)LP;
    a.LQ = (int)LQ;
    a.Jtot = J;
    a.base = base;
    a.step = step;
    a.q_scale = q_scale;
    a.q_slope = q_slope;
    int splits;
    wgrad_split_grouped_plan(N, groups, Bg, LP, &a.chunks, &a.CH, &a.items, &splits);
    const int RT = a.Ag > 32 ? 2 : 1;
    const int jt_max = RT == 1 ? 6 : 3;                    // taps per WAVE: RT x JT accumulator tiles (<= 96 registers beside the 16 + 32 prefetch registers and the two staging paths); a launch covers 2 JT taps
    const int nl = (J + 2 * jt_max - 1) / (2 * jt_max);    // launches, with tap ranges of (nearly) equal size
    const int tl = (J + nl - 1) / nl, jt = (tl + 1) / 2;
    for (int j0 = 0; j0 < J; j0 += 2 * jt) {
        a.j0 = j0;
        const int jlast = std::min(j0 + 2 * jt, (int)J) - 1;
        a.minoff = base + j0 * step;                       // (step >= 1: the first tap of the launch has the smallest offset)
        a.span = (jlast - j0) * step;
        const int rc = RT == 1 ? launch_wgrad_split_grouped_jt<1>(a, jt, splits, s) : launch_wgrad_split_grouped_jt<2>(a, jt, splits, s);
        if (rc) return rc;
    }
    const long AB = (long)A * Bg;
    const long blocks = (AB * J + 31) / 32;
    TTSC_REQUIRE(blocks < (1l << 31), "ttsc_conv_wgrad_split_grouped: weight tensor too large");
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)a.part, g_dev, splits, (int)J, AB, (const float*)nullptr,
                       (float*)nullptr, 0, 0, 0u);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("wgrad_reduce_kernel launch failed: %s", hipGetErrorString(e));
        return TTSC_EHIP;
    }
    return TTSC_OK;
}
