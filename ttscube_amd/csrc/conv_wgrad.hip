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

namespace ttsc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

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
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ G, int splits, int J, long AB) {
    __shared__ float red[8][32];
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
