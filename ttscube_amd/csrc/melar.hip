// Autoregressive mel decoder loop of CubenetTextcoder as ONE persistent kernel (gfx950).
//
// Reference loop: cube/networks/textcoder.py:174-185 — per AR step: PreNet (2 x [Linear -> ReLU -> dropout p=0.5 ALWAYS on],
// cube/networks/modules.py:159-164) on the last emitted mel frame, concat with the overlay-BiLSTM state, one step of a
// 2-layer LSTM (512), Linear 512 -> 240 (three frames), feed the last 80 back.  In PyTorch that is ~12 kernel launches
// per step; here the whole S-step loop of an utterance runs inside one workgroup:
//   * the overlay part of the layer-1 input projection (1024 of its 1280 input columns) does not depend on the recurrence
//     and is hoisted for all steps into one MFMA GEMM by the caller (xg1 = overlay . W_ih1[:, :1024]^T + b_ih1 + b_hh1);
//   * thread j owns hidden unit j of BOTH LSTM layers (its 4 gate rows per matrix are streamed from L2 as packed 16-byte
//     loads with explicit double-buffered prefetch, c_j stays in a register, h in LDS); PreNet rows and output rows are
//     owned by the first 256 / 240 threads; 5 workgroup barriers per step, no inter-workgroup communication.
// Dropout masks: injected ({0,1} floats, parity tests) or drawn in-kernel from Philox-4x32-10 counters (step, layer, unit).
#include "common.hpp"
#include "../../include/ttscube_math.h"
#include "rnn_chain.hpp"

namespace ttsc {

struct MelArArgs {
    const float* xg1;     // [B, S, 4H]
    const float* w_p2l;   // W_ih1[:, 1024:1280] packed [P/4][4H][4]   (P = prenet size 256)
    const float* w_hh1;   // [H/4][4H][4]
    const float* w_ih2;   // [H/4][4H][4]
    const float* w_hh2;   // [H/4][4H][4]
    const float* b2;      // [4H] = b_ih2 + b_hh2
    const float* w_out;   // [H/4][O][4]
    const float* b_out;   // [O]
    const float* w_pn1;   // [M/4][P][4]   (M = 80 mel bins)
    const float* b_pn1;
    const float* w_pn2;   // [P/4][P][4]
    const float* b_pn2;
    const float* masks;   // [B, S, 2, P] or null
    const int* steps;     // [B] valid steps per utterance or null
    float* y;             // [B, S, O]
    int B, S, H, P, M, O;
    float init_mel;       // -5
    unsigned long long seed;
};

template <int NG, int UN>
__device__ __forceinline__ void ar_chain(float (&acc)[NG], const float* __restrict__ wp, int rows, int gstride, int row,
                                         const float* v, int K) {
    // The lane's row index is made opaque at every call: the persistent kernels call this inside their time-step loop with the same
    // weights every step, so all UN x NG load addresses are loop-invariant — hoisted out of the step loop they are 64-bit per-lane
    // values that do not fit the register file (round-3 review: up to 225 VGPRs in scratch, each reload a dependent
    // scratch_load -> global_load pair inside the step).  Recomputing them per call is two VALU instructions per load.
    asm volatile("" : "+v"(row));
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
            const float4 hv = *reinterpret_cast<const float4*>(v + 4 * (kb0 + q));
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                float x = acc[g];
                x = fmaf(w[q][g].x, hv.x, x);
                x = fmaf(w[q][g].y, hv.y, x);
                x = fmaf(w[q][g].z, hv.z, x);
                x = fmaf(w[q][g].w, hv.w, x);
                acc[g] = x;
            }
        }
    };
    float4 wa[UN][NG], wb[UN][NG];
    const int NB = KB / UN;  // caller guarantees KB % UN == 0
    load(wa, 0);
    for (int bi = 0; bi < NB; bi += 2) {
        if (bi + 1 < NB) load(wb, (bi + 1) * UN);
        fma_batch(wa, bi * UN);
        if (bi + 2 < NB) load(wa, (bi + 2) * UN);
        if (bi + 1 < NB) fma_batch(wb, (bi + 1) * UN);
    }
}

__global__ __launch_bounds__(512) void melar_kernel(MelArArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int H = a.H, P = a.P, M = a.M, O = a.O, H4 = 4 * H;
    float* h1 = sm;               // [2][H]
    float* h2 = h1 + 2 * H;       // [2][H]
    float* p1 = h2 + 2 * H;       // [P]
    float* p2 = p1 + P;           // [P]
    float* lm = p2 + P;           // [M] last mel frame
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const int nsteps = a.steps ? a.steps[b] : a.S;
    for (int i = tid; i < 4 * H; i += blockDim.x) sm[i] = 0.f;
    for (int i = tid; i < M; i += blockDim.x) lm[i] = a.init_mel;
    float c1 = 0.f, c2 = 0.f;
    __syncthreads();
    int cur = 0;
    for (int t = 0; t < nsteps; ++t) {
        const int nxt = cur ^ 1;
        auto mask = [&](int layer, int unit) -> float {
            if (a.masks) return a.masks[(((size_t)b * a.S + t) * 2 + layer) * P + unit];
            uint32_t r4[4];
            ttsc_philox4x32((uint32_t)(unit >> 2), (uint32_t)t, (uint32_t)b, (uint32_t)layer, (uint32_t)a.seed, (uint32_t)(a.seed >> 32), r4);
            return (r4[unit & 3] & 1u) ? 1.f : 0.f;   // Bernoulli(0.5)
        };
        // ---- PreNet layer 1: relu(W1 . last_mel + b1) * mask * 2 ----
        if (tid < P) {
            float acc[1] = {a.b_pn1[tid]};
            ar_chain<1, 5>(acc, a.w_pn1, P, 0, tid, lm, M);   // M = 80 -> 20 k-blocks
            p1[tid] = fmaxf(acc[0], 0.f) * (mask(0, tid) * 2.f);
        }
        __syncthreads();
        // ---- PreNet layer 2 ----
        if (tid < P) {
            float acc[1] = {a.b_pn2[tid]};
            ar_chain<1, 8>(acc, a.w_pn2, P, 0, tid, p1, P);
            p2[tid] = fmaxf(acc[0], 0.f) * (mask(1, tid) * 2.f);
        }
        __syncthreads();
        // ---- LSTM layer 1: gates = xg1[t] + W_ih1[:, prenet part] . p2 + W_hh1 . h1 ----
        if (tid < H) {
            float acc[4];
            const float* xr = a.xg1 + ((size_t)b * a.S + t) * H4 + tid;
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = xr[g * H];
            ar_chain<4, 2>(acc, a.w_p2l, H4, H, tid, p2, P);
            ar_chain<4, 2>(acc, a.w_hh1, H4, H, tid, h1 + cur * H, H);
            const float ig = ttsc_sigmoidf(acc[0]), fg = ttsc_sigmoidf(acc[1]), gg = ttsc_tanhf(acc[2]), og = ttsc_sigmoidf(acc[3]);
            c1 = fmaf(fg, c1, ig * gg);
            h1[nxt * H + tid] = og * ttsc_tanhf(c1);
        }
        __syncthreads();
        // ---- LSTM layer 2 ----
        if (tid < H) {
            float acc[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = a.b2[g * H + tid];
            ar_chain<4, 2>(acc, a.w_ih2, H4, H, tid, h1 + nxt * H, H);
            ar_chain<4, 2>(acc, a.w_hh2, H4, H, tid, h2 + cur * H, H);
            const float ig = ttsc_sigmoidf(acc[0]), fg = ttsc_sigmoidf(acc[1]), gg = ttsc_tanhf(acc[2]), og = ttsc_sigmoidf(acc[3]);
            c2 = fmaf(fg, c2, ig * gg);
            h2[nxt * H + tid] = og * ttsc_tanhf(c2);
        }
        __syncthreads();
        // ---- output Linear H -> O (three frames); the last M values are fed back ----
        if (tid < O) {
            float acc[1] = {a.b_out[tid]};
            ar_chain<1, 8>(acc, a.w_out, O, 0, tid, h2 + nxt * H, H);
            a.y[((size_t)b * a.S + t) * O + tid] = acc[0];
            if (tid >= O - M) lm[tid - (O - M)] = acc[0];
        }
        __syncthreads();
        cur = nxt;
    }
    // steps beyond this utterance's own length are zero
    for (int t = nsteps; t < a.S; ++t)
        for (int i = tid; i < O; i += blockDim.x) a.y[((size_t)b * a.S + t) * O + i] = 0.f;
}


// ---------------------------------------------------------------------------------------------------------------
// Split variant: the reference API decodes ONE utterance (B = 1), i.e. one workgroup streaming 15.5 MB of weights per AR
// step from L2 through one CU's load path (209 us/step).  Here G workgroups share the utterance: member m owns H/G hidden
// units of BOTH LSTM layers (their gate rows only: (2.1 + 3 x 4.2) MB / G per step), its 512 threads are (unit, k-slice)
// pairs reduced through LDS, and the members exchange h1_t and h2_t (two hand-offs per step, protocol of rnn_chain.hpp).
// The PreNet (0.34 MB) and the output Linear (0.5 MB) are evaluated redundantly by every member, so the fed-back frame
// never has to be exchanged; member 0 writes y.  Results differ from melar_kernel in summation order only.
struct MelArSplitArgs {
    MelArArgs a;
    float* xh;            // [B][2 (layer)][2 (step parity)][H] exchanged hidden states
    unsigned* cnt;        // [B] monotonic counters
    unsigned* abort_word;
    int G, HU, KS;
};

__global__ __launch_bounds__(512) void melar_split_kernel(MelArSplitArgs s) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ int ok_s;
    const MelArArgs& a = s.a;
    const int H = a.H, P = a.P, M = a.M, O = a.O, H4 = 4 * H, HU = s.HU, KS = s.KS;
    float* h1 = sm;               // [H] full h1_{t-1} / h1_t
    float* h2 = h1 + H;           // [H]
    float* p1 = h2 + H;           // [P]
    float* p2 = p1 + P;           // [P]
    float* lm = p2 + P;           // [M] last mel frame
    float* part = lm + M;         // [KS][4][HU]
    const int tid = threadIdx.x, u = tid % HU, ks = tid / HU;
    const int m = blockIdx.x, b = blockIdx.y;
    const int j = m * HU + u;
    const bool owner = ks == 0;
    const int KP = P / KS, KH = H / KS;
    const int nsteps = a.steps ? a.steps[b] : a.S;
    unsigned* cnt = s.cnt + b;
    float* xh1 = s.xh + (size_t)b * 4 * H;
    float* xh2 = xh1 + 2 * H;
    for (int i = tid; i < 2 * H; i += 512) sm[i] = 0.f;
    for (int i = tid; i < M; i += 512) lm[i] = a.init_mel;
    float c1 = 0.f, c2 = 0.f;
    __syncthreads();
    // this thread's slice of a weight matrix: k-slice ks (Kslice inputs) of gate rows j, j + H, ...; uniform base + one 32-bit lane offset
    auto chain = [&](float (&acc)[1][4], const float* w, int Kslice, const float* v, int vstride) __attribute__((always_inline)) {
        if ((Kslice >> 2) % 2 == 0)
            lstm_chain_u<1, 4, 2>(acc, w, (unsigned)((ks * Kslice / 4) * H4 + j), H4, H, v, vstride, Kslice);
        else
            lstm_chain<1, 4, 2>(acc, w + (size_t)(ks * Kslice / 4) * H4 * 4, H4, H, j, v, vstride, Kslice);
    };
    for (int t = 0; t < nsteps; ++t) {
        const int par = t & 1;
        auto mask = [&](int layer, int unit) -> float {
            if (a.masks) return a.masks[(((size_t)b * a.S + t) * 2 + layer) * P + unit];
            uint32_t r4[4];
            ttsc_philox4x32((uint32_t)(unit >> 2), (uint32_t)t, (uint32_t)b, (uint32_t)layer, (uint32_t)a.seed, (uint32_t)(a.seed >> 32), r4);
            return (r4[unit & 3] & 1u) ? 1.f : 0.f;
        };
        float xv[4] = {0.f, 0.f, 0.f, 0.f};
        if (owner) {
            const float* xr = a.xg1 + ((size_t)b * a.S + t) * H4 + j;
#pragma unroll
            for (int g = 0; g < 4; ++g) xv[g] = xr[g * H];
        }
        // ---- PreNet (every member, identical arithmetic to melar_kernel) ----
        if (tid < P) {
            float acc[1] = {a.b_pn1[tid]};
            ar_chain<1, 5>(acc, a.w_pn1, P, 0, tid, lm, M);
            p1[tid] = fmaxf(acc[0], 0.f) * (mask(0, tid) * 2.f);
        }
        __syncthreads();
        if (tid < P) {
            float acc[1] = {a.b_pn2[tid]};
            ar_chain<1, 8>(acc, a.w_pn2, P, 0, tid, p1, P);
            p2[tid] = fmaxf(acc[0], 0.f) * (mask(1, tid) * 2.f);
        }
        __syncthreads();
        // ---- LSTM layer 1, this member's units: partial sums over the k-slices of p2 and h1_{t-1} ----
        {
            float acc[1][4] = {{0.f, 0.f, 0.f, 0.f}};
            chain(acc, a.w_p2l, KP, p2 + ks * KP, P);
            chain(acc, a.w_hh1, KH, h1 + ks * KH, H);
#pragma unroll
            for (int g = 0; g < 4; ++g) part[(ks * 4 + g) * HU + u] = acc[0][g];
        }
        __syncthreads();
        if (owner) {
            float gs[4] = {xv[0], xv[1], xv[2], xv[3]};
            for (int q = 0; q < KS; ++q)
#pragma unroll
                for (int g = 0; g < 4; ++g) gs[g] += part[(q * 4 + g) * HU + u];
            const float ig = ttsc_sigmoidf(gs[0]), fg = ttsc_sigmoidf(gs[1]), gg = ttsc_tanhf(gs[2]), og = ttsc_sigmoidf(gs[3]);
            c1 = fmaf(fg, c1, ig * gg);
            g_st(xh1 + par * H + j, og * ttsc_tanhf(c1));
        }
        g_publish(cnt);
        if (!g_wait(cnt, (unsigned)(2 * t + 1) * (unsigned)s.G, s.abort_word, &ok_s)) return;
        for (int i = tid; i < H; i += 512) h1[i] = g_ld(xh1 + par * H + i);
        __syncthreads();
        // ---- LSTM layer 2 ----
        {
            float acc[1][4] = {{0.f, 0.f, 0.f, 0.f}};
            chain(acc, a.w_ih2, KH, h1 + ks * KH, H);
            chain(acc, a.w_hh2, KH, h2 + ks * KH, H);
#pragma unroll
            for (int g = 0; g < 4; ++g) part[(ks * 4 + g) * HU + u] = acc[0][g];
        }
        __syncthreads();
        if (owner) {
            float gs[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) gs[g] = a.b2[g * H + j];
            for (int q = 0; q < KS; ++q)
#pragma unroll
                for (int g = 0; g < 4; ++g) gs[g] += part[(q * 4 + g) * HU + u];
            const float ig = ttsc_sigmoidf(gs[0]), fg = ttsc_sigmoidf(gs[1]), gg = ttsc_tanhf(gs[2]), og = ttsc_sigmoidf(gs[3]);
            c2 = fmaf(fg, c2, ig * gg);
            g_st(xh2 + par * H + j, og * ttsc_tanhf(c2));
        }
        g_publish(cnt);
        if (!g_wait(cnt, (unsigned)(2 * t + 2) * (unsigned)s.G, s.abort_word, &ok_s)) return;
        for (int i = tid; i < H; i += 512) h2[i] = g_ld(xh2 + par * H + i);
        __syncthreads();
        // ---- output Linear H -> O (every member; member 0 stores); the last M values are fed back ----
        if (tid < O) {
            float acc[1] = {a.b_out[tid]};
            ar_chain<1, 8>(acc, a.w_out, O, 0, tid, h2, H);
            if (m == 0) a.y[((size_t)b * a.S + t) * O + tid] = acc[0];
            if (tid >= O - M) lm[tid - (O - M)] = acc[0];
        }
        __syncthreads();
    }
    if (m == 0)
        for (int t = nsteps; t < a.S; ++t)
            for (int i = tid; i < O; i += 512) a.y[((size_t)b * a.S + t) * O + i] = 0.f;
}

}  // namespace ttsc

using namespace ttsc;

struct ttsc_melar {
    int H = 512, P = 256, M = 80, O = 240;
    float *w_p2l = nullptr, *w_hh1 = nullptr, *w_ih2 = nullptr, *w_hh2 = nullptr, *b2 = nullptr, *w_out = nullptr, *b_out = nullptr;
    float *w_pn1 = nullptr, *b_pn1 = nullptr, *w_pn2 = nullptr, *b_pn2 = nullptr;
};

static int up(float** dst, const float* host, size_t n) {
    if (*dst) (void)hipFree(*dst);
    *dst = nullptr;
    TTSC_HIP_CHECK(hipMalloc((void**)dst, n * sizeof(float)));
    TTSC_HIP_CHECK(hipMemcpy(*dst, host, n * sizeof(float), hipMemcpyHostToDevice));
    return TTSC_OK;
}

// host [rows, ld] (columns c0 .. c0+K) -> device [K/4][rows][4]
static int up_packed4(float** dst, const float* host, int64_t rows, int64_t ld, int64_t c0, int64_t K) {
    std::vector<float> t((size_t)rows * K);
    for (int64_t r = 0; r < rows; ++r)
        for (int64_t k = 0; k < K; ++k) t[((size_t)(k >> 2) * rows + r) * 4 + (k & 3)] = host[(size_t)r * ld + c0 + k];
    return up(dst, t.data(), t.size());
}

extern "C" int ttsc_melar_create(int32_t H, int32_t P, int32_t M, int32_t O, ttsc_melar** out) {
    TTSC_REQUIRE(out, "ttsc_melar_create: null argument");
    TTSC_REQUIRE(H > 0 && H <= 512 && H % 16 == 0, "ttsc_melar_create: LSTM size must be a multiple of 16 and <= 512 (got %d)", H);
    TTSC_REQUIRE(P > 0 && P <= 512 && P % 32 == 0 && M > 0 && M % 20 == 0 && O >= M && O <= 512,
                 "ttsc_melar_create: unsupported sizes P=%d M=%d O=%d", P, M, O);
    ttsc_melar* m = new ttsc_melar();
    m->H = H; m->P = P; m->M = M; m->O = O;
    *out = m;
    return TTSC_OK;
}

extern "C" void ttsc_melar_destroy(ttsc_melar* m) {
    if (!m) return;
    for (float* p : {m->w_p2l, m->w_hh1, m->w_ih2, m->w_hh2, m->b2, m->w_out, m->b_out, m->w_pn1, m->b_pn1, m->w_pn2, m->b_pn2})
        if (p) (void)hipFree(p);
    delete m;
}

// All weights in torch layout (host fp32): w_ih1 [4H, overlay+P] (only its last P columns are used here; `ld1` = its row
// length), w_hh1 [4H,H], w_ih2 [4H,H], w_hh2 [4H,H], b_ih2/b_hh2 [4H], w_out [O,H], b_out [O], prenet w1 [P,M], b1 [P], w2 [P,P], b2 [P]
extern "C" int ttsc_melar_set_weights(ttsc_melar* m, const float* w_ih1, int64_t ld1, const float* w_hh1, const float* w_ih2,
                                      const float* w_hh2, const float* b_ih2, const float* b_hh2, const float* w_out,
                                      const float* b_out, const float* pn_w1, const float* pn_b1, const float* pn_w2,
                                      const float* pn_b2) {
    TTSC_REQUIRE(m && w_ih1 && w_hh1 && w_ih2 && w_hh2 && b_ih2 && b_hh2 && w_out && b_out && pn_w1 && pn_b1 && pn_w2 && pn_b2,
                 "ttsc_melar_set_weights: null argument");
    TTSC_REQUIRE(ld1 >= m->P, "ttsc_melar_set_weights: w_ih1 row length < prenet size");
    const int H = m->H, P = m->P;
    int rc;
    if ((rc = up_packed4(&m->w_p2l, w_ih1, 4 * H, ld1, ld1 - P, P))) return rc;
    if ((rc = up_packed4(&m->w_hh1, w_hh1, 4 * H, H, 0, H))) return rc;
    if ((rc = up_packed4(&m->w_ih2, w_ih2, 4 * H, H, 0, H))) return rc;
    if ((rc = up_packed4(&m->w_hh2, w_hh2, 4 * H, H, 0, H))) return rc;
    std::vector<float> b(4 * H);
    for (int i = 0; i < 4 * H; ++i) b[i] = b_ih2[i] + b_hh2[i];
    if ((rc = up(&m->b2, b.data(), b.size()))) return rc;
    if ((rc = up_packed4(&m->w_out, w_out, m->O, H, 0, H))) return rc;
    if ((rc = up(&m->b_out, b_out, m->O))) return rc;
    if ((rc = up_packed4(&m->w_pn1, pn_w1, P, m->M, 0, m->M))) return rc;
    if ((rc = up(&m->b_pn1, pn_b1, P))) return rc;
    if ((rc = up_packed4(&m->w_pn2, pn_w2, P, P, 0, P))) return rc;
    return up(&m->b_pn2, pn_b2, P);
}


// split factor for the AR decoder: 15.5 MB of weights per step make G = 8 the measured optimum for one utterance
// (G = 1 / 2 / 4 / 8: 12.1 / 9.5 / 8.0 / 5.9 ms for 58 steps), >= 32 units per member, all workgroups co-resident
static int melar_split_members(int B, int H, int P) {
    const int cus = device_cus();   // of the CURRENT device
    int gmax = 8;
    if (const char* ev = getenv("TTSC_MELAR_SPLIT")) gmax = atoi(ev);
    int G = 1;
    while (G * 2 <= gmax && (long)G * 2 * B <= cus && H % (G * 2) == 0 && H / (G * 2) >= 32 && 512 % (H / (G * 2)) == 0) {
        const int HU = H / (G * 2), KS = 512 / HU;
        if (H % KS != 0 || (H / KS) % 8 != 0 || P % KS != 0 || (P / KS) % 8 != 0) break;
        G *= 2;
    }
    return G;
}

// 0 = every hand-off of the split AR launches on this device since the last call completed, 1 = a bounded spin timed out
// (sticky until read).  Synchronises the device.  The exchange area [B][4][H] + counters live in one HandoffArea per (device, stream).
extern "C" int32_t ttsc_melar_split_status(void) { return handoff_status("melar"); }

extern "C" int ttsc_melar_decode(const ttsc_melar* m, const float* xg1_dev, int32_t B, int32_t S, const float* masks_dev,
                                 uint64_t seed, const int32_t* steps_dev, float* y_dev, void* stream) {
    TTSC_REQUIRE(m && xg1_dev && y_dev, "ttsc_melar_decode: null argument");
    TTSC_REQUIRE(B > 0 && S > 0, "ttsc_melar_decode: bad B/S");
    if (!m->w_pn2) {
        set_error("ttsc_melar_decode: weights not set");
        return TTSC_ESTATE;
    }
    MelArArgs a{xg1_dev, m->w_p2l, m->w_hh1, m->w_ih2, m->w_hh2, m->b2, m->w_out, m->b_out, m->w_pn1, m->b_pn1, m->w_pn2, m->b_pn2,
                masks_dev, steps_dev, y_dev, B, S, m->H, m->P, m->M, m->O, -5.0f, seed};
    const int G = B <= 1024 ? melar_split_members(B, m->H, m->P) : 1;
    if (G > 1) {
        hipStream_t st = (hipStream_t)stream;
        HandoffArea* ar = handoff_area("melar", st, 1024, (size_t)B * 4 * 512 * sizeof(float));
        TTSC_REQUIRE(ar, "ttsc_melar_decode: cannot allocate the exchange area");
        TTSC_HIP_CHECK(ar->rearm(st));   // counters and this launch's abort word restart at zero; the sticky copy stays
        MelArSplitArgs sa{};
        sa.a = a;
        sa.xh = reinterpret_cast<float*>(ar->buf);
        sa.cnt = ar->words;
        sa.abort_word = ar->abort_word();
        sa.G = G;
        sa.HU = m->H / G;
        sa.KS = 512 / sa.HU;
        // the exchange area is indexed with H, not 512
        const size_t ldsb = ((size_t)2 * m->H + 2 * m->P + m->M + (size_t)sa.KS * 4 * sa.HU + 16) * sizeof(float);
        hipLaunchKernelGGL(melar_split_kernel, dim3((unsigned)G, (unsigned)B), dim3(512), ldsb, st, sa);
        hipError_t e2 = hipGetLastError();
        if (e2 != hipSuccess) {
            set_error("melar_split_kernel launch failed: %s", hipGetErrorString(e2));
            return TTSC_EHIP;
        }
        return TTSC_OK;
    }
    const size_t lds = ((size_t)4 * m->H + 2 * m->P + m->M + 16) * sizeof(float);
    hipLaunchKernelGGL(melar_kernel, dim3(B), dim3(512), lds, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("melar_kernel launch failed: %s", hipGetErrorString(e));
        return TTSC_EHIP;
    }
    return TTSC_OK;
}
