// GRU sequence recurrence + backward through time for TRAINING the WaveRNN vocoder (SURVEY.md §8 row a9:
// `WaveRNN._train_forward` cube/networks/modules.py:505-539 runs torch.nn.GRU over the whole teacher-forced sequence,
// 24 000 steps per utterance; `training_step` 553-563 differentiates it) — gfx950.
//
// torch.nn.GRU semantics (gate order r,z,n):  r = s(xr + W_hr h + b_hr), z = s(xz + W_hz h + b_hz),
// n = tanh(xn + r * (W_hn h + b_hn)), h' = (1 - z) * n + z * h, with x* = W_i* x + b_i* hoisted into one GEMM for all steps.
// Same shape as lstm.hip: one persistent workgroup per utterance, thread j owns hidden unit j (its three gate rows are
// streamed from L2 as 16-byte packed loads, h lives in LDS double-buffered), one barrier per step.  The training forward
// saves r, z, n and the linear part hn = W_hn h + b_hn; the backward kernel walks the steps in reverse, publishes the three
// hidden-side gate gradients in LDS and evaluates its own dh_prev[k] = z*dh + sum_r W_hh[r,k] * dGh[r] as one chain over
// 3H rows of the transposed packing.  Weight / input gradients are plain GEMMs over the saved per-step gate gradients.
#include <algorithm>

#include "common.hpp"
#include "../../include/ttscube_math.h"
#include "rnn_chain.hpp"

namespace ttsc {

struct GruArgs {
    const float* xg;     // [B, T, 3H]   W_ih x + b_ih
    const float* whh;    // [H/4][3H][4]
    const float* bhh;    // [3H]
    float* y;            // [B, T, H]
    float* saved;        // [B, T, 4H]   r, z, n, hn  (training) or null
    const float* h_0;    // [B, H] or null
    int B, T, H;
};

__global__ __launch_bounds__(512) void gru_seq_kernel(GruArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];  // h[2][H]
    const int H = a.H, H3 = 3 * H;
    const int j = threadIdx.x;
    const bool unit = j < H;
    const int b = blockIdx.x;
    if (unit) sm[j] = a.h_0 ? a.h_0[(size_t)b * H + j] : 0.f;
    float br = 0.f, bz = 0.f, bn = 0.f;
    if (unit) {
        br = a.bhh[j];
        bz = a.bhh[H + j];
        bn = a.bhh[2 * H + j];
    }
    __syncthreads();
    int cur = 0;
    const float* xb = a.xg + (size_t)b * a.T * H3 + j;
    float xr = 0.f, xz = 0.f, xn = 0.f;
    if (unit) {
        xr = xb[0];
        xz = xb[H];
        xn = xb[2 * H];
    }
    for (int t = 0; t < a.T; ++t) {
        const float* hc = sm + cur * H;
        float* hn = sm + (cur ^ 1) * H;
        if (unit) {
            float acc[1][3] = {{br, bz, bn}};
            const float cxr = xr, cxz = xz, cxn = xn;
            if (t + 1 < a.T) {   // next step's input projection, in flight during the chain
                const float* xp = xb + (size_t)(t + 1) * H3;
                xr = xp[0];
                xz = xp[H];
                xn = xp[2 * H];
            }
            lstm_chain<1, 3, 2>(acc, a.whh, H3, H, j, hc, H, H);
            const float r = ttsc_sigmoidf(cxr + acc[0][0]);
            const float z = ttsc_sigmoidf(cxz + acc[0][1]);
            const float n = ttsc_tanhf(fmaf(r, acc[0][2], cxn));
            const float hv = fmaf(z, hc[j] - n, n);   // (1 - z) * n + z * h
            hn[j] = hv;
            a.y[((size_t)b * a.T + t) * H + j] = hv;
            if (a.saved) {
                float* sp = a.saved + ((size_t)b * a.T + t) * (4 * (size_t)H) + j;
                sp[0] = r;
                sp[H] = z;
                sp[2 * H] = n;
                sp[3 * H] = acc[0][2];
            }
        }
        __syncthreads();
        cur ^= 1;
    }
}

struct GruBwdArgs {
    const float* dy;      // [B, T, H]
    const float* saved;   // [B, T, 4H]
    const float* y;       // [B, T, H]  (h_t; h_{-1} = h_0 or zero)
    const float* h_0;     // [B, H] or null
    const float* whhT;    // [3H/4][H][4]
    float* dgi;           // [B, T, 3H]  gradient wrt (W_ih x + b_ih):  dr_pre, dz_pre, dn_pre
    float* dgh;           // [B, T, 3H]  gradient wrt (W_hh h + b_hh):  dr_pre, dz_pre, dn_pre * r
    int B, T, H;
};

__global__ __launch_bounds__(512) void gru_bwd_kernel(GruBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];  // dGh[3H]
    const int H = a.H, H3 = 3 * H;
    const int j = threadIdx.x;
    const bool unit = j < H;
    const int b = blockIdx.x;
    const float* sb = a.saved + (size_t)b * a.T * (4 * (size_t)H) + j;
    const float* yb = a.y + (size_t)b * a.T * H + j;
    const float* dyb = a.dy + (size_t)b * a.T * H + j;
    float* gib = a.dgi + (size_t)b * a.T * H3 + j;
    float* ghb = a.dgh + (size_t)b * a.T * H3 + j;
    const float h0 = (unit && a.h_0) ? a.h_0[(size_t)b * H + j] : 0.f;
    float dh_rec = 0.f;
    float r = 0.f, z = 0.f, n = 0.f, hl = 0.f, hp = 0.f, dyv = 0.f;
    auto fetch = [&](int t) __attribute__((always_inline)) {
        const float* s = sb + (size_t)t * (4 * (size_t)H);
        r = s[0];
        z = s[H];
        n = s[2 * H];
        hl = s[3 * H];
        hp = t > 0 ? yb[(size_t)(t - 1) * H] : h0;
        dyv = dyb[(size_t)t * H];
    };
    if (unit) fetch(a.T - 1);
    for (int t = a.T - 1; t >= 0; --t) {
        float dh_direct = 0.f;
        if (unit) {
            const float dh = dyv + dh_rec;
            const float dn_pre = dh * (1.f - z) * (1.f - n * n);
            const float dz_pre = dh * (hp - n) * z * (1.f - z);
            const float dr_pre = dn_pre * hl * r * (1.f - r);
            const float dhn = dn_pre * r;
            dh_direct = dh * z;
            float* gi = gib + (size_t)t * H3;
            gi[0] = dr_pre;
            gi[H] = dz_pre;
            gi[2 * H] = dn_pre;
            float* gh = ghb + (size_t)t * H3;
            gh[0] = dr_pre;
            gh[H] = dz_pre;
            gh[2 * H] = dhn;
            sm[j] = dr_pre;
            sm[H + j] = dz_pre;
            sm[2 * H + j] = dhn;
        }
        __syncthreads();
        if (unit) {
            if (t > 0) fetch(t - 1);
            float acc[1][1] = {{dh_direct}};
            lstm_chain<1, 1, 4>(acc, a.whhT, H, 0, j, sm, H3, H3);
            dh_rec = acc[0][0];
        }
        __syncthreads();
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Split recurrence: G workgroups ("members") per utterance.  With B = 16 utterances the kernels above keep 16 of the 256
// CUs busy, each bound by ONE CU's L2 load path (3 MB of W_hh per step ~ 20 us).  Here member m owns H/G hidden units and
// streams only its 1/G of the rows; the 512 threads of a member are (unit u, k-slice ks) pairs whose partial sums meet in
// LDS.  Per step the members exchange the full state through the output tensors themselves (y for the forward, dgh for the
// backward: write-through agent-scope stores, one monotonic counter per utterance, bounded spins with a shared abort word —
// the hand-off protocol of wavernn_cluster.hip).  All G*B workgroups must be co-resident: the host only takes this path
// when G*B <= number of CUs.
struct GruSplitArgs {
    GruArgs f;
    GruBwdArgs bw;
    unsigned* cnt;     // [B] monotonic counters (zeroed by the host before the launch)
    unsigned* abort_word;
    int G, HU, KS;     // members per utterance, units per member, k-slices per unit (HU * KS = 512 threads)
};

__global__ __launch_bounds__(512) void gru_seq_split_kernel(GruSplitArgs s) {
    extern __shared__ __attribute__((aligned(16))) float sm[];   // h[H] | part[KS][3][HU]
    __shared__ int ok_s;
    const GruArgs& a = s.f;
    const int H = a.H, H3 = 3 * H, HU = s.HU, KS = s.KS;
    const int tid = threadIdx.x, u = tid % HU, ks = tid / HU;
    const int m = blockIdx.x, b = blockIdx.y;
    const int j = m * HU + u;                 // hidden unit of this thread
    const int KL = H / KS;                    // inputs per k-slice
    float* hs = sm;
    float* part = sm + H;
    unsigned* cnt = s.cnt + b;
    const bool owner = ks == 0;
    float br = 0.f, bz = 0.f, bn = 0.f;
    if (owner) {
        br = a.bhh[j];
        bz = a.bhh[H + j];
        bn = a.bhh[2 * H + j];
    }
    const float* xb = a.xg + (size_t)b * a.T * H3 + j;
    float* yb = a.y + (size_t)b * a.T * H;
    for (int t = 0; t < a.T; ++t) {
        float xr = 0.f, xz = 0.f, xn = 0.f;
        if (owner) {   // issued before the wait: in flight while the other members finish step t-1
            const float* xp = xb + (size_t)t * H3;
            xr = xp[0];
            xz = xp[H];
            xn = xp[2 * H];
        }
        if (t > 0) {
            if (!g_wait(cnt, (unsigned)t * (unsigned)s.G, s.abort_word, &ok_s)) return;
            for (int i = tid; i < H; i += 512) hs[i] = g_ld(yb + (size_t)(t - 1) * H + i);
        } else {
            for (int i = tid; i < H; i += 512) hs[i] = a.h_0 ? a.h_0[(size_t)b * H + i] : 0.f;
        }
        __syncthreads();
        float acc[1][3] = {{0.f, 0.f, 0.f}};
        lstm_chain<1, 3, 2>(acc, a.whh + (size_t)(ks * KL / 4) * H3 * 4, H3, H, j, hs + ks * KL, H, KL);
        part[(ks * 3 + 0) * HU + u] = acc[0][0];
        part[(ks * 3 + 1) * HU + u] = acc[0][1];
        part[(ks * 3 + 2) * HU + u] = acc[0][2];
        __syncthreads();
        if (owner) {
            float hr = br, hz = bz, hl = bn;
            for (int q = 0; q < KS; ++q) {
                hr += part[(q * 3 + 0) * HU + u];
                hz += part[(q * 3 + 1) * HU + u];
                hl += part[(q * 3 + 2) * HU + u];
            }
            const float r = ttsc_sigmoidf(xr + hr);
            const float z = ttsc_sigmoidf(xz + hz);
            const float n = ttsc_tanhf(fmaf(r, hl, xn));
            const float hv = fmaf(z, hs[j] - n, n);
            g_st(yb + (size_t)t * H + j, hv);
            if (a.saved) {
                float* sp = a.saved + ((size_t)b * a.T + t) * (4 * (size_t)H) + j;
                sp[0] = r;
                sp[H] = z;
                sp[2 * H] = n;
                sp[3 * H] = hl;
            }
        }
        g_publish(cnt);
    }
}

__global__ __launch_bounds__(512) void gru_bwd_split_kernel(GruSplitArgs s) {
    extern __shared__ __attribute__((aligned(16))) float sm[];   // dGh[3H] | part[KS][HU]
    __shared__ int ok_s;
    const GruBwdArgs& a = s.bw;
    const int H = a.H, H3 = 3 * H, HU = s.HU, KS = s.KS;
    const int tid = threadIdx.x, u = tid % HU, ks = tid / HU;
    const int m = blockIdx.x, b = blockIdx.y;
    const int j = m * HU + u;
    const int KL = H3 / KS;
    float* dg = sm;
    float* part = sm + H3;
    unsigned* cnt = s.cnt + b;
    const bool owner = ks == 0;
    const float* sb = a.saved + (size_t)b * a.T * (4 * (size_t)H) + j;
    const float* yb = a.y + (size_t)b * a.T * H + j;
    const float* dyb = a.dy + (size_t)b * a.T * H + j;
    float* gib = a.dgi + (size_t)b * a.T * H3;
    float* ghb = a.dgh + (size_t)b * a.T * H3;
    const float h0 = (owner && a.h_0) ? a.h_0[(size_t)b * H + j] : 0.f;
    float dh_rec = 0.f;
    unsigned step = 0;
    for (int t = a.T - 1; t >= 0; --t) {
        float dh_direct = 0.f;
        if (owner) {
            const float* sv = sb + (size_t)t * (4 * (size_t)H);
            const float r = sv[0], z = sv[H], n = sv[2 * H], hl = sv[3 * H];
            const float hp = t > 0 ? yb[(size_t)(t - 1) * H] : h0;
            const float dh = dyb[(size_t)t * H] + dh_rec;
            const float dn_pre = dh * (1.f - z) * (1.f - n * n);
            const float dz_pre = dh * (hp - n) * z * (1.f - z);
            const float dr_pre = dn_pre * hl * r * (1.f - r);
            dh_direct = dh * z;
            float* gi = gib + (size_t)t * H3 + j;
            gi[0] = dr_pre;
            gi[H] = dz_pre;
            gi[2 * H] = dn_pre;
            float* gh = ghb + (size_t)t * H3 + j;
            g_st(gh, dr_pre);
            g_st(gh + H, dz_pre);
            g_st(gh + 2 * H, dn_pre * r);
        }
        g_publish(cnt);
        ++step;
        if (t == 0) break;   // dh_{-1} is not needed
        if (!g_wait(cnt, step * (unsigned)s.G, s.abort_word, &ok_s)) return;
        for (int i = tid; i < H3; i += 512) dg[i] = g_ld(ghb + (size_t)t * H3 + i);
        __syncthreads();
        float acc[1][1] = {{0.f}};
        lstm_chain<1, 1, 4>(acc, a.whhT + (size_t)(ks * KL / 4) * H * 4, H, 0, j, dg + ks * KL, H3, KL);
        part[ks * HU + u] = acc[0][0];
        __syncthreads();
        if (owner) {
            float v = dh_direct;
            for (int q = 0; q < KS; ++q) v += part[q * HU + u];
            dh_rec = v;
        }
        __syncthreads();   // part / dg are rewritten in the next step
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The same split with the member's weights RESIDENT IN REGISTERS and the state handed over as 8-byte {value, step tag} granules that the
// consumers poll directly (the scheme of lstm_seq_split_res_kernel / lstm_bwd_split_res_kernel): H = 512 over 16 members (all 256 CUs for 16
// utterances) or H = 256 over 4 — 3 x 32 forward weights / 96 backward weights per thread.  The kernels above stream 768 KB (G = 4) of W_hh per
// member and step from L2 and pay counter + data round trips: 8 + 12 us per time step at H = 512, i.e. 0.48 s for the vocoder's 24 000-step
// training sequences.
typedef unsigned long long gru_u64;
typedef float gru_f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ bool gru_poll(const gru_u64* src, unsigned tag, unsigned* abort_word, float* out) {
    gru_u64 gq;
    unsigned spins = 0;
    for (;;) {
        gq = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(gq >> 32) == tag) break;
        if (++spins > GS_SPIN_LIMIT || ((spins & 63u) == 0u && __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
            __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(abort_word + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sticky copy
            return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    *out = __uint_as_float((unsigned)gq);
    return true;
}

// Up to N granules per thread (src + i0 + r * stride for r < n) in ONE round trip: all loads are issued before the first tag is looked at, and only a pass in
// which some granule is still missing is repeated.  gru_poll in a loop pays the L2 round trip once per granule even when every granule has arrived (the backward
// recurrence hands over 3H values to 512 threads: three dependent round trips per step).
template <int N>
__device__ __forceinline__ bool gru_poll_n(const gru_u64* src, int i0, int stride, int n, unsigned tag, unsigned* abort_word, float* out) {
    gru_u64 g[N];
    unsigned spins = 0;
    for (;;) {
#pragma unroll
        for (int r = 0; r < N; ++r)
            if (r < n) g[r] = __hip_atomic_load(src + i0 + r * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool all = true;
#pragma unroll
        for (int r = 0; r < N; ++r)
            if (r < n) all = all && ((unsigned)(g[r] >> 32) == tag);
        if (all) break;
        if (++spins > GS_SPIN_LIMIT || ((spins & 63u) == 0u && __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
            __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(abort_word + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sticky copy
            return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll
    for (int r = 0; r < N; ++r)
        if (r < n) out[i0 + r * stride] = __uint_as_float((unsigned)g[r]);
    return true;
}

template <int KL>   // inputs per k-slice = H / KS
__global__ __launch_bounds__(512) void gru_seq_split_res_kernel(GruSplitArgs s, gru_u64* ring) {
    extern __shared__ __attribute__((aligned(16))) float sm[];   // h[H] | part[KS][3][HU] | act[3][HU]
    const GruArgs& a = s.f;
    const int H = a.H, H3 = 3 * H, HU = s.HU, KS = s.KS;
    const int tid = threadIdx.x, u = tid % HU, ks = tid / HU;
    const int m = blockIdx.x, b = blockIdx.y;
    const int j = m * HU + u;
    float* hs = sm;
    float* part = sm + H;
    float* act = part + KS * 3 * HU;   // [3][HU]: r, z, W_hn h + b_hn of this step
    gru_u64* rg = ring + (size_t)b * 2 * H;
    const bool owner = ks == 0;
    // gates r and z side by side: one v_pk_fma_f32 advances both k-ordered chains (per lane the fused multiply-add of the scalar form: same bits)
    gru_f32x2 wrz[KL];
    float wn[KL];
    {
        // packed [H/4][3H][4]: row g*H + j, k-block kb holds k = 4*kb .. 4*kb+3
        const float4* w4 = reinterpret_cast<const float4*>(a.whh) + j;
#pragma unroll
        for (int kb = 0; kb < KL / 4; ++kb) {
            const float4 vr = w4[(size_t)(ks * (KL / 4) + kb) * H3];
            const float4 vz = w4[(size_t)(ks * (KL / 4) + kb) * H3 + H];
            const float4 vn = w4[(size_t)(ks * (KL / 4) + kb) * H3 + 2 * H];
            wrz[4 * kb] = gru_f32x2{vr.x, vz.x};
            wrz[4 * kb + 1] = gru_f32x2{vr.y, vz.y};
            wrz[4 * kb + 2] = gru_f32x2{vr.z, vz.z};
            wrz[4 * kb + 3] = gru_f32x2{vr.w, vz.w};
            wn[4 * kb] = vn.x;
            wn[4 * kb + 1] = vn.y;
            wn[4 * kb + 2] = vn.z;
            wn[4 * kb + 3] = vn.w;
        }
    }
    // k-slice row g (ks = 0, 1, 2) finishes gate g of the member's units: bias + the KS partial sums in slice order (+ the sigmoid for r and z) — three rows side
    // by side instead of row 0 walking all three; row 0 then forms the candidate and h.  Per value the same operations in the same order: same bits.
    float bg = 0.f;
    if (ks < 3) bg = a.bhh[ks * H + j];
    const float* xb = a.xg + (size_t)b * a.T * H3 + j;
    float* yb = a.y + (size_t)b * a.T * H;
    for (int t = 0; t < a.T; ++t) {
        float xg = 0.f, xg_n = 0.f;   // issued before the wait: in flight while the other members finish step t-1
        if (ks < 2) xg = xb[(size_t)t * H3 + ks * H];
        if (owner) xg_n = xb[(size_t)t * H3 + 2 * H];
        bool fail = false;
        if (t > 0) {
            const gru_u64* src = rg + (size_t)((t - 1) & 1) * H;
            for (int i = tid; i < H; i += 512) fail = !gru_poll(src + i, (unsigned)t, s.abort_word, &hs[i]) || fail;
        } else {
            for (int i = tid; i < H; i += 512) hs[i] = a.h_0 ? a.h_0[(size_t)b * H + i] : 0.f;
        }
        if (__syncthreads_or(fail)) return;
        const float hprev = owner ? hs[j] : 0.f;   // (hs is rewritten by the next step's poll while the owners are still combining)
        gru_f32x2 arz = {0.f, 0.f};
        float an = 0.f;
        {
            const float4* h4 = reinterpret_cast<const float4*>(hs + ks * KL);
#pragma unroll
            for (int kb = 0; kb < KL / 4; ++kb) {
                const float4 hv = h4[kb];
                arz = __builtin_elementwise_fma(wrz[4 * kb], (gru_f32x2){hv.x, hv.x}, arz);
                arz = __builtin_elementwise_fma(wrz[4 * kb + 1], (gru_f32x2){hv.y, hv.y}, arz);
                arz = __builtin_elementwise_fma(wrz[4 * kb + 2], (gru_f32x2){hv.z, hv.z}, arz);
                arz = __builtin_elementwise_fma(wrz[4 * kb + 3], (gru_f32x2){hv.w, hv.w}, arz);
                an = fmaf(wn[4 * kb], hv.x, an);
                an = fmaf(wn[4 * kb + 1], hv.y, an);
                an = fmaf(wn[4 * kb + 2], hv.z, an);
                an = fmaf(wn[4 * kb + 3], hv.w, an);
            }
        }
        part[(ks * 3 + 0) * HU + u] = arz.x;
        part[(ks * 3 + 1) * HU + u] = arz.y;
        part[(ks * 3 + 2) * HU + u] = an;
        __syncthreads();
        if (ks < 3) {
            float hg = bg;
            for (int q = 0; q < KS; ++q) hg += part[(q * 3 + ks) * HU + u];
            act[ks * HU + u] = ks == 2 ? hg : ttsc_sigmoidf(xg + hg);
        }
        __syncthreads();
        if (owner) {
            const float r = act[u], z = act[HU + u], hl = act[2 * HU + u];
            const float n = ttsc_tanhf(fmaf(r, hl, xg_n));
            const float hv = fmaf(z, hprev - n, n);
            __hip_atomic_store(rg + (size_t)(t & 1) * H + j, ((gru_u64)(unsigned)(t + 1) << 32) | (gru_u64)__float_as_uint(hv), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            yb[(size_t)t * H + j] = hv;
            if (a.saved) {
                float* sp = a.saved + ((size_t)b * a.T + t) * (4 * (size_t)H) + j;
                sp[0] = r;
                sp[H] = z;
                sp[2 * H] = n;
                sp[3 * H] = hl;
            }
        }
    }
}

template <int KL>   // gate rows per k-slice = 3H / KS
__global__ __launch_bounds__(512) void gru_bwd_split_res_kernel(GruSplitArgs s, gru_u64* ring) {
    extern __shared__ __attribute__((aligned(16))) float sm[];   // dGh[3H] | part[KS][HU]
    const GruBwdArgs& a = s.bw;
    const int H = a.H, H3 = 3 * H, HU = s.HU, KS = s.KS;
    const int tid = threadIdx.x, u = tid % HU, ks = tid / HU;
    const int m = blockIdx.x, b = blockIdx.y;
    const int j = m * HU + u;
    float* dg = sm;
    float* part = sm + H3;
    gru_u64* rg = ring + (size_t)b * 2 * H3;
    const bool owner = ks == 0;
    float w[KL];
    {
        // transposed pack [3H/4][H][4]: k-block kb of this slice holds gate rows ks*KL + 4*kb .. + 3 of column j
        const float4* w4 = reinterpret_cast<const float4*>(a.whhT + (size_t)(ks * (KL / 4)) * H * 4) + j;
#pragma unroll
        for (int kb = 0; kb < KL / 4; ++kb) {
            const float4 v = w4[(size_t)kb * H];
            w[4 * kb] = v.x;
            w[4 * kb + 1] = v.y;
            w[4 * kb + 2] = v.z;
            w[4 * kb + 3] = v.w;
        }
    }
    const float* sb = a.saved + (size_t)b * a.T * (4 * (size_t)H) + j;
    const float* yb = a.y + (size_t)b * a.T * H + j;
    const float* dyb = a.dy + (size_t)b * a.T * H + j;
    float* gib = a.dgi + (size_t)b * a.T * H3;
    float* ghb = a.dgh + (size_t)b * a.T * H3;
    const float h0 = (owner && a.h_0) ? a.h_0[(size_t)b * H + j] : 0.f;
    float dh_rec = 0.f;
    float pr = 0.f, pz = 0.f, pn = 0.f, phl = 0.f, php = 0.f, pdy = 0.f;   // saved state of the step about to be differentiated
    auto fetch = [&](int t) __attribute__((always_inline)) {
        const float* sv = sb + (size_t)t * (4 * (size_t)H);
        pr = sv[0];
        pz = sv[H];
        pn = sv[2 * H];
        phl = sv[3 * H];
        php = t > 0 ? yb[(size_t)(t - 1) * H] : h0;
        pdy = dyb[(size_t)t * H];
    };
    if (owner) fetch(a.T - 1);
    for (int t = a.T - 1; t >= 0; --t) {
        const unsigned tag = (unsigned)(a.T - t);
        gru_u64* slot = rg + (size_t)(t & 1) * H3;
        float dh_direct = 0.f;
        if (owner) {
            const float r = pr, z = pz, n = pn, hl = phl, hp = php;
            const float dh = pdy + dh_rec;
            const float dn_pre = dh * (1.f - z) * (1.f - n * n);
            const float dz_pre = dh * (hp - n) * z * (1.f - z);
            const float dr_pre = dn_pre * hl * r * (1.f - r);
            dh_direct = dh * z;
            if (t > 0) {   // hand-off first
                const gru_u64 tg = (gru_u64)tag << 32;
                __hip_atomic_store(slot + j, tg | (gru_u64)__float_as_uint(dr_pre), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(slot + H + j, tg | (gru_u64)__float_as_uint(dz_pre), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(slot + 2 * H + j, tg | (gru_u64)__float_as_uint(dn_pre * r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            float* gi = gib + (size_t)t * H3 + j;
            gi[0] = dr_pre;
            gi[H] = dz_pre;
            gi[2 * H] = dn_pre;
            float* gh = ghb + (size_t)t * H3 + j;
            gh[0] = dr_pre;
            gh[H] = dz_pre;
            gh[2 * H] = dn_pre * r;
        }
        if (t == 0) break;   // dh_{-1} is not needed
        if (owner) fetch(t - 1);   // in flight while the exchange completes
        bool fail = false;
        for (int i0 = tid; i0 < H3; i0 += 4 * 512) {   // (H = 512: all three granules of a thread in one round trip)
            const int n = (H3 - i0 + 511) / 512;
            fail = !gru_poll_n<4>(slot, i0, 512, n < 4 ? n : 4, tag, s.abort_word, dg) || fail;
        }
        if (__syncthreads_or(fail)) return;
        float x = 0.f;
        {
            const float4* d4 = reinterpret_cast<const float4*>(dg + ks * KL);
#pragma unroll
            for (int kb = 0; kb < KL / 4; ++kb) {
                const float4 dv = d4[kb];
                x = fmaf(w[4 * kb], dv.x, x);
                x = fmaf(w[4 * kb + 1], dv.y, x);
                x = fmaf(w[4 * kb + 2], dv.z, x);
                x = fmaf(w[4 * kb + 3], dv.w, x);
            }
        }
        part[ks * HU + u] = x;
        __syncthreads();
        if (owner) {
            float v = dh_direct;
            for (int q = 0; q < KS; ++q) v += part[q * HU + u];
            dh_rec = v;
        }
    }
}

// weight_hh [3H][H] (device) -> forward pack [H/4][3H][4] (transpose = 0) or transposed pack [3H/4][H][4] (transpose = 1)
__global__ void gru_pack_kernel(const float* __restrict__ w, float* __restrict__ out, int H, int transpose) {
    const long total = (long)3 * H * H;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int q = (int)(i & 3);
        const long t = i >> 2;
        float v;
        if (!transpose) {
            const int r = (int)(t % (3 * H)), k4 = (int)(t / (3 * H));
            v = w[(size_t)r * H + 4 * k4 + q];
        } else {
            const int k = (int)(t % H), r4 = (int)(t / H);
            v = w[(size_t)(4 * r4 + q) * H + k];
        }
        out[i] = v;
    }
}

}  // namespace ttsc

using namespace ttsc;

static int gru_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s launch failed: %s", what, hipGetErrorString(e));
        return TTSC_EHIP;
    }
    return TTSC_OK;
}


// members per utterance for the split kernels: power of two, units per member >= 32, all workgroups co-resident
static int gru_split_members(int B, int H) {
    const int cus = device_cus();   // of the CURRENT device
    int gmax = 4;   // measured on MI355X (H=512, B=16, T=24000): G=2.. see DESIGN.md; 4 is the best trade between row stream and hand-off cost
    if (const char* ev = getenv("TTSC_GRU_SPLIT")) gmax = atoi(ev);
    int G = 1;
    while (G * 2 <= gmax && (long)G * 2 * B <= cus && H % (G * 2) == 0 && H / (G * 2) >= 32 && 512 % (H / (G * 2)) == 0) {
        const int HU = H / (G * 2), KS = 512 / HU;
        if (H % KS != 0 || (H / KS) % 8 != 0 || (3 * H / KS) % 16 != 0) break;   // chain batches: K slice multiple of 4*UN
        G *= 2;
    }
    return G;
}

// Counters / abort words of the split recurrences: one area per (device, stream) — common.hpp HandoffArea
static HandoffArea* gru_area(int B, hipStream_t s) {
    if (B > 4096) return nullptr;
    HandoffArea* ar = handoff_area("gru", s, 4096, 0);
    if (!ar || ar->rearm(s) != hipSuccess) return nullptr;   // counters and this launch's abort word restart at zero
    return ar;
}

// members per utterance of the register-resident kernels (0 = not applicable): every thread holds 3 x 32 forward / 96 backward weights, i.e.
// H = 512 over 16 members or H = 256 over 4; all members of all utterances must be co-resident, one workgroup per CU
static int gru_resident_members(int B, int H) {
    static const bool on = !(getenv("TTSC_GRU_RESIDENT") && atoi(getenv("TTSC_GRU_RESIDENT")) == 0);
    if (!on) return 0;
    const int G = H == 512 ? 16 : (H == 256 ? 4 : 0);
    if (!G || (long)G * B > device_cus() || B > 4096) return 0;
    return G;
}
static HandoffArea* gru_ring_area(hipStream_t s, size_t ring_bytes) {
    HandoffArea* ar = handoff_area("gru", s, 4096, ring_bytes);
    if (!ar || ar->rearm(s) != hipSuccess) return nullptr;
    return ar;
}

// 0 = every hand-off of the split GRU launches on this device since the last call completed; 1 = a bounded spin timed out (that
// launch's results are invalid).  Synchronises the device; meant for tests and debugging.
extern "C" int32_t ttsc_gru_split_status(void) { return handoff_status("gru"); }

extern "C" int ttsc_gru_pack_whh_device(const float* whh_dev, int32_t H, int32_t transpose, float* out_dev, void* stream) {
    TTSC_REQUIRE(whh_dev && out_dev, "ttsc_gru_pack_whh_device: null argument");
    TTSC_REQUIRE(H >= 4 && H <= 512 && H % 4 == 0, "ttsc_gru_pack_whh_device: need H %% 4 == 0, H <= 512 (got %d)", H);
    const long total = (long)3 * H * H;
    hipLaunchKernelGGL(gru_pack_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 2048)), dim3(256), 0, (hipStream_t)stream, whh_dev, out_dev, H,
                       transpose);
    return gru_check_launch("gru_pack_kernel");
}

extern "C" int ttsc_gru_seq_forward(const float* xg_dev, const float* whh_packed_dev, const float* bhh_dev, float* y_dev, float* saved_dev,
                                    const float* h0_dev, int32_t B, int32_t T, int32_t H, void* stream) {
    TTSC_REQUIRE(xg_dev && whh_packed_dev && bhh_dev && y_dev, "ttsc_gru_seq_forward: null argument");
    TTSC_REQUIRE(B > 0 && T > 0 && H >= 4 && H <= 512 && H % 4 == 0, "ttsc_gru_seq_forward: bad shape B=%d T=%d H=%d", B, T, H);
    GruArgs a{xg_dev, whh_packed_dev, bhh_dev, y_dev, saved_dev, h0_dev, B, T, H};
    if (const int Gr = gru_resident_members(B, H)) {   // weights in registers, granule hand-off (H = 512: 16 members, H = 256: 4)
        HandoffArea* ar = gru_ring_area((hipStream_t)stream, (size_t)B * 2 * H * sizeof(gru_u64));
        TTSC_REQUIRE(ar, "ttsc_gru_seq_forward: cannot allocate the hand-off ring");
        TTSC_HIP_CHECK(hipMemsetAsync(ar->buf, 0, (size_t)B * 2 * H * sizeof(gru_u64), (hipStream_t)stream));
        GruSplitArgs sa{};
        sa.f = a;
        sa.cnt = ar->words;
        sa.abort_word = ar->abort_word();
        sa.G = Gr;
        sa.HU = H / Gr;
        sa.KS = 512 / sa.HU;
        const size_t lds = ((size_t)H + (size_t)sa.KS * 3 * sa.HU + (size_t)3 * sa.HU) * sizeof(float);
        hipLaunchKernelGGL(gru_seq_split_res_kernel<32>, dim3((unsigned)Gr, (unsigned)B), dim3(512), lds, (hipStream_t)stream, sa, reinterpret_cast<gru_u64*>(ar->buf));
        return gru_check_launch("gru_seq_split_res_kernel");
    }
    const int G = gru_split_members(B, H);
    if (G > 1) {
        HandoffArea* ar = gru_area(B, (hipStream_t)stream);
        TTSC_REQUIRE(ar, "ttsc_gru_seq_forward: cannot allocate the hand-off counters");
        GruSplitArgs sa{};
        sa.f = a;
        sa.cnt = ar->words;
        sa.abort_word = ar->abort_word();
        sa.G = G;
        sa.HU = H / G;
        sa.KS = 512 / sa.HU;
        const size_t lds = ((size_t)H + (size_t)sa.KS * 3 * sa.HU) * sizeof(float);
        hipLaunchKernelGGL(gru_seq_split_kernel, dim3((unsigned)G, (unsigned)B), dim3(512), lds, (hipStream_t)stream, sa);
        return gru_check_launch("gru_seq_split_kernel");
    }
    hipLaunchKernelGGL(gru_seq_kernel, dim3((unsigned)B), dim3((unsigned)round_up(H, 64)), (size_t)2 * H * sizeof(float), (hipStream_t)stream, a);
    return gru_check_launch("gru_seq_kernel");
}

extern "C" int ttsc_gru_seq_backward(const float* dy_dev, const float* saved_dev, const float* y_dev, const float* h0_dev,
                                     const float* whhT_packed_dev, float* dgi_dev, float* dgh_dev, int32_t B, int32_t T, int32_t H, void* stream) {
    TTSC_REQUIRE(dy_dev && saved_dev && y_dev && whhT_packed_dev && dgi_dev && dgh_dev, "ttsc_gru_seq_backward: null argument");
    TTSC_REQUIRE(B > 0 && T > 0 && H >= 4 && H <= 512 && H % 4 == 0, "ttsc_gru_seq_backward: bad shape B=%d T=%d H=%d", B, T, H);
    GruBwdArgs a{dy_dev, saved_dev, y_dev, h0_dev, whhT_packed_dev, dgi_dev, dgh_dev, B, T, H};
    if (const int Gr = gru_resident_members(B, H)) {
        HandoffArea* ar = gru_ring_area((hipStream_t)stream, (size_t)B * 2 * 3 * H * sizeof(gru_u64));
        TTSC_REQUIRE(ar, "ttsc_gru_seq_backward: cannot allocate the hand-off ring");
        TTSC_HIP_CHECK(hipMemsetAsync(ar->buf, 0, (size_t)B * 2 * 3 * H * sizeof(gru_u64), (hipStream_t)stream));
        GruSplitArgs sa{};
        sa.bw = a;
        sa.cnt = ar->words;
        sa.abort_word = ar->abort_word();
        sa.G = Gr;
        sa.HU = H / Gr;
        sa.KS = 512 / sa.HU;
        const size_t lds = ((size_t)3 * H + (size_t)sa.KS * sa.HU) * sizeof(float);
        hipLaunchKernelGGL(gru_bwd_split_res_kernel<96>, dim3((unsigned)Gr, (unsigned)B), dim3(512), lds, (hipStream_t)stream, sa, reinterpret_cast<gru_u64*>(ar->buf));
        return gru_check_launch("gru_bwd_split_res_kernel");
    }
    const int G = gru_split_members(B, H);
    if (G > 1) {
        HandoffArea* ar = gru_area(B, (hipStream_t)stream);
        TTSC_REQUIRE(ar, "ttsc_gru_seq_backward: cannot allocate the hand-off counters");
        GruSplitArgs sa{};
        sa.bw = a;
        sa.cnt = ar->words;
        sa.abort_word = ar->abort_word();
        sa.G = G;
        sa.HU = H / G;
        sa.KS = 512 / sa.HU;
        const size_t lds = ((size_t)3 * H + (size_t)sa.KS * sa.HU) * sizeof(float);
        hipLaunchKernelGGL(gru_bwd_split_kernel, dim3((unsigned)G, (unsigned)B), dim3(512), lds, (hipStream_t)stream, sa);
        return gru_check_launch("gru_bwd_split_kernel");
    }
    hipLaunchKernelGGL(gru_bwd_kernel, dim3((unsigned)B), dim3((unsigned)round_up(H, 64)), (size_t)3 * H * sizeof(float), (hipStream_t)stream, a);
    return gru_check_launch("gru_bwd_kernel");
}
