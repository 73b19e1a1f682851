// WaveRNN decode, WEIGHT-STATIONARY across a cluster of 32 workgroups (one per CU, ideally one XCD) — gfx950.
//
// The single-workgroup kernel (wavernn.hip) streams the whole fp32 weight set (3.8 MB for H=512) from L2 every step and is
// bound by the per-CU load path (~64 B/clk): ~40 us per step however small the batch.  Here a cluster of NC = 32
// workgroups keeps ALL weights resident in LDS (each member holds 1/32 of every matrix: 16 hidden units x 3 gates of
// W_hh / W_ih, 8 rows of the pre-output layer, 8 rows of the output layer = 140 KB for H=512) and steps a tile of up to 32
// utterances together.  Per step the members exchange four small vectors through L2 (never HBM):
//     h_t [H x 32] -> every member          (after the GRU slice)
//     pre [256 x 32] -> every member        (after the pre-output slice)
//     logits [32 x 256] -> the sampling member of each utterance
//     last_x [32] -> every member           (after Gumbel-max + mu-law decode)
// Hand-off protocol (MI355X_MICROARCH.md "Inter-workgroup visibility", recipe R1 with write-through payload): payload is
// stored AND loaded with agent-scope relaxed atomics (sc1: bypass the non-coherent per-CU L1 / write through the XCD L2),
// every storing wave drains vmcnt(0), one lane bumps a monotonic arrival counter, consumers poll that counter from one
// lane.  Placement-independent (works for any block->XCD mapping; same-XCD is merely faster).  Every spin is bounded: on
// timeout a global abort word is set, every member leaves, and the host reports TTSC_ESTATE instead of hanging the GPU.
//
// Arithmetic is IDENTICAL to wavernn.hip / oracle/wavernn_ref.c (k-ordered fmaf chains seeded with the bias, cached input
// prefixes, ttscube_math.h transcendentals), so indices and logits stay bit-exact; only the work distribution changes.
#include "common.hpp"
#include "../../include/ttscube_math.h"

namespace ttsc {

constexpr int WC_NC = 32;       // members per cluster
constexpr int WC_BU = 32;       // utterances per cluster
constexpr int WC_THREADS = 512;
constexpr unsigned WC_SPIN_LIMIT = 1u << 22;

struct WcArgs {
    const float* mel;      // [B, T, n_mel]
    const float* interp;   // [B, Tl*up_low]
    const float* feats;    // [B, 20, Tl]
    // per-member weight slices, packed on the host (see pack functions): member m at offset m * stride
    const float* whh;      // [NC][H/4][3*UPW][4]
    const float* wih;      // [NC][I0P/4][3*UPW][4]   (I0P = in_dim rounded up to 4, zero padded)
    const float* bih;      // [NC][3*UPW]
    const float* bhh;      // [NC][3*UPW]
    const float* wpre;     // [NC][H/4][8][4]
    const float* bpre;     // [NC][8]
    const float* wout;     // [NC][256/4][SR][4]   (SR = rows of the output layer per member = S/NC)
    const float* bout;     // [NC][SR]
    const float* lut;
    const float* noise;    // [B, L, S] or null
    const float* forced_x; // [B, L] or null
    uint8_t* out_idx;
    float* out_wav;
    float* out_logits;
    // exchange area (device memory, zeroed before every launch)
    float* xh;             // [G][2][H/2][BU][2]   h_t, two consecutive k per 8-byte item
    float* xpre;           // [G][2][256/2][BU][2]
    float* xlog;           // [G][2][BU][S]
    float* xlx;            // [G][2][BU]
    unsigned* cnt;         // [G][4] arrival counters (h, pre, logits, last_x) + [G*4] = abort word
    int B, T, Tl, H, UPW, I0, I0P, use_lowres, up, up_low, S, SR, n_mel, out_kind, mode, L, G;
    unsigned long long seed;
};

typedef unsigned long long u64;

__device__ __forceinline__ void st_f32(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float2 ld_f32x2(const float* p) {
    const u64 x = __hip_atomic_load(reinterpret_cast<const u64*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float2 r;
    r.x = __uint_as_float((unsigned)x);
    r.y = __uint_as_float((unsigned)(x >> 32));
    return r;
}
__device__ __forceinline__ float ld_f32(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// one lane polls a monotonic counter; bounded; returns false after a timeout / when another member aborted
__device__ __forceinline__ bool wait_count(unsigned* cnt, unsigned want, unsigned* abort_word) {
    __shared__ int ok_s;
    if (threadIdx.x == 0) {
        int ok = 1;
        unsigned spins = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
            if (++spins > WC_SPIN_LIMIT || __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        ok_s = ok;
    }
    __syncthreads();
    const bool r = ok_s != 0;
    __syncthreads();
    return r;
}

// every storing wave drains its write-through stores, then ONE lane bumps the arrival counter
__device__ __forceinline__ void publish(unsigned* cnt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// acc[r] <- k-ordered fmaf chain over K inputs of an EXCHANGED vector (global, layout [K/2][BU][2], written by other
// workgroups with write-through stores) for NR weight rows held in LDS (row r at wr[r] + (k>>2)*kstride + (k&3)).
// Every (k-pair, utterance) item is fetched ONCE per workgroup: all 512 threads issue 8-byte sc1 loads (4 chunks = 8 loads
// per thread in flight), park them in a double-buffered LDS staging tile of 32 k-pairs x 32 utterances, and the owner
// threads (`active`) read their utterance's pairs back as conflict-free 8-byte LDS reads.  One barrier per chunk.
template <int NR>
__device__ __forceinline__ void wc_stage_chain(float (&acc)[NR], const float* const (&wr)[NR], int kstride, bool active, int u,
                                               const float* __restrict__ src, int K, float* stage) {
    constexpr int CK2 = 32, PD = 4;                    // k-pairs per chunk, prefetch depth (chunks)
    const int K2 = K >> 1;
    const int nch = (K2 + CK2 - 1) / CK2;
    const int tid = threadIdx.x;
    // this thread's two items of a chunk: item i = tid + e*512 -> (k2 = i / 32, utterance = i % 32); contiguous in memory
    u64 pf[PD][2];
    auto issue = [&](int c, u64 (&dst)[2]) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            int k2 = c * CK2 + ((tid + e * WC_THREADS) >> 5);
            k2 = k2 < K2 ? k2 : K2 - 1;
            dst[e] = __hip_atomic_load(reinterpret_cast<const u64*>(src) + (size_t)k2 * WC_BU + (tid & 31), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    u64* st64 = reinterpret_cast<u64*>(stage);         // [2][CK2*BU]
#pragma unroll
    for (int d = 0; d < PD; ++d)
        if (d < nch) issue(d, pf[d]);
    for (int c0 = 0; c0 < nch; c0 += PD) {
#pragma unroll
        for (int d = 0; d < PD; ++d) {
            const int c = c0 + d;
            if (c < nch) {
                u64* sb = st64 + (size_t)(c & 1) * (CK2 * WC_BU);
                sb[tid] = pf[d][0];
                sb[tid + WC_THREADS] = pf[d][1];
                if (c + PD < nch) issue(c + PD, pf[d]);
                __syncthreads();
                if (active) {
                    const int kn = min(CK2, K2 - c * CK2);
                    const float2* sv = reinterpret_cast<const float2*>(sb) + u;
#pragma unroll 8
                    for (int q = 0; q < kn; ++q) {
                        const float2 hv = sv[q * WC_BU];
                        const int k = 2 * (c * CK2 + q);
#pragma unroll
                        for (int r = 0; r < NR; ++r) {
                            const float* w = wr[r] + (k >> 2) * kstride + (k & 3);
                            acc[r] = fmaf(w[0], hv.x, acc[r]);
                            acc[r] = fmaf(w[1], hv.y, acc[r]);
                        }
                    }
                }
            }
        }
    }
    __syncthreads();   // the staging tile may be rewritten by the next caller
}

__global__ __launch_bounds__(WC_THREADS) void wr_cluster_kernel(WcArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int H = a.H, UPW = a.UPW, S = a.S, SR = a.SR, NM = a.n_mel, I0P = a.I0P;
    const int R3 = 3 * UPW;
    // cluster / member ids: with the observed round-robin block->XCD placement, `bid % G` keeps a cluster on one XCD
    const int g = blockIdx.x % a.G;
    const int m = blockIdx.x / a.G;
    float* Whh = sm;                          // [H/4][R3][4]
    float* Wih = Whh + (size_t)H * R3;        // [I0P/4][R3][4]
    float* Wpre = Wih + (size_t)I0P * R3;     // [H/4][8][4]
    float* Wout = Wpre + (size_t)H * 8;       // [64][SR][4]
    float* scr = Wout + (size_t)256 * SR;     // [S] sampling scratch
    float* stage = scr + S;                   // [2][32 k-pairs][32 utterances][2] staging tile of exchanged vectors (16 KB)
    const int tid = threadIdx.x;
    // ---- load this member's weight slices into LDS (once) ----
    {
        const float* s0 = a.whh + (size_t)m * H * R3;
        for (int i = tid; i < H * R3; i += WC_THREADS) Whh[i] = s0[i];
        const float* s1 = a.wih + (size_t)m * I0P * R3;
        for (int i = tid; i < I0P * R3; i += WC_THREADS) Wih[i] = s1[i];
        const float* s2 = a.wpre + (size_t)m * H * 8;
        for (int i = tid; i < H * 8; i += WC_THREADS) Wpre[i] = s2[i];
        const float* s3 = a.wout + (size_t)m * 256 * SR;
        for (int i = tid; i < 256 * SR; i += WC_THREADS) Wout[i] = s3[i];
    }
    const int u = tid & 31;          // utterance slot
    const int j = tid >> 5;          // local hidden unit (GRU) / local row (pre, out)
    const int bu = g * WC_BU + u;    // utterance index
    const bool uok = bu < a.B;
    const int bc = uok ? bu : a.B - 1;
    const int nu = min(WC_BU, a.B - g * WC_BU);   // utterances of this cluster
    const bool gru_thr = j < UPW;
    float bih[3] = {0, 0, 0}, bhh[3] = {0, 0, 0};
    if (gru_thr) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            bih[q] = a.bih[(size_t)m * R3 + q * UPW + j];
            bhh[q] = a.bhh[(size_t)m * R3 + q * UPW + j];
        }
    }
    const float bpre = (j < 8) ? a.bpre[m * 8 + j] : 0.f;
    const float bout = (j < SR) ? a.bout[m * SR + j] : 0.f;
    float pmel[3] = {0, 0, 0}, plow[3] = {0, 0, 0};
    float hprev = 0.f;   // h_{t-1}[unit 'm*UPW + j'][utterance u] stays in a register (each (unit, utterance) has one owner)
    unsigned* cnt = a.cnt + (size_t)g * 4;
    unsigned* abort_word = a.cnt + (size_t)a.G * 4;
    float* xh = a.xh + (size_t)g * 2 * H * WC_BU;
    float* xpre = a.xpre + (size_t)g * 2 * 256 * WC_BU;
    float* xlog = a.xlog + (size_t)g * 2 * WC_BU * S;
    float* xlx = a.xlx + (size_t)g * 2 * WC_BU;
    __syncthreads();

    int fr = 0, fr_phase = 0, lo = 0, lo_phase = 0;
    for (int t = 0; t < a.L; ++t) {
        const int par = t & 1;
        // ---- cached prefixes of the layer-0 input chain (same order as wavernn.hip: mel | low-res feats | interp | last_x) ----
        if (gru_thr) {
            if (fr_phase == 0) {
#pragma unroll
                for (int q = 0; q < 3; ++q) pmel[q] = bih[q];
                const float* mf = a.mel + ((size_t)bc * a.T + fr) * NM;
                for (int k = 0; k < NM; ++k) {
                    const float v = mf[k];
#pragma unroll
                    for (int q = 0; q < 3; ++q) pmel[q] = fmaf(Wih[((k >> 2) * R3 + q * UPW + j) * 4 + (k & 3)], v, pmel[q]);
                }
            }
            if (a.use_lowres && lo_phase == 0) {
#pragma unroll
                for (int q = 0; q < 3; ++q) plow[q] = pmel[q];
                for (int f = 0; f < 20; ++f) {
                    const int k = NM + f;
                    const float v = a.feats[((size_t)bc * 20 + f) * a.Tl + lo];
#pragma unroll
                    for (int q = 0; q < 3; ++q) plow[q] = fmaf(Wih[((k >> 2) * R3 + q * UPW + j) * 4 + (k & 3)], v, plow[q]);
                }
            }
        }
        // ---- phase A: GRU slice.  needs h_{t-1} of every unit (exchange) and last_x_{t-1} ----
        if (t > 0) {
            if (!wait_count(cnt + 0, (unsigned)t * WC_NC, abort_word)) return;
            if (!wait_count(cnt + 3, (unsigned)t * (unsigned)nu, abort_word)) return;
        }
        float gh[3] = {bhh[0], bhh[1], bhh[2]};
        if (t > 0) {   // h_{-1} = 0: fmaf(w, 0, acc) == acc, the chain over zeros is skipped at t = 0
            const float* const wr[3] = {Whh + (0 * UPW + (gru_thr ? j : 0)) * 4, Whh + (1 * UPW + (gru_thr ? j : 0)) * 4,
                                        Whh + (2 * UPW + (gru_thr ? j : 0)) * 4};
            wc_stage_chain<3>(gh, wr, R3 * 4, gru_thr, u, xh + (size_t)(par ^ 1) * H * WC_BU, H, stage);
        }
        if (gru_thr) {
            float gi[3];
            const float lx = (t > 0) ? ld_f32(xlx + (par ^ 1) * WC_BU + u) : 0.f;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                float acc = a.use_lowres ? plow[q] : pmel[q];
                if (a.use_lowres) {
                    const int k = a.I0 - 2;
                    acc = fmaf(Wih[((k >> 2) * R3 + q * UPW + j) * 4 + (k & 3)], a.interp[(size_t)bc * ((size_t)a.Tl * a.up_low) + t], acc);
                }
                const int k = a.I0 - 1;
                gi[q] = fmaf(Wih[((k >> 2) * R3 + q * UPW + j) * 4 + (k & 3)], lx, acc);
            }
            const float r = ttsc_sigmoidf(gi[0] + gh[0]);
            const float z = ttsc_sigmoidf(gi[1] + gh[1]);
            const float rg = r * gh[2];
            const float nn = ttsc_tanhf(gi[2] + rg);
            const float d = hprev - nn;
            hprev = fmaf(z, d, nn);
            const int unit = m * UPW + j;
            st_f32(xh + (size_t)par * H * WC_BU + ((size_t)(unit >> 1) * WC_BU + u) * 2 + (unit & 1), hprev);
        }
        publish(cnt + 0);
        // ---- phase B: pre-output slice (8 rows) over the full h_t ----
        if (!wait_count(cnt + 0, (unsigned)(t + 1) * WC_NC, abort_word)) return;
        {
            float acc[1] = {bpre};
            const float* const wr[1] = {Wpre + ((j < 8) ? j : 0) * 4};
            wc_stage_chain<1>(acc, wr, 8 * 4, j < 8, u, xh + (size_t)par * H * WC_BU, H, stage);
            if (j < 8) {
                const int row = m * 8 + j;
                st_f32(xpre + (size_t)par * 256 * WC_BU + ((size_t)(row >> 1) * WC_BU + u) * 2 + (row & 1), ttsc_tanhf(acc[0]));
            }
        }
        publish(cnt + 1);
        // ---- phase C: output slice (SR rows) over the full pre-output ----
        if (!wait_count(cnt + 1, (unsigned)(t + 1) * WC_NC, abort_word)) return;
        {
            float acc[1] = {bout};
            const float* const wr[1] = {Wout + ((j < SR) ? j : 0) * 4};
            wc_stage_chain<1>(acc, wr, SR * 4, j < SR, u, xpre + (size_t)par * 256 * WC_BU, 256, stage);
            if (j < SR) {
                const int s_ = m * SR + j;
                st_f32(xlog + ((size_t)par * WC_BU + u) * S + s_, acc[0]);
                if (a.out_logits && uok) a.out_logits[((size_t)bu * a.L + t) * S + s_] = acc[0];
            }
        }
        publish(cnt + 2);
        // ---- phase D: member m samples utterance m of the cluster ----
        if (m < nu) {
            if (!wait_count(cnt + 2, (unsigned)(t + 1) * WC_NC, abort_word)) return;
            const int bs = g * WC_BU + m;
            if (tid < S) {
                float g_ = 0.f;
                const size_t o = ((size_t)bs * a.L + t) * S + tid;
                if (a.mode == 1) {
                    g_ = a.noise[o];
                } else if (a.mode == 2) {
                    uint32_t r4[4];
                    ttsc_philox4x32((uint32_t)(tid >> 2), (uint32_t)t, (uint32_t)bs, 0u, (uint32_t)a.seed, (uint32_t)(a.seed >> 32), r4);
                    g_ = ttsc_gumbel(r4[tid & 3]);
                }
                scr[tid] = ld_f32(xlog + ((size_t)par * WC_BU + m) * S + tid) + g_;
            }
            __syncthreads();
            if (tid < 64) {
                float bs_ = scr[tid];
                int bi = tid;
                for (int s = tid + 64; s < S; s += 64) {
                    const float v = scr[s];
                    if (v > bs_) {
                        bs_ = v;
                        bi = s;
                    }
                }
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) {
                    const float os = __shfl_xor(bs_, off);
                    const int oi = __shfl_xor(bi, off);
                    if (os > bs_ || (os == bs_ && oi < bi)) {
                        bs_ = os;
                        bi = oi;
                    }
                }
                if (tid == 0) {
                    const float wv = a.out_kind == 0 ? a.lut[bi] : (((float)bi / 255.0f) - 0.5f) * 2.0f;
                    const size_t o = (size_t)bs * a.L + t;
                    a.out_idx[o] = (uint8_t)bi;
                    a.out_wav[o] = wv;
                    st_f32(xlx + par * WC_BU + m, a.forced_x ? a.forced_x[o] : wv);
                }
            }
            publish(cnt + 3);
        }
        if (++fr_phase == a.up) { fr_phase = 0; ++fr; }
        if (++lo_phase == a.up_low) { lo_phase = 0; ++lo; }
    }
}

}  // namespace ttsc
