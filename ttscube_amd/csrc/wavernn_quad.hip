// WaveRNN decode with every tile of 4 utterances spread over 4 workgroups ("quad") — gfx950.
//
// The single-workgroup kernel (wavernn.hip) streams the whole fp32 weight set (3.8 MB for H=512) from L2 every step and is
// bound by ONE CU's load path (~64 B/clk): 45 us per step at B = 256.  Here a quad of NC = 4 workgroups steps BU = 4
// utterances together; member m owns a quarter of the rows of every matrix (H/4 hidden units x 3 gates, 64 rows of the
// pre-output layer, S/4 rows of the output layer) and streams ONLY those rows — each 16-byte weight word is used for
// four utterances, so a member moves a quarter of the bytes per step while B = 256 still fills all 256 CUs (64 quads).
// Per step the members exchange four small vectors (h_t, pre, logits, last_x): write-through (agent-scope) payload stores,
// one monotonic arrival counter per edge, bounded spins + abort word; with 4 members a hand-off costs ~1 us (measured on the
// GRU training kernels, gru.hip).  (A 32-member weight-stationary cluster variant was built in round 1 and measured slower
// than both this kernel and the streaming one — ~8 us per all-to-all hand-off — and was removed.)
//
// Arithmetic is IDENTICAL to wavernn.hip / oracle/wavernn_ref.c: rows are split across members, never the reduction —
// every (row, utterance) is one k-ordered fmaf chain seeded with the bias — so indices and logits stay bit-exact.
#include "rnn_chain.hpp"

namespace ttsc {

constexpr int WQ_NC = 4;        // members per quad
constexpr int WQ_BU = 4;        // utterances per quad
constexpr int WQ_THREADS = 512;

constexpr unsigned WC_SPIN_LIMIT = 1u << 22;   // bounded spins: a member that is not resident must not hang the GPU

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

struct WqArgs {
    const float* mel;      // [B, T, n_mel]
    const float* interp;   // [B, Tl*up_low]
    const float* feats;    // [B, 20, Tl]
    // per-member row slices, packed [K/4][rows][4]; member m at offset m * (K * rows)
    const float* whh;      // rows = 3*UPW (gate q, local unit j -> q*UPW + j), K = H
    const float* wih;      // rows = 3*UPW, K = I0P (in_dim rounded up to 4, zero padded)
    const float* bih;      // [NC][3*UPW]
    const float* bhh;      // [NC][3*UPW]
    const float* wpre;     // rows = PR = 256/NC, K = H
    const float* bpre;     // [NC][PR]
    const float* wout;     // rows = SR = S/NC, K = 256
    const float* bout;     // [NC][SR]
    const float* lut;
    const float* noise;    // [B, L, S] or null
    const float* forced_x; // [B, L] or null
    uint8_t* out_idx;
    float* out_wav;
    float* out_logits;
    // exchange area (device memory), per quad g
    float* xh;             // [G][2][BU][H]
    float* xpre;           // [G][2][BU][256]
    float* xlog;           // [G][2][BU][S]
    float* xlx;            // [G][2][BU]
    unsigned* cnt;         // [G][4] arrival counters (h, pre, logits, last_x); word [G*4] = abort
    int B, T, Tl, H, UPW, I0, I0P, use_lowres, up, up_low, S, SR, PR, n_mel, out_kind, mode, L, G;
    unsigned long long seed;
};

__global__ __launch_bounds__(WQ_THREADS) void wr_quad_kernel(WqArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];   // vec[BU][max(H,256)] | scr[S]
    const int H = a.H, UPW = a.UPW, S = a.S, SR = a.SR, PR = a.PR, NM = a.n_mel, I0P = a.I0P;
    const int R3 = 3 * UPW;
    const int g = blockIdx.x / WQ_NC, m = blockIdx.x % WQ_NC;
    const int VW = H > 256 ? H : 256;
    float* vec = sm;
    float* scr = sm + (size_t)WQ_BU * VW;
    const int tid = threadIdx.x;
    const int u = tid & (WQ_BU - 1);   // utterance slot
    const int j = tid >> 2;            // local hidden unit (GRU) / local row (pre, out)
    const int bu = g * WQ_BU + u;
    const bool uok = bu < a.B;
    const int bc = uok ? bu : a.B - 1;
    const int nu = min(WQ_BU, a.B - g * WQ_BU);
    const bool gru_thr = j < UPW;
    const int jc = gru_thr ? j : 0;
    const float* Whh = a.whh + (size_t)m * H * R3;
    const float* Wih = a.wih + (size_t)m * I0P * R3;
    const float* Wpre = a.wpre + (size_t)m * H * PR;
    const float* Wout = a.wout + (size_t)m * 256 * SR;
    float bih[3], bhh[3], w_int[3], w_lx[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        bih[q] = a.bih[(size_t)m * R3 + q * UPW + jc];
        bhh[q] = a.bhh[(size_t)m * R3 + q * UPW + jc];
        const int k1 = a.I0 - 1, k2 = a.I0 >= 2 ? a.I0 - 2 : 0;
        w_lx[q] = Wih[((size_t)(k1 >> 2) * R3 + q * UPW + jc) * 4 + (k1 & 3)];
        w_int[q] = Wih[((size_t)(k2 >> 2) * R3 + q * UPW + jc) * 4 + (k2 & 3)];
    }
    const float bpre = a.bpre[m * PR + (j < PR ? j : 0)];
    const float bout = a.bout[m * SR + (j < SR ? j : 0)];
    float pmel[3] = {0, 0, 0}, plow[3] = {0, 0, 0};
    float hprev = 0.f;   // h_{t-1}[unit m*UPW + j][utterance u]: each (unit, utterance) has exactly one owner thread
    unsigned* cnt = a.cnt + (size_t)g * 4;
    unsigned* abort_word = a.cnt + (size_t)a.G * 4;
    float* xh = a.xh + (size_t)g * 2 * WQ_BU * H;
    float* xpre = a.xpre + (size_t)g * 2 * WQ_BU * 256;
    float* xlog = a.xlog + (size_t)g * 2 * WQ_BU * S;
    float* xlx = a.xlx + (size_t)g * 2 * WQ_BU;

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
                    for (int q = 0; q < 3; ++q) pmel[q] = fmaf(Wih[((size_t)(k >> 2) * R3 + q * UPW + j) * 4 + (k & 3)], v, pmel[q]);
                }
            }
            if (a.use_lowres && lo_phase == 0) {
#pragma unroll
                for (int q = 0; q < 3; ++q) plow[q] = pmel[q];
                for (int f = 0; f < 20; ++f) {
                    const int k = NM + f;
                    const float v = a.feats[((size_t)bc * 20 + f) * a.Tl + lo];
#pragma unroll
                    for (int q = 0; q < 3; ++q) plow[q] = fmaf(Wih[((size_t)(k >> 2) * R3 + q * UPW + j) * 4 + (k & 3)], v, plow[q]);
                }
            }
        }
        // ---- phase A: GRU slice over h_{t-1} of every unit ----
        if (t > 0) {
            if (!wait_count(cnt + 0, (unsigned)t * WQ_NC, abort_word)) return;
            if (!wait_count(cnt + 3, (unsigned)t * (unsigned)nu, abort_word)) return;
            const float* src = xh + (size_t)(par ^ 1) * WQ_BU * H;
            for (int i = tid; i < WQ_BU * H; i += WQ_THREADS) vec[(i / H) * VW + (i % H)] = ld_f32(src + i);
        } else {
            for (int i = tid; i < WQ_BU * VW; i += WQ_THREADS) vec[i] = 0.f;   // h_{-1} = 0: fmaf(w, 0, acc) == acc
        }
        __syncthreads();
        if (gru_thr) {
            float gh[1][3] = {{bhh[0], bhh[1], bhh[2]}};
            lstm_chain<1, 3, 2>(gh, Whh, R3, UPW, j, vec + u * VW, VW, H);
            const float lx = (t > 0) ? ld_f32(xlx + (par ^ 1) * WQ_BU + u) : 0.f;
            float gi[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                float acc = a.use_lowres ? plow[q] : pmel[q];
                if (a.use_lowres) acc = fmaf(w_int[q], a.interp[(size_t)bc * ((size_t)a.Tl * a.up_low) + t], acc);
                gi[q] = fmaf(w_lx[q], lx, acc);
            }
            const float r = ttsc_sigmoidf(gi[0] + gh[0][0]);
            const float z = ttsc_sigmoidf(gi[1] + gh[0][1]);
            const float rg = r * gh[0][2];
            const float nn = ttsc_tanhf(gi[2] + rg);
            const float d = hprev - nn;
            hprev = fmaf(z, d, nn);
            st_f32(xh + ((size_t)par * WQ_BU + u) * H + m * UPW + j, hprev);
        }
        publish(cnt + 0);
        // ---- phase B: pre-output slice (PR rows) over the full h_t ----
        if (!wait_count(cnt + 0, (unsigned)(t + 1) * WQ_NC, abort_word)) return;
        {
            const float* src = xh + (size_t)par * WQ_BU * H;
            for (int i = tid; i < WQ_BU * H; i += WQ_THREADS) vec[(i / H) * VW + (i % H)] = ld_f32(src + i);
        }
        __syncthreads();
        if (j < PR) {
            float acc[1][1] = {{bpre}};
            lstm_chain<1, 1, 4>(acc, Wpre, PR, 0, j, vec + u * VW, VW, H);
            st_f32(xpre + ((size_t)par * WQ_BU + u) * 256 + m * PR + j, ttsc_tanhf(acc[0][0]));
        }
        publish(cnt + 1);
        // ---- phase C: output slice (SR rows) over the full pre-output ----
        if (!wait_count(cnt + 1, (unsigned)(t + 1) * WQ_NC, abort_word)) return;
        {
            const float* src = xpre + (size_t)par * WQ_BU * 256;
            for (int i = tid; i < WQ_BU * 256; i += WQ_THREADS) vec[(i >> 8) * VW + (i & 255)] = ld_f32(src + i);
        }
        __syncthreads();
        if (j < SR) {
            float acc[1][1] = {{bout}};
            lstm_chain<1, 1, 4>(acc, Wout, SR, 0, j, vec + u * VW, VW, 256);
            const int s_ = m * SR + j;
            st_f32(xlog + ((size_t)par * WQ_BU + u) * S + s_, acc[0][0]);
            if (a.out_logits && uok) a.out_logits[((size_t)bu * a.L + t) * S + s_] = acc[0][0];
        }
        publish(cnt + 2);
        // ---- phase D: member m samples utterance m of the quad ----
        if (m < nu) {
            if (!wait_count(cnt + 2, (unsigned)(t + 1) * WQ_NC, abort_word)) return;
            const int bs = g * WQ_BU + m;
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
                scr[tid] = ld_f32(xlog + ((size_t)par * WQ_BU + m) * S + tid) + g_;
            }
            __syncthreads();
            if (tid < 64) {
                float bs_ = scr[tid < S ? tid : 0];
                int bi = tid < S ? tid : 0;
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
                    st_f32(xlx + par * WQ_BU + m, a.forced_x ? a.forced_x[o] : wv);
                }
            }
            publish(cnt + 3);
        }
        if (++fr_phase == a.up) { fr_phase = 0; ++fr; }
        if (++lo_phase == a.up_low) { lo_phase = 0; ++lo; }
    }
}

}  // namespace ttsc
