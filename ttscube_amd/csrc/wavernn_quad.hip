// WaveRNN decode with every tile of NC utterances spread over NC workgroups (a "quad" for NC = 4, an octet for NC = 8) — gfx950.
//
// The single-workgroup kernel (wavernn.hip) streams the whole fp32 weight set (3.8 MB for H=512) from L2 every step and is
// bound by ONE CU's vector-memory path (64 B/clk): 45 us per step at B = 256.  Here NC workgroups step BU = NC utterances
// together; member m owns 1/NC of the rows of every matrix (H/NC hidden units x 3 gates, 256/NC rows of the pre-output
// layer, S/NC rows of the output layer) and streams ONLY those rows, while B = 256 still fills all 256 CUs.
//
// The row x utterance products run on the matrix pipe as v_mfma_f32_4x4x1_16B_f32: one instruction is 16 independent
// 4 (rows) x 4 (utterances) rank-1 updates  acc += w[row][k] * h[k][utt], i.e. one fused multiply-add per accumulator and per
// k — bit for bit the k-ordered fmaf chain of wavernn.hip / oracle/wavernn_ref.c (tools/probes/mfma_f32_exact.hip checks
// the instruction against fmaf, denormals included).  What it buys over the vector ALU: lane l loads the 16-byte weight word
// of ITS row only (no lane duplicates a neighbour's load, which is what bound round 1's version of this kernel: 4 lanes per
// row = 4x the vector-memory traffic), and the h operand is one 16-byte LDS read per 4 k (lane l reads utterance l & 3)
// instead of one per utterance.
//
// Per step the members exchange four small vectors (h_t, pre, logits, last_x): write-through (agent-scope) payload stores,
// one monotonic arrival counter per edge, bounded spins + abort word.  Members of a tile are placed on the same XCD
// (blockIdx -> XCD is round-robin), so a hand-off is an L2 round trip.  (A 32-member weight-stationary cluster variant was
// built in round 1 and measured slower than both this kernel and the streaming one, and was removed.)
//
// Arithmetic is IDENTICAL to wavernn.hip / oracle/wavernn_ref.c: rows are split across members, never the reduction —
// every (row, utterance) is one k-ordered fused-multiply-add chain seeded with the bias — so indices and logits stay bit-exact.
#include "rnn_chain.hpp"

namespace ttsc {

constexpr int WQ_THREADS = 512;
constexpr int WQ_XCDS = 8;

constexpr unsigned WC_SPIN_LIMIT = 1u << 22;   // bounded spins: a member that is not resident must not hang the GPU

typedef unsigned long long u64;
typedef float f32x4_t __attribute__((ext_vector_type(4)));

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
    // exchange area (device memory), per tile g
    float* xh;             // [G][2][H][BU]
    float* xpre;           // [G][2][256][BU]
    float* xlog;           // [G][2][BU][S]
    float* xlx;            // [G][2][BU]
    unsigned* cnt;         // [G][4] arrival counters (h, pre, logits, last_x); word [G*4] = abort
    int B, T, Tl, H, UPW, I0, I0P, use_lowres, up, up_low, S, SR, PR, n_mel, out_kind, mode, L, G, GP;
    unsigned long long seed;
    unsigned long long* prof;   // -DTTSC_ABLATE: [workgroup][16] accumulated 100 MHz ticks per phase segment (thread 0), or null
};

// One 64-row block of  acc[row][utt] = fma-chain_k( W[row][k] * v[utt][k] )  on the 4x4x1 fp32 matrix instruction.
//   lane l streams the packed weight words of row `row` (its own; clamped by the caller when the block is ragged),
//   reads v[(l & 3) + 4*nb][4*kb .. 4*kb+3] from LDS, and ends up with rows 4*(l >> 2) + i (register i), utterance (l & 3) + 4*nb.
// The weight stream is software-pipelined by hand (two register sets of UN 16-byte words), as in rnn_chain.hpp.
template <int NB, int UN>
__device__ __forceinline__ void mfma_rows(f32x4_t (&acc)[NB], const float* __restrict__ wp, int rows, int row, const float* v, int vstride,
                                          int K, int lane) {
    const float4* w4 = reinterpret_cast<const float4*>(wp) + row;
    const float* hb = v + (lane & 3) * vstride;
    const int KB = K >> 2;
    // one batch = UN k-blocks: the weight words (global) AND the h words (LDS) of batch i+1 are issued before the chain of
    // batch i runs, so neither latency sits between two dependent matrix instructions
    auto load = [&](float4 (&w)[UN], float4 (&hv)[UN][NB], int kb0) {
#pragma unroll
        for (int q = 0; q < UN; ++q) w[q] = w4[(size_t)(kb0 + q) * rows];
#pragma unroll
        for (int q = 0; q < UN; ++q)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) hv[q][nb] = *reinterpret_cast<const float4*>(hb + 4 * nb * vstride + 4 * (kb0 + q));
    };
    auto fma_batch = [&](const float4 (&w)[UN], const float4 (&hv)[UN][NB]) {
#pragma unroll
        for (int q = 0; q < UN; ++q) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[q].x, hv[q][nb].x, acc[nb], 0, 0, 0);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[q].y, hv[q][nb].y, acc[nb], 0, 0, 0);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[q].z, hv[q][nb].z, acc[nb], 0, 0, 0);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[q].w, hv[q][nb].w, acc[nb], 0, 0, 0);
        }
    };
    if (KB % UN == 0) {
        float4 wa[UN], wb[UN], ha[UN][NB], hb2[UN][NB];
        const int NBt = KB / UN;
        load(wa, ha, 0);
        int bi = 0;
        // no conditional loads inside the loop: hipcc's wait-count insertion falls back to vmcnt(0) at a merge point, which
        // would serialise the prefetch with the chain
        for (; bi + 2 < NBt; bi += 2) {
            load(wb, hb2, (bi + 1) * UN);
            __builtin_amdgcn_sched_barrier(0);
            fma_batch(wa, ha);
            __builtin_amdgcn_sched_barrier(0);
            load(wa, ha, (bi + 2) * UN);
            __builtin_amdgcn_sched_barrier(0);
            fma_batch(wb, hb2);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (bi + 1 < NBt) {
            load(wb, hb2, (bi + 1) * UN);
            __builtin_amdgcn_sched_barrier(0);
            fma_batch(wa, ha);
            fma_batch(wb, hb2);
        } else {
            fma_batch(wa, ha);
        }
    } else {
        for (int kb = 0; kb < KB; ++kb) {
            const float4 w = w4[(size_t)kb * rows];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const float4 hv = *reinterpret_cast<const float4*>(hb + 4 * nb * vstride + 4 * kb);
                acc[nb] = __builtin_amdgcn_mfma_f32_4x4x1f32(w.x, hv.x, acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_4x4x1f32(w.y, hv.y, acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_4x4x1f32(w.z, hv.z, acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_4x4x1f32(w.w, hv.w, acc[nb], 0, 0, 0);
            }
        }
    }
}

#ifdef TTSC_ABLATE
#define WQ_TICK(i)                                                     \
    do {                                                               \
        if (a.prof && tid == 0) {                                      \
            const unsigned long long now_ = wall_clock64();            \
            pacc[i] += now_ - plast;                                   \
            plast = now_;                                              \
        }                                                              \
    } while (0)
#else
#define WQ_TICK(i) do {} while (0)
#endif

// NC members per tile, BU = NC = 4*NB utterances per tile
template <int NC, int NB>
__global__ __launch_bounds__(WQ_THREADS) void wr_quad_kernel(WqArgs a) {
    constexpr int BU = 4 * NB;
    static_assert(BU == NC, "member m samples utterance m of its tile");
    extern __shared__ __attribute__((aligned(16))) float sm[];   // hvec[BU][VH] | pvec[BU][VP] | gbuf[3*UPW][BU] | scr[S]
    const int H = a.H, UPW = a.UPW, S = a.S, SR = a.SR, PR = a.PR, NM = a.n_mel, I0P = a.I0P;
    const int R3 = 3 * UPW;
    // tile placement: workgroups are dealt to the XCDs round-robin, so the members of a tile take blockIdx values that are
    // congruent mod 8 (same XCD, same L2); GP = number of tiles rounded up to a multiple of 8, surplus workgroups leave at once
    const int xcd = blockIdx.x % WQ_XCDS, slot = blockIdx.x / WQ_XCDS;
    const int g = xcd * (a.GP / WQ_XCDS) + slot / NC, m = slot % NC;
    if (g >= a.G) return;
    const int VH = H + 4, VP = 256 + 4;   // +4: the four utterances a wave reads per LDS access land in different banks
    float* hvec = sm;
    float* pvec = hvec + (size_t)BU * VH;
    float* gbuf = pvec + (size_t)BU * VP;
    float* scr = gbuf + (size_t)R3 * BU;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int u = tid % BU;            // utterance slot (element-wise work)
    const int j = tid / BU;            // local hidden unit
    const int bu = g * BU + u;
    const bool uok = bu < a.B;
    const int bc = uok ? bu : a.B - 1;
    const int nu = min(BU, a.B - g * BU);
    const bool gru_thr = j < UPW;
    const int jc = gru_thr ? j : 0;
    const float* Whh = a.whh + (size_t)m * H * R3;
    const float* Wih = a.wih + (size_t)m * I0P * R3;
    const float* Wpre = a.wpre + (size_t)m * H * PR;
    const float* Wout = a.wout + (size_t)m * 256 * SR;
    float bih[3], w_int[3], w_lx[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        bih[q] = a.bih[(size_t)m * R3 + q * UPW + jc];
        const int k1 = a.I0 - 1, k2 = a.I0 >= 2 ? a.I0 - 2 : 0;
        w_lx[q] = Wih[((size_t)(k1 >> 2) * R3 + q * UPW + jc) * 4 + (k1 & 3)];
        w_int[q] = Wih[((size_t)(k2 >> 2) * R3 + q * UPW + jc) * 4 + (k2 & 3)];
    }
    // matrix-pipe ownership: lane l of a 64-row block holds rows 4*(l >> 2) + i, utterance (l & 3) + 4*nb
    const int mrow = 4 * (lane >> 2);
    const int mutt = lane & 3;
    const float* bhh_m = a.bhh + (size_t)m * R3;
    const float* bpre_m = a.bpre + (size_t)m * PR;
    const float* bout_m = a.bout + (size_t)m * SR;
    const int nblk = (R3 + 63) >> 6;
    float pmel[3] = {0, 0, 0}, plow[3] = {0, 0, 0};
    float hprev = 0.f;   // h_{t-1}[unit m*UPW + j][utterance u]: each (unit, utterance) has exactly one owner thread
    unsigned* cnt = a.cnt + (size_t)g * 4;
    unsigned* abort_word = a.cnt + (size_t)a.G * 4;
    float* xh = a.xh + (size_t)g * 2 * BU * H;
    float* xpre = a.xpre + (size_t)g * 2 * BU * 256;
    float* xlog = a.xlog + (size_t)g * 2 * BU * S;
    float* xlx = a.xlx + (size_t)g * 2 * BU;

    for (int i = tid; i < BU * VH; i += WQ_THREADS) hvec[i] = 0.f;   // h_{-1} = 0: fma(w, 0, acc) == acc
    __syncthreads();

#ifdef TTSC_ABLATE
    unsigned long long pacc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, plast = wall_clock64();
#endif
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
        WQ_TICK(0);
        // ---- phase A: GRU slice over h_{t-1} (already in hvec: staged by phase B of the previous step) ----
        for (int blk = wave; blk < nblk; blk += WQ_THREADS / 64) {
            const int r0 = blk * 64;
            f32x4_t acc[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[nb][i] = bhh_m[min(r0 + mrow + i, R3 - 1)];
            mfma_rows<NB, (NB == 1 ? 8 : 4)>(acc, Whh, R3, min(r0 + lane, R3 - 1), hvec, VH, H, lane);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (r0 + mrow + i < R3) gbuf[(r0 + mrow + i) * BU + mutt + 4 * nb] = acc[nb][i];
        }
        WQ_TICK(1);
        if (t > 0 && !wait_count(cnt + 3, (unsigned)t * (unsigned)nu, abort_word)) return;   // last_x of step t-1
        __syncthreads();
        WQ_TICK(2);
        if (gru_thr) {
            const float lx = (t > 0) ? ld_f32(xlx + (par ^ 1) * BU + u) : 0.f;
            float gi[3], gh[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                float acc = a.use_lowres ? plow[q] : pmel[q];
                if (a.use_lowres) acc = fmaf(w_int[q], a.interp[(size_t)bc * ((size_t)a.Tl * a.up_low) + t], acc);
                gi[q] = fmaf(w_lx[q], lx, acc);
                gh[q] = gbuf[(q * UPW + j) * BU + u];
            }
            const float r = ttsc_sigmoidf(gi[0] + gh[0]);
            const float z = ttsc_sigmoidf(gi[1] + gh[1]);
            const float rg = r * gh[2];
            const float nn = ttsc_tanhf(gi[2] + rg);
            const float d = hprev - nn;
            hprev = fmaf(z, d, nn);
            st_f32(xh + ((size_t)par * H + m * UPW + j) * BU + u, hprev);   // [k][u]: consecutive threads, consecutive words
        }
        publish(cnt + 0);
        WQ_TICK(3);
        // ---- phase B: pre-output slice (PR rows) over the full h_t ----
        if (!wait_count(cnt + 0, (unsigned)(t + 1) * NC, abort_word)) return;
        WQ_TICK(4);
        {
            const float* src = xh + (size_t)par * H * BU;
            for (int i = tid * 2; i < BU * H; i += WQ_THREADS * 2) {
                const float2 v = ld_f32x2(src + i);
                const int k = i / BU, uu = i % BU;   // BU is even: both words belong to unit k
                hvec[uu * VH + k] = v.x;
                hvec[(uu + 1) * VH + k] = v.y;
            }
        }
        __syncthreads();
        WQ_TICK(5);
        if (wave == 0) {
            f32x4_t acc[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[nb][i] = bpre_m[min(mrow + i, PR - 1)];
            mfma_rows<NB, (NB == 1 ? 8 : 4)>(acc, Wpre, PR, min(lane, PR - 1), hvec, VH, H, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    if (mrow + i < PR) st_f32(xpre + ((size_t)par * 256 + m * PR + mrow + i) * BU + mutt + 4 * nb, ttsc_tanhf(acc[nb][i]));
        }
        publish(cnt + 1);
        WQ_TICK(6);
        // ---- phase C: output slice (SR rows) over the full pre-output ----
        if (!wait_count(cnt + 1, (unsigned)(t + 1) * NC, abort_word)) return;
        WQ_TICK(7);
        {
            const float* src = xpre + (size_t)par * 256 * BU;
            for (int i = tid * 2; i < BU * 256; i += WQ_THREADS * 2) {
                const float2 v = ld_f32x2(src + i);
                const int k = i / BU, uu = i % BU;
                pvec[uu * VP + k] = v.x;
                pvec[(uu + 1) * VP + k] = v.y;
            }
        }
        __syncthreads();
        WQ_TICK(8);
        if (wave == 0) {
            f32x4_t acc[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[nb][i] = bout_m[min(mrow + i, SR - 1)];
            mfma_rows<NB, (NB == 1 ? 8 : 4)>(acc, Wout, SR, min(lane, SR - 1), pvec, VP, 256, lane);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int s_ = m * SR + mrow + i, uu = mutt + 4 * nb;
                    if (mrow + i < SR) {
                        st_f32(xlog + ((size_t)par * BU + uu) * S + s_, acc[nb][i]);
                        if (a.out_logits && g * BU + uu < a.B) a.out_logits[((size_t)(g * BU + uu) * a.L + t) * S + s_] = acc[nb][i];
                    }
                }
        }
        publish(cnt + 2);
        WQ_TICK(9);
        // ---- phase D: member m samples utterance m of the tile ----
        if (m < nu) {
            if (!wait_count(cnt + 2, (unsigned)(t + 1) * NC, abort_word)) return;
            WQ_TICK(10);
            const int bs = g * BU + m;
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
                scr[tid] = ld_f32(xlog + ((size_t)par * BU + m) * S + tid) + g_;
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
                    st_f32(xlx + par * BU + m, a.forced_x ? a.forced_x[o] : wv);
                }
            }
            publish(cnt + 3);
            WQ_TICK(11);
        }
        if (++fr_phase == a.up) { fr_phase = 0; ++fr; }
        if (++lo_phase == a.up_low) { lo_phase = 0; ++lo; }
    }
#ifdef TTSC_ABLATE
    if (a.prof && tid == 0)
        for (int i = 0; i < 16; ++i) a.prof[(size_t)blockIdx.x * 16 + i] = pacc[i];
#endif
}

}  // namespace ttsc
