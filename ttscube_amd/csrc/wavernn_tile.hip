// WaveRNN decode with every tile of 8 utterances spread over 8 workgroups ("members") — gfx950.
//
// The single-workgroup kernel (wavernn.hip) streams the whole fp32 weight set (3.8 MB for H=512) from L2 every step and is
// bound by ONE CU's vector-memory path (64 B/clk): 45 us per step at B = 256.  Here 8 workgroups step 8 utterances
// together; member m owns 1/8 of the rows of every matrix (H/8 hidden units x 3 gates, 32 rows of the pre-output layer,
// S/8 rows of the output layer).  Its pre-output and output slices stay in LDS for the whole decode (98 KB at H = 512); only
// the recurrent slice (393 KB) is streamed from L2 each step.  B = 256 still fills all 256 CUs (32 tiles).
//
// Arithmetic.  The row x utterance products run on the matrix pipe as v_mfma_f32_4x4x1_16B_f32: one instruction is 16
// independent 4 (rows) x 4 (utterances) rank-1 updates  acc += w[row][k] * h[k][utt], one fused multiply-add per accumulator
// and per k — bit for bit the k-ordered fmaf chain of wavernn.hip / oracle/wavernn_ref.c (tools/probes/mfma_f32_exact.hip
// checks the instruction against fmaf, denormals included).  Lane l loads the 16-byte weight word of ITS row only and reads
// the h words of ITS utterance, so no lane duplicates a neighbour's global load (round 1's version of this kernel did — 4
// lanes per row — and was bound by exactly that).  Rows are split across members, never the reduction: every
// (row, utterance) is one k-ordered chain seeded with the bias, so indices and logits stay bit-exact.
//
// Schedule.  A step is  h_t -> pre_t -> logits_t -> sample_t -> (gates of step t+1).  Only the last link needs the sample:
// the recurrent product W_hh . h_t does not, and it is 80 % of the bytes.  So the workgroup forks after h_t is staged:
//     waves 4..7  "tail" of step t, the two output Linears: wave 4 + q owns QUARTER q of either contraction (round 6: the contract of
//                 oracle/wavernn_ref.c::matvec_chain4 — four k-ordered chains over consecutive quarters, added in order) — its quarter of the
//                 member's pre-output slice (H / 4 dependent matrix instructions instead of H fused multiply-adds on one lane: the workgroup
//                 timeline of round 6 put 9.4 of the step's 16.0 us into that chain and its hand-off, profiles/r06_wavernn_phase_timeline.log),
//                 the four partials added through LDS, tanh, hand-off; then it gathers ITS quarter of everybody's pre-output vector and runs
//                 its quarter of the output slice (64 matrix instructions) on it
//     wave 0      draws the step's Gumbel noise meanwhile, adds the four output partials, reduces / publishes the candidates, samples
//     waves 1..3  recurrent product of step t+1 (three 64-row blocks, two 4-utterance accumulators each)
// and joins for the gate math of step t+1 and the h_{t+1} hand-off.
//
// Hand-offs.  Every exchanged value is an 8-byte granule {fp32 value, step tag} written with ONE agent-scope store; a
// consumer lane polls the granules it needs until they carry the tag of the step (bounded spin, shared abort word).  No
// separate flag or counter, no producer-side drain, no barrier on the consumer side beyond the one that publishes the staged
// vector in LDS.  Buffers are double-buffered by step parity: a member overwrites the step-t granule only at step t+2, which
// it cannot reach before every member has published step t+1, i.e. has consumed step t.  (tools/probes/handoff_probe.hip:
// 0.75 us per hand-off with agent-scope accesses, same or different XCD; workgroup-scope accesses are NOT coherent across
// CUs.)  Members of a tile are still placed on one XCD (blockIdx -> XCD is round-robin).
#include "rnn_chain.hpp"
#include "wavernn_sampler.hpp"

namespace ttsc {

constexpr int WT_NC = 8;          // members per tile = utterances per tile
constexpr int WT_THREADS = 512;
constexpr int WT_XCDS = 8;
#ifndef WT_STREAM_UN
#define WT_STREAM_UN 8   // k-blocks of the recurrent weight stream in flight per wave (16-byte words per lane)
#endif
#ifndef WT_TAIL_PRIO
#define WT_TAIL_PRIO 0   // issue priority of waves 4..7 while they run the output Linears.  Measured (profiles/r06_wavernn_phase_timeline.log): 2 shortens the tail (wave 0 waits 5.7 instead of 7.2 us) and lengthens the recurrent product behind it by more — 14.5 against 14.1 us per step: off
#endif
constexpr unsigned WT_SPIN_LIMIT = 1u << 20;   // bounded spins: a member that is not resident must not hang the GPU

typedef unsigned long long u64;
typedef float f32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void st_granule(u64* p, float v, unsigned tag) {
    __hip_atomic_store(p, ((u64)tag << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Poll N granules (p[i * stride]) until all carry `tag`; false after a timeout or when another member aborted.
template <int N>
__device__ __forceinline__ bool ld_granules(const u64* p, int stride, unsigned tag, float (&v)[N], unsigned* abort_word) {
    u64 g[N];
    unsigned spins = 0;
    for (;;) {
#pragma unroll
        for (int i = 0; i < N; ++i) g[i] = __hip_atomic_load(p + (size_t)i * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool all = true;
#pragma unroll
        for (int i = 0; i < N; ++i) all = all && ((unsigned)(g[i] >> 32) == tag);
        if (all) break;
        if (++spins > WT_SPIN_LIMIT || ((spins & 63u) == 0u && __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
            __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
        __builtin_amdgcn_s_sleep(1);   // (longer back-offs only add latency: measured 18.1 / 18.7 / 20.8 us per step for 0 / 1k / 4k clocks)
    }
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = __uint_as_float((unsigned)g[i]);
    return true;
}

// same, granules p[off[i]]
template <int N>
__device__ __forceinline__ bool ld_granules_at(const u64* p, const int (&off)[N], unsigned tag, float (&v)[N], unsigned* abort_word) {
    u64 g[N];
    unsigned spins = 0;
    for (;;) {
#pragma unroll
        for (int i = 0; i < N; ++i) g[i] = __hip_atomic_load(p + off[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool all = true;
#pragma unroll
        for (int i = 0; i < N; ++i) all = all && ((unsigned)(g[i] >> 32) == tag);
        if (all) break;
        if (++spins > WT_SPIN_LIMIT || ((spins & 63u) == 0u && __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
            __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
        __builtin_amdgcn_s_sleep(1);   // (longer back-offs only add latency: measured 18.1 / 18.7 / 20.8 us per step for 0 / 1k / 4k clocks)
    }
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = __uint_as_float((unsigned)g[i]);
    return true;
}

struct WtArgs {
    const float* mel;      // [B, T, n_mel]
    const float* interp;   // [B, Tl*up_low]
    const float* feats;    // [B, 20, Tl]
    // per-member row slices, packed [K/4][rows][4]; member m at offset m * (K * rows)
    const float* whh;      // rows = 3*UPW (gate q, local unit j -> q*UPW + j), K = H
    const float* wih;      // rows = 3*UPW, K = I0P (in_dim rounded up to 4, zero padded)
    const float* bih;      // [NC][3*UPW]
    const float* bhh;      // [NC][3*UPW]
    const float* wpre;     // rows = 32 (= 256/NC), K = H
    const float* bpre;     // [NC][32]
    const float* wout;     // rows = 32 (first SR = SP/NC real, rest zero), K = 256; SP = S rounded up to a multiple of NC (rows >= S zero)
    const float* bout;     // [NC][32]
    // second GRU layer (NL == 2; input = h1, K = H): same member slices and packing as whh
    const float* whh2;
    const float* wih2;
    const float* bih2;     // [NC][3*UPW]
    const float* bhh2;     // [NC][3*UPW]
    const float* lut;
    const float* noise;    // [B, L, S] or null
    const float* forced_x; // [B, L] or null
    uint8_t* out_idx;
    float* out_wav;
    float* out_logits;
    // exchange area (device memory, zeroed before the launch), per tile g, in granules
    u64* xh;               // [G][2][H][8]
    u64* xh2;              // [G][2][H][8]   state of the second layer (NL == 2)
    u64* xpre;             // [G][2][256][8]
    u64* xlog;             // [G][2][8][SP]
    u64* xlx;              // [G][2][8]
    unsigned* abort_word;
    int B, T, Tl, H, UPW, I0, I0P, use_lowres, up, up_low, S, SP, SR, n_mel, out_kind, mode, L, G, GP, NL;
    unsigned long long seed;
    unsigned long long* prof;   // -DTTSC_ABLATE: [workgroup][16] accumulated 100 MHz ticks per segment (thread 0), or null
};

// A k-ordered chain block on the 4x4x1 fp32 matrix instruction, NB accumulators (4 utterances each) per lane:
//   lane l streams the packed weight words w4[kb * wstride] of its own row and reads v[utt0 + 4*nb][4*kb .. 4*kb+3] from LDS;
//   it ends up with rows (4 * (l >> 2) + i) of its 16 blocks in register i.  W_LDS: the weight words come from LDS.
// Both operand streams are software-pipelined by hand (two register sets of UN k-blocks): the words of batch i+1 are issued
// before the chain of batch i runs, so neither latency sits between two dependent matrix instructions.
template <int NB, int UN>
__device__ __forceinline__ void mfma_chain(f32x4_t (&acc)[NB], const float4* w4, int wstride, const float* hb, int vstride, int K) {
    const int KB = K >> 2;
    auto load = [&](float4 (&w)[UN], float4 (&hv)[UN][NB], int kb0) {
#pragma unroll
        for (int q = 0; q < UN; ++q) w[q] = w4[(size_t)(kb0 + q) * wstride];
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
            const float4 w = w4[(size_t)kb * wstride];
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

// One k-ordered fmaf chain per LANE on the vector ALU, both operands in LDS (w4[kb * wstride] = four consecutive k of the lane's row, h = the
// lane's utterance): a dependent v_fma_f32 issues every ~7 clocks against ~14 for the 4x4x1 matrix instruction, so for the short chains that
// sit on the critical path of a step (the pre-output layer) the vector ALU halves the latency — same fused multiply-adds, same order, same
// bits.  Operands are prefetched UN k-blocks ahead (two register sets).
template <int UN>
__device__ __forceinline__ float fma_chain_lds(const float4* w4, int wstride, const float* h, int K, float acc) {
    const int KB = K >> 2;
    float4 wa[UN], ha[UN], wb[UN], hb[UN];
    auto load = [&](float4 (&w)[UN], float4 (&hv)[UN], int kb0) {
#pragma unroll
        for (int q = 0; q < UN; ++q) {
            w[q] = w4[(size_t)(kb0 + q) * wstride];
            hv[q] = *reinterpret_cast<const float4*>(h + 4 * (kb0 + q));
        }
    };
    auto run = [&](const float4 (&w)[UN], const float4 (&hv)[UN]) {
#pragma unroll
        for (int q = 0; q < UN; ++q) {
            acc = fmaf(w[q].x, hv[q].x, acc);
            acc = fmaf(w[q].y, hv[q].y, acc);
            acc = fmaf(w[q].z, hv[q].z, acc);
            acc = fmaf(w[q].w, hv[q].w, acc);
        }
    };
    const int NBt = KB / UN;   // KB is a multiple of UN (H % 32 == 0, UN = 8)
    load(wa, ha, 0);
    int bi = 0;
    for (; bi + 2 < NBt; bi += 2) {
        load(wb, hb, (bi + 1) * UN);
        __builtin_amdgcn_sched_barrier(0);
        run(wa, ha);
        __builtin_amdgcn_sched_barrier(0);
        load(wa, ha, (bi + 2) * UN);
        __builtin_amdgcn_sched_barrier(0);
        run(wb, hb);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (bi + 1 < NBt) {
        load(wb, hb, (bi + 1) * UN);
        __builtin_amdgcn_sched_barrier(0);
        run(wa, ha);
        run(wb, hb);
    } else {
        run(wa, ha);
    }
    return acc;
}

// The same chain with the weight words streamed from global memory: their prefetch distance is one batch of UN k-blocks
// (L2 latency), while the h words (LDS latency) are read only two k-blocks ahead — which keeps the register count of the
// two-accumulator recurrent blocks at ~100 instead of ~200.
template <int NB, int UN>
// Wu = WAVE-UNIFORM base of the weight block, lrow = this lane's row: every load is (scalar base of k-block kb0 + q) + (one 32-bit lane
// offset) — written as w4 + lane and stepped by a run-time stride the compiler kept UN running 64-bit per-lane pointers (16 VGPRs),
// five of which lived in scratch and were reloaded, one dependent scratch_load -> global_load pair after the other, in every step.
__device__ __forceinline__ void mfma_chain_g(f32x4_t (&acc)[NB], const float4* Wu, int lrow_, int wstride, const float* hb, int vstride, int K) {
    static_assert(UN % 4 == 0, "UN is consumed in pairs of pairs");
    const int KB = K >> 2;
    // the lane offset is made opaque HERE, inside the caller's step loop: the first batch's addresses are loop-invariant, and hoisted out of the
    // step loop they are 64-bit per-lane values again (that is what was spilled); recomputed per step they are scalar base + this one register
    unsigned lrow = (unsigned)lrow_;
    asm volatile("" : "+v"(lrow));
    auto loadw = [&](float4 (&w)[UN], int kb0) {
#pragma unroll
        for (int q = 0; q < UN; ++q) {
            const float4* rb = Wu + (size_t)(kb0 + q) * wstride;   // uniform
            w[q] = rb[lrow];
        }
    };
    auto readh = [&](float4 (&hv)[2][NB], int kb) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) hv[q][nb] = *reinterpret_cast<const float4*>(hb + 4 * nb * vstride + 4 * (kb + q));
    };
    auto mf2 = [&](const float4& w0, const float4& w1, const float4 (&hv)[2][NB]) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_4x4x1f32(w0.x, hv[0][nb].x, acc[nb], 0, 0, 0);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_4x4x1f32(w0.y, hv[0][nb].y, acc[nb], 0, 0, 0);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_4x4x1f32(w0.z, hv[0][nb].z, acc[nb], 0, 0, 0);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_4x4x1f32(w0.w, hv[0][nb].w, acc[nb], 0, 0, 0);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_4x4x1f32(w1.x, hv[1][nb].x, acc[nb], 0, 0, 0);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_4x4x1f32(w1.y, hv[1][nb].y, acc[nb], 0, 0, 0);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_4x4x1f32(w1.z, hv[1][nb].z, acc[nb], 0, 0, 0);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_4x4x1f32(w1.w, hv[1][nb].w, acc[nb], 0, 0, 0);
    };
    auto fma_batch = [&](const float4 (&w)[UN], int kb0) {
        float4 h0[2][NB], h1[2][NB];
        readh(h0, kb0);
#pragma unroll
        for (int q = 0; q < UN; q += 4) {
            readh(h1, kb0 + q + 2);
            __builtin_amdgcn_sched_barrier(0);
            mf2(w[q], w[q + 1], h0);
            __builtin_amdgcn_sched_barrier(0);
            if (q + 4 < UN) readh(h0, kb0 + q + 4);
            __builtin_amdgcn_sched_barrier(0);
            mf2(w[q + 2], w[q + 3], h1);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    if (KB % UN == 0) {
        float4 wa[UN], wb[UN];
        const int NBt = KB / UN;
        loadw(wa, 0);
        int bi = 0;
        for (; bi + 2 < NBt; bi += 2) {   // no conditional loads inside the loop (see mfma_chain)
            loadw(wb, (bi + 1) * UN);
            __builtin_amdgcn_sched_barrier(0);
            fma_batch(wa, bi * UN);
            loadw(wa, (bi + 2) * UN);
            __builtin_amdgcn_sched_barrier(0);
            fma_batch(wb, (bi + 1) * UN);
        }
        if (bi + 1 < NBt) {
            loadw(wb, (bi + 1) * UN);
            __builtin_amdgcn_sched_barrier(0);
            fma_batch(wa, bi * UN);
            fma_batch(wb, (bi + 1) * UN);
        } else {
            fma_batch(wa, bi * UN);
        }
    } else {
        mfma_chain<NB, 1>(acc, Wu + lrow, wstride, hb, vstride, K);
    }
}

// One 64-row block of a member's (3*UPW rows) x (8 utterances) product  out[row][utt] = bias[row] + sum_k W[row][k] v[utt][k]  on one
// wave: two 4-utterance accumulators per lane, weights streamed from global memory.  NOT inlined on purpose: the two-layer kernel
// runs three such products per step and three inlined copies of the software-pipelined chain (two register sets of weight words
// each) push the kernel ~90 VGPRs into scratch; one out-of-line copy keeps it in registers.
// bias / v / out live in LDS: typed as address-space-3 pointers so that the out-of-line body still uses ds_read / ds_write (a generic
// pointer would turn every h read of the chain into a flat load).
typedef __attribute__((address_space(3))) float lds_float;
typedef __attribute__((address_space(1))) float glb_float;   // (likewise global: a generic pointer would make the weight stream flat loads,
                                                             // whose lgkmcnt accounting serialises them with the LDS reads of the chain)
__device__ __noinline__ void wt_row_block(const glb_float* W_g, const lds_float* bias_l, const lds_float* v_l, lds_float* out_l, int r0,
                                         int R3, int VH, int H, int BU) {
    const float* W = (const float*)W_g;
    const float* bias = (const float*)bias_l;
    const float* v = (const float*)v_l;
    float* out = (float*)out_l;
    const int lane = threadIdx.x & 63;
    const int mrow = 4 * (lane >> 2), mutt = lane & 3;
    f32x4_t acc[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[nb][i] = bias[min(r0 + mrow + i, R3 - 1)];
    // (an out-of-line function receives its arguments in vector registers: tell the compiler which of them are wave-uniform)
    const unsigned long long wu = (unsigned long long)reinterpret_cast<uintptr_t>(W);
    const unsigned w_lo = __builtin_amdgcn_readfirstlane((unsigned)wu), w_hi = __builtin_amdgcn_readfirstlane((unsigned)(wu >> 32));
    const float4* Wu = reinterpret_cast<const float4*>((uintptr_t)(((unsigned long long)w_hi << 32) | w_lo));
    const int R3u = __builtin_amdgcn_readfirstlane(R3), Hu = __builtin_amdgcn_readfirstlane(H);
    mfma_chain_g<2, WT_STREAM_UN>(acc, Wu, min(r0 + lane, R3 - 1), R3u, v + mutt * VH, VH, Hu);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (r0 + mrow + i < R3) out[(r0 + mrow + i) * BU + mutt + 4 * nb] = acc[nb][i];
}

#ifdef TTSC_ABLATE
#define WT_TICK(i)                                                     \
    do {                                                               \
        if (a.prof && tid == 0) {                                      \
            const unsigned long long now_ = wall_clock64();            \
            pacc[i] += now_ - plast;                                   \
            plast = now_;                                              \
        }                                                              \
    } while (0)
// the same for lane 0 of wave 4 (the output Linears' quarter 0): slots 8..
#define WT_TICKW(i)                                                    \
    do {                                                               \
        if (a.prof && tid == 256) {                                    \
            const unsigned long long now_ = wall_clock64();            \
            pacc[i] += now_ - plast;                                   \
            plast = now_;                                              \
        }                                                              \
    } while (0)
#else
#define WT_TICK(i) do {} while (0)
#define WT_TICKW(i) do {} while (0)
#endif

// CONT: continuous output head (MOL / Gaussian / Beta) — a separate instantiation, so that the discrete kernel does not carry
// the samplers' registers.
// L2: two GRU layers (the reference class default, cube/networks/modules.py:392-400).  Member m also owns its H/8 units of the
// second layer.  Per step the second layer adds ONE product to the critical path — W_ih2 . h1_t, which needs this step's h1 —
// and one hand-off (h2_t); its recurrent product W_hh2 . h2_{t-1} rides with W_hh1 . h1_{t-1} on waves 1..3 behind the tail of
// the previous step.  All three products stream their member slice from L2 / Infinity Cache (1.18 MB per member and step).
template <bool CONT, bool L2>
__global__ __launch_bounds__(WT_THREADS) void wr_tile_kernel(WtArgs a) {
    constexpr int NC = WT_NC, BU = WT_NC, PR = 256 / WT_NC;   // PR = 32 pre-output rows per member
    // LDS: wpre[H/4][32][4] | wout[64][32][4] | hvec[BU][VH] | pvec[BU][VP] | gbuf[3*UPW][BU] | bias[32 + 32 + 3*UPW] | tail_fail, pre_ready
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int H = a.H, UPW = a.UPW, S = a.S, SP = a.SP, SR = a.SR, NM = a.n_mel, I0P = a.I0P;
    const int R3 = 3 * UPW;
    // tile placement: workgroups are dealt to the XCDs round-robin, so the members of a tile take blockIdx values that are
    // congruent mod 8 (same XCD, same L2); GP = number of tiles rounded up to a multiple of 8, surplus workgroups leave at once
    const int xcd = blockIdx.x % WT_XCDS, slot = blockIdx.x / WT_XCDS;
    const int g = xcd * (a.GP / WT_XCDS) + slot / NC, m = slot % NC;
    if (g >= a.G) return;
    const int VH = H + 4, VP = 256 + 4;   // +4: the utterances a wave reads per LDS access land in different banks
    float* wpreL = sm;
    float* woutL = wpreL + (size_t)H * PR;
    float* hvec = woutL + (size_t)256 * PR;
    float* pvec = hvec + (size_t)BU * VH;
    float* gbuf = pvec + (size_t)BU * VP;
    float* biasL = gbuf + (size_t)R3 * BU;   // bpre[32] | bout[32] | bhh[R3]
    int* tail_fail = reinterpret_cast<int*>(biasL + 64 + R3);
    int* pre_ready = tail_fail + 1;   // helper waves that have staged the pre-output vector (monotonic)
    float* ybuf = reinterpret_cast<float*>(tail_fail + 4);   // [32]: the output values of utterance m (continuous heads)
    float* lxv = ybuf + 32;   // [8]: the fed-back samples of the 8 utterances (discrete heads: every member derives them itself)
    // second layer: h2 vector of the 8 utterances | W_hh2 h2 products | b_ih2, b_hh2   (W_ih2 h1 products reuse gbuf: the
    // first layer's gate math has consumed it by then)
    float* hvec2 = ybuf + 60;
    float* gbuf2 = hvec2 + (size_t)BU * VH;
    float* bias2 = gbuf2 + (size_t)R3 * BU;   // bih2[R3] | bhh2[R3]
    float* hvecT = L2 ? hvec2 : hvec;         // what the tail (pre-output layer) reads: the LAST layer's state
    // partial sums of the two output Linears, [quarter][row of the member's slice (32)][utterance]: written by wave 4 + quarter, added in quarter order
    float* ppart = L2 ? bias2 + 2 * R3 : ybuf + 60;
    float* opart = ppart + 4 * PR * BU;
    int* part_ready = tail_fail + 2;   // waves 4..7 that have written their pre-output partial (monotonic)
    int* out_ready = tail_fail + 3;    // ... their output partial (monotonic)
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
    float bih[3], w_int[3], w_lx[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        bih[q] = a.bih[(size_t)m * R3 + q * UPW + jc];
        const int k1 = a.I0 - 1, k2 = a.I0 >= 2 ? a.I0 - 2 : 0;
        w_lx[q] = Wih[((size_t)(k1 >> 2) * R3 + q * UPW + jc) * 4 + (k1 & 3)];
        w_int[q] = Wih[((size_t)(k2 >> 2) * R3 + q * UPW + jc) * 4 + (k2 & 3)];
    }
    // matrix-pipe ownership
    //   recurrent blocks (64 rows, two accumulators): lane l holds rows 4*(l >> 2) + i, utterances (l & 3) and (l & 3) + 4
    //   pre / output blocks (32 rows, one accumulator): lanes 0..31 take utterances 0..3, lanes 32..63 the same rows for 4..7
    const int mrow = 4 * (lane >> 2);
    const int mutt = lane & 3;
    const int trow = 4 * ((lane & 31) >> 2);
    const int tutt = (lane & 3) + 4 * (lane >> 5);
    const float* bpre_m = biasL;
    const float* bout_m = biasL + 32;
    const float* bhh_m = biasL + 64;
    const int nblk = (R3 + 63) >> 6;
    float pmel[3] = {0, 0, 0}, plow[3] = {0, 0, 0};
    float hprev = 0.f;   // h_{t-1}[unit m*UPW + j][utterance u]: each (unit, utterance) has exactly one owner thread
    float hprev2 = 0.f;  // the same for the second layer
    u64* xh = a.xh + (size_t)g * 2 * BU * H;
    u64* xh2 = L2 ? a.xh2 + (size_t)g * 2 * BU * H : nullptr;
    const float* Whh2 = L2 ? a.whh2 + (size_t)m * H * R3 : nullptr;
    const float* Wih2 = L2 ? a.wih2 + (size_t)m * H * R3 : nullptr;
    u64* xpre = a.xpre + (size_t)g * 2 * BU * 256;
    u64* xlog = a.xlog + (size_t)g * 2 * BU * SP;
    u64* xlx = a.xlx + (size_t)g * 2 * BU;

    // resident slices of the pre-output and output layers, h_{-1} = 0 (fma(w, 0, acc) == acc)
    {
        const float4* sp = reinterpret_cast<const float4*>(a.wpre + (size_t)m * H * PR);
        float4* dp = reinterpret_cast<float4*>(wpreL);
        for (int i = tid; i < H * PR / 4; i += WT_THREADS) dp[i] = sp[i];
        const float4* so = reinterpret_cast<const float4*>(a.wout + (size_t)m * 256 * PR);
        float4* dq = reinterpret_cast<float4*>(woutL);
        for (int i = tid; i < 256 * PR / 4; i += WT_THREADS) dq[i] = so[i];
        for (int i = tid; i < BU * VH; i += WT_THREADS) hvec[i] = 0.f;
        for (int i = tid; i < 64 + R3; i += WT_THREADS)
            biasL[i] = i < 32 ? a.bpre[(size_t)m * PR + i] : (i < 64 ? a.bout[(size_t)m * PR + i - 32] : a.bhh[(size_t)m * R3 + i - 64]);
        if (tid == 0) {
            *tail_fail = 0;
            *pre_ready = 0;
            *part_ready = 0;
            *out_ready = 0;
        }
        if (L2) {
            for (int i = tid; i < BU * VH; i += WT_THREADS) hvec2[i] = 0.f;
            for (int i = tid; i < 2 * R3; i += WT_THREADS) bias2[i] = i < R3 ? a.bih2[(size_t)m * R3 + i] : a.bhh2[(size_t)m * R3 + i - R3];
        }
    }
    __syncthreads();
#ifdef TTSC_ABLATE
    unsigned long long pacc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, plast = wall_clock64();
#endif

    // wait until the workgroup-scope counter `cnt` (bumped once per wave 4..7 and step) has reached 4 * (s + 1)
    auto wait_four = [&](int* cnt, int s) -> bool {
        unsigned spins = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4 * (s + 1)) {
            if (++spins > WT_SPIN_LIMIT) return false;
            __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        return true;
    };
    auto signal_one = [&](int* cnt) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };

    // The two output Linears of step s on waves 4..7 (hvecT holds the last layer's h_s); wave 4 + q owns quarter q of both contractions.
    //   1. quarter q of the member's pre-output slice: 32 rows x 8 utterances over the H / 4 inputs [q H / 4, (q + 1) H / 4) on the matrix pipe
    //      (lanes 0..31 utterances 0..3, lanes 32..63 the same rows for 4..7; H / 4 dependent 4x4x1 instructions), quarter 0 seeded with the bias
    //   2. partials through LDS; once all four are there, thread (row, utterance) adds them in quarter order, tanh, publishes the granule
    //   3. the wave gathers the rows [64 q, 64 q + 64) of EVERYBODY's pre-output vector — the inputs of its quarter of the output layer — into pvec
    //   4. quarter q of the output slice (64 dependent instructions) -> opart; wave 0 adds the quarters (tail)
    auto linears_quarter = [&](int s) {
        const int q = wave - 4;
        const int KBq = H >> 4;   // 4-input blocks per quarter of the pre-output contraction (H % 32 == 0)
        WT_TICKW(8);
        // these two short dependent chains ARE the step's critical path, and waves 5..7 share their SIMD's matrix pipe with the recurrent product of
        // waves 1..3 (two independent accumulators: it fills every slot it is given, and has ~10 us to do 4 us of work): the tail goes first
        if (WT_TAIL_PRIO) __builtin_amdgcn_s_setprio(WT_TAIL_PRIO);
        {
            f32x4_t acc[1];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[0][i] = q == 0 ? bpre_m[trow + i] : 0.f;
            mfma_chain<1, 4>(acc, reinterpret_cast<const float4*>(wpreL) + (size_t)q * KBq * PR + (lane & 31), PR, hvecT + tutt * VH + 4 * q * KBq, 0, 4 * KBq);
#pragma unroll
            for (int i = 0; i < 4; ++i) ppart[(q * PR + trow + i) * BU + tutt] = acc[0][i];
        }
        WT_TICKW(9);
        signal_one(part_ready);
        bool okq = wait_four(part_ready, s);
        WT_TICKW(10);
        {
            const int hl = tid - 4 * 64;   // 0..255
            const int row = hl & 31, utt = hl >> 5;
            const float v = ((ppart[row * BU + utt] + ppart[(PR + row) * BU + utt]) + ppart[(2 * PR + row) * BU + utt]) + ppart[(3 * PR + row) * BU + utt];
            st_granule(xpre + ((size_t)(s & 1) * 256 + m * PR + row) * BU + utt, ttsc_tanhf(v), (unsigned)s + 1u);
        }
        WT_TICKW(11);
        {
            const u64* src = xpre + ((size_t)(s & 1) * 256 + 64 * q) * BU + lane;
            float v[8];
            okq = ld_granules<8>(src, 64, (unsigned)s + 1u, v, a.abort_word) && okq;
            WT_TICKW(12);
#pragma unroll
            for (int r = 0; r < 8; ++r) pvec[(lane & 7) * VP + 64 * q + r * 8 + (lane >> 3)] = v[r];
            if (!__all(okq) && lane == 0) *tail_fail = 1;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the chain below reads what this wave's own lanes just wrote
            __builtin_amdgcn_wave_barrier();
        }
        {
            f32x4_t acc[1];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[0][i] = q == 0 ? bout_m[trow + i] : 0.f;
            mfma_chain<1, 4>(acc, reinterpret_cast<const float4*>(woutL) + (size_t)q * 16 * PR + (lane & 31), PR, pvec + tutt * VP + 64 * q, 0, 64);
#pragma unroll
            for (int i = 0; i < 4; ++i) opart[(q * PR + trow + i) * BU + tutt] = acc[0][i];
        }
        signal_one(out_ready);
        if (WT_TAIL_PRIO) __builtin_amdgcn_s_setprio(0);
        WT_TICKW(13);
    };

    // The tail of step s on wave 0 (hvec holds h_s): output slice, candidates / sample.  Tag of step s = s + 1.
    auto tail = [&](int s) -> bool {
        const int par = s & 1;
        const unsigned tag = (unsigned)s + 1u;
        bool ok = true;
        // While waves 4..7 compute and gather the pre-output vector, this wave draws the step's Gumbel noise for its classes (it depends on
        // (class, step, utterance) only): off the critical path instead of behind the output chain.
        float gn[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (!CONT) {
            const int bs = g * BU + tutt, s0 = m * SR + trow;   // s0 is a multiple of 4: one Philox block covers this lane's classes
            if (a.mode == 1 && bs < a.B) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (trow + i < SR && s0 + i < S) gn[i] = a.noise[((size_t)bs * a.L + s) * S + s0 + i];
            } else if (a.mode == 2) {
                uint32_t r4[4];
                ttsc_philox4x32((uint32_t)(s0 >> 2), (uint32_t)s, (uint32_t)bs, 0u, (uint32_t)a.seed, (uint32_t)(a.seed >> 32), r4);
#pragma unroll
                for (int i = 0; i < 4; ++i) gn[i] = ttsc_gumbel(r4[i]);
            }
        }
        WT_TICK(4);
        // the four quarters of the member's output slice come from waves 4..7 (linears_quarter): wait for them, add them in quarter order
        if (!wait_four(out_ready, s)) ok = false;
        WT_TICK(5);
        {   // output layer: SR (<= 32) rows x 8 utterances over the 256 pre-output values
            f32x4_t acc[1];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[0][i] = ((opart[(trow + i) * BU + tutt] + opart[(PR + trow + i) * BU + tutt]) + opart[(2 * PR + trow + i) * BU + tutt]) +
                            opart[(3 * PR + trow + i) * BU + tutt];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int s_ = m * SR + trow + i;
                if (trow + i < SR) {
                    if constexpr (CONT) st_granule(xlog + ((size_t)par * BU + tutt) * SP + s_, acc[0][i], tag);
                    if (a.out_logits && s_ < S && g * BU + tutt < a.B) a.out_logits[((size_t)(g * BU + tutt) * a.L + s) * S + s_] = acc[0][i];
                }
            }
            if constexpr (!CONT) {
                // Discrete heads: the Gumbel-max is taken HIERARCHICALLY.  This member reduces its SR classes of every utterance to one
                // candidate (score, class) — first maximum wins, as in the sequential scan of the oracle — and publishes 8 candidates
                // instead of 8 x SR logits; every member then reduces the 8 x 8 candidates itself (below), so all of them know all
                // eight samples and the fed-back value needs no hand-off of its own (three edges per step instead of four).
                const int s0 = m * SR + trow;
                float best = -INFINITY;
                int bi = 1 << 20;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float sc = acc[0][i] + gn[i];
                    if (trow + i < SR && s0 + i < S && (bi == (1 << 20) || sc > best)) {
                        best = sc;
                        bi = s0 + i;
                    }
                }
#pragma unroll
                for (int off2 = 4; off2 <= 16; off2 <<= 1) {   // the 8 lanes that hold this utterance's other row groups
                    const float os = __shfl_xor(best, off2);
                    const int oi = __shfl_xor(bi, off2);
                    if (oi < (1 << 20) && (bi == (1 << 20) || os > best || (os == best && oi < bi))) {
                        best = os;
                        bi = oi;
                    }
                }
                if (((lane & 31) >> 2) == 0)   // candidate granule {score, (tag << 8) | class}: xlog area, [parity][utterance][member]
                    __hip_atomic_store(xlog + ((size_t)par * BU + tutt) * SP + m, ((u64)((tag << 8) | (unsigned)(bi & 255)) << 32) | (u64)__float_as_uint(best),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        WT_TICK(6);
        if constexpr (CONT) {
          if (m < nu) {   // continuous heads (MOL / Gaussian / Beta): S <= 30 output values of utterance m
            const int bs = g * BU + m;
            float v[1] = {0.f};
            if (lane < SP) ok = ld_granules<1>(xlog + ((size_t)par * BU + m) * SP + lane, 1, tag, v, a.abort_word) && ok;
            if (lane < 32) ybuf[lane] = v[0];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            float wv;
            int bi;
            const size_t o = (size_t)bs * a.L + s;
            wr_sample_continuous(a.out_kind, a.mode, ybuf, a.noise, o, s, bs, a.seed, lane, wv, bi);
            if (lane == 0) {
                a.out_idx[o] = (uint8_t)bi;
                a.out_wav[o] = wv;
                st_granule(xlx + par * BU + m, a.forced_x ? a.forced_x[o] : wv, tag);
            }
          }
        } else {   // discrete heads: every member reduces the 8 (members) x 8 (utterances) candidates; lane -> (utterance l >> 3, member l & 7)
            const int cu = lane >> 3, cm = lane & 7;
            const u64* src = xlog + ((size_t)par * BU + cu) * SP + cm;
            u64 gq;
            unsigned spins = 0;
            for (;;) {
                gq = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned)(gq >> 40) == (tag & 0xFFFFFFu)) break;
                if (++spins > WT_SPIN_LIMIT || ((spins & 63u) == 0u && __hip_atomic_load(a.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                    __hip_atomic_store(a.abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = false;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            float best = __uint_as_float((unsigned)gq);
            int bi = (int)((gq >> 32) & 255u);
#pragma unroll
            for (int off2 = 1; off2 <= 4; off2 <<= 1) {   // classes of a lower member are lower: "first maximum wins" = lower class on ties
                const float os = __shfl_xor(best, off2);
                const int oi = __shfl_xor(bi, off2);
                if (os > best || (os == best && oi < bi)) {
                    best = os;
                    bi = oi;
                }
            }
            if (cm == 0) {
                const int bs = g * BU + cu;
                const float wv = a.out_kind == 0 ? a.lut[bi] : (((float)bi / 255.0f) - 0.5f) * 2.0f;
                float fed = wv;
                if (bs < a.B) {
                    const size_t o = (size_t)bs * a.L + s;
                    if (a.forced_x) fed = a.forced_x[o];
                    if (cu == m) {   // the utterance's own member writes its outputs
                        a.out_idx[o] = (uint8_t)bi;
                        a.out_wav[o] = wv;
                    }
                }
                lxv[cu] = fed;   // read by the gate math of the next step after the join barrier
            }
        }
        WT_TICK(7);
        return __all(ok);
    };

    int fr = 0, fr_phase = 0, lo = 0, lo_phase = 0;
    for (int t = 0; t < a.L; ++t) {
        const int par = t & 1;
        WT_TICK(0);
        // ---- fork: wave 0 finishes step t-1, waves 1..3 run the recurrent product of step t (both read hvec = h_{t-1}) ----
        const float interp_t = (a.use_lowres && gru_thr) ? a.interp[(size_t)bc * ((size_t)a.Tl * a.up_low) + t] : 0.f;
        if (wave == 0) {
            if (t > 0 && !tail(t - 1)) *tail_fail = 1;
        } else if (wave >= 4) {
            if (t > 0) linears_quarter(t - 1);
        } else if (wave - 1 < nblk) {
            const int r0 = (wave - 1) * 64;
            if (L2) {   // (two-layer kernel: out-of-line block, see wt_row_block) recurrent products of BOTH layers for this step
                wt_row_block((const glb_float*)Whh, (const lds_float*)bhh_m, (const lds_float*)hvec, (lds_float*)gbuf, r0, R3, VH, H, BU);
                wt_row_block((const glb_float*)Whh2, (const lds_float*)(bias2 + R3), (const lds_float*)hvec2, (lds_float*)gbuf2, r0, R3, VH, H, BU);
            } else {
                f32x4_t acc[2];
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[nb][i] = bhh_m[min(r0 + mrow + i, R3 - 1)];
                mfma_chain_g<2, WT_STREAM_UN>(acc, reinterpret_cast<const float4*>(Whh), min(r0 + lane, R3 - 1), R3, hvec + mutt * VH, VH, H);
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (r0 + mrow + i < R3) gbuf[(r0 + mrow + i) * BU + mutt + 4 * nb] = acc[nb][i];
            }
        }
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
#pragma unroll 4
                for (int f = 0; f < 20; ++f) {
                    const int k = NM + f;
                    const float v = a.feats[((size_t)bc * 20 + f) * a.Tl + lo];
#pragma unroll
                    for (int q = 0; q < 3; ++q) plow[q] = fmaf(Wih[((size_t)(k >> 2) * R3 + q * UPW + j) * 4 + (k & 3)], v, plow[q]);
                }
            }
        }
        __syncthreads();   // join
        WT_TICK(1);
        // ---- gate math of step t: needs the sample of step t-1 of every utterance of the tile ----
        bool ok = true;
        if (gru_thr) {
            float lx[1] = {0.f};
            if constexpr (CONT) {
                if (t > 0 && u < nu) ok = ld_granules<1>(xlx + (par ^ 1) * BU + u, 1, (unsigned)t, lx, a.abort_word);
            } else if (t > 0) {
                lx[0] = lxv[u];   // written by wave 0 in the tail of step t-1, before the join barrier
            }
            float gi[3], gh[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                float acc = a.use_lowres ? plow[q] : pmel[q];
                if (a.use_lowres) acc = fmaf(w_int[q], interp_t, acc);
                gi[q] = fmaf(w_lx[q], lx[0], acc);
                gh[q] = gbuf[(q * UPW + j) * BU + u];
            }
            const float r = ttsc_sigmoidf(gi[0] + gh[0]);
            const float z = ttsc_sigmoidf(gi[1] + gh[1]);
            const float rg = r * gh[2];
            const float nn = ttsc_tanhf(gi[2] + rg);
            const float d = hprev - nn;
            hprev = fmaf(z, d, nn);
            st_granule(xh + ((size_t)par * H + m * UPW + j) * BU + u, hprev, (unsigned)t + 1u);   // [k][u]: consecutive threads, consecutive granules
        }
        WT_TICK(2);
        // ---- stage the full h_t of the 8 utterances: BU * H granules ----
        {
            const u64* src = xh + (size_t)par * H * BU;
            for (int i0 = tid; i0 < BU * H; i0 += 8 * WT_THREADS) {
                if (i0 + 7 * WT_THREADS < BU * H) {   // H = 512: one round trip for the whole vector
                    float v[8];
                    ok = ld_granules<8>(src + i0, WT_THREADS, (unsigned)t + 1u, v, a.abort_word) && ok;
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        const int i = i0 + r * WT_THREADS;
                        hvec[(i & 7) * VH + (i >> 3)] = v[r];
                    }
                } else {
                    for (int i = i0; i < BU * H; i += WT_THREADS) {
                        float v[1];
                        ok = ld_granules<1>(src + i, 1, (unsigned)t + 1u, v, a.abort_word) && ok;
                        hvec[(i & 7) * VH + (i >> 3)] = v[0];
                    }
                }
            }
        }
        if (__syncthreads_or((!ok) || *tail_fail)) return;   // also publishes hvec
        WT_TICK(3);
        if (L2) {
            // ---- second layer, critical part: W_ih2 . h1_t  (needs this step's h1; 3*UPW rows x 8 utterances on waves 1..3) ----
            if (wave >= 1 && wave - 1 < nblk) wt_row_block((const glb_float*)Wih2, (const lds_float*)bias2, (const lds_float*)hvec, (lds_float*)gbuf, (wave - 1) * 64, R3, VH, H, BU);   // (gbuf is free: the first layer's gates are done)
            __syncthreads();
            // ---- gate math of the second layer, h2_t hand-off and staging ----
            if (gru_thr) {
                float gi[3], gh[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    gi[q] = gbuf[(q * UPW + j) * BU + u];
                    gh[q] = gbuf2[(q * UPW + j) * BU + u];
                }
                const float r = ttsc_sigmoidf(gi[0] + gh[0]);
                const float z = ttsc_sigmoidf(gi[1] + gh[1]);
                const float rg = r * gh[2];
                const float nn = ttsc_tanhf(gi[2] + rg);
                const float d = hprev2 - nn;
                hprev2 = fmaf(z, d, nn);
                st_granule(xh2 + ((size_t)par * H + m * UPW + j) * BU + u, hprev2, (unsigned)t + 1u);
            }
            {
                const u64* src = xh2 + (size_t)par * H * BU;
                for (int i0 = tid; i0 < BU * H; i0 += 8 * WT_THREADS) {
                    if (i0 + 7 * WT_THREADS < BU * H) {
                        float v[8];
                        ok = ld_granules<8>(src + i0, WT_THREADS, (unsigned)t + 1u, v, a.abort_word) && ok;
#pragma unroll
                        for (int r = 0; r < 8; ++r) {
                            const int i = i0 + r * WT_THREADS;
                            hvec2[(i & 7) * VH + (i >> 3)] = v[r];
                        }
                    } else {
                        for (int i = i0; i < BU * H; i += WT_THREADS) {
                            float v[1];
                            ok = ld_granules<1>(src + i, 1, (unsigned)t + 1u, v, a.abort_word) && ok;
                            hvec2[(i & 7) * VH + (i >> 3)] = v[0];
                        }
                    }
                }
            }
            if (__syncthreads_or(!ok)) return;   // publishes hvec2
        }
        if (++fr_phase == a.up) { fr_phase = 0; ++fr; }
        if (++lo_phase == a.up_low) { lo_phase = 0; ++lo; }
    }
    if (wave >= 4) linears_quarter(a.L - 1);
    if (wave == 0) tail(a.L - 1);
#ifdef TTSC_ABLATE
    if (a.prof && tid == 0)
        for (int i = 0; i < 8; ++i) a.prof[(size_t)blockIdx.x * 16 + i] = pacc[i];
    if (a.prof && tid == 256)
        for (int i = 8; i < 16; ++i) a.prof[(size_t)blockIdx.x * 16 + i] = pacc[i];
#endif
}

}  // namespace ttsc
