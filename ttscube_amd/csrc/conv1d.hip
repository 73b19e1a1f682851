// Conv1d / ConvTranspose1d as an fp32-MFMA implicit GEMM for gfx950 (CDNA4).
//
//   out[b, co, q*os + oo] = act( ( bias[co] + resid + sum_{j<ntaps} sum_{ci} W_j[co, ci] *
//                                  lrelu(in_scale * x[b, ci, q + tap_base + j*tap_step]) ) * out_scale )
//
// A plain (dilated) Conv1d is one launch with tap_step = dilation, tap_base = -padding, os = 1.
// A ConvTranspose1d with stride s is s polyphase launches: phase r owns outputs o with
// (o + p) % s == r, its taps are k = r + j*s and it reads x[q - j]  (tap_step = -1).
//
// GEMM view per workgroup: M = output channels (MT rows), N = NT consecutive q positions, K = Cin*ntaps.
// v_mfma_f32_32x32x2_f32 is an exact k-ordered fp32 fmaf chain at the fp32 vector rate (157 TF/chip).  The B operand
// (activations, with the dilation halo) is staged per 16-channel chunk through double-buffered LDS with the leaky-relu
// prologue fused into the staging pass, the A operand (weights) is pre-packed into MFMA fragment order (on the host for
// inference, by pack_w_kernel from the live parameter for training) so every fragment is ONE coalesced 256-byte wave load
// that hits L2; both streams are software-pipelined explicitly (see conv_mfma_kernel).  Epilogue fuses bias, residual
// add, scaling, tanh, the running sum over residual blocks and — for data-gradient launches — the leaky-relu derivative.
// The split-precision kernels further down (conv_f16x3_kernel, respair32_f16x3_kernel) are the default for the generator.
#include <cmath>
#include <cstdlib>
#include <type_traits>

#include "common.hpp"
#include "conv_kernels.hpp"

namespace ttsc {

// ---------------------------------------------------------------------------------------------------------------
// Wide-tile variant for the square layers of the wide stages (Cin = Cout in {128, 256}, K in {3, 7, 11}, dilation 1/3/5).
// What bounds conv_f16x3_kernel there (PMC, round 1: MFMA pipe 47 % busy, 4.8 VALU per MFMA, a third of the wave cycles
// parked at barriers) is structural: a workgroup owns 64 output channels, so the same activation window is loaded,
// leaky-relu'ed and split by Cout/64 workgroups, and every channel chunk costs two barriers with nothing in flight.
// Here
//   * a wave owns 64 x 128 outputs (MI = 2, NJ = 4: 24 MFMAs per tap for 8 activation- and 4 weight-fragment reads), four
//     waves form a 128 x 256 workgroup tile: the activation window is staged once per 128 output channels;
//   * weight fragments travel global -> LDS by LDS-DMA one tap ahead (no VGPRs, two 1-KB instructions per wave and tap) and
//     are shared by the two waves of a row tile; the barrier that ends a tap publishes them.  (Round-2 ablation: per-wave
//     global -> register fragment loads cost ~30 % of this kernel's time in the CU's vector-memory path.)
//   * the activation tile is double-buffered in LDS: chunk c+1 is converted and written while chunk c is multiplied
//     (the conversion's VALU work sits between MFMAs of the same wave);
//   * the global loads of chunk c+2 are issued right after chunk c+1 left the staging registers.
// K and the dilation are template parameters: the tap loop is unrolled and every LDS address is base register + immediate.
// (Round 5, measured and dropped: FOUR weight slots with the fragments of two steps travelling together and a barrier every second step —
// 4 / 8 / 12 barriers per two chunks for K = 3 / 7 / 11 instead of 6 / 14 / 22: 42.84-42.93 ms per forward against 42.96-43.03 with two slots in
// the product build, i.e. inside the noise, and 30-55 % SLOWER in the -DTTSC_ABLATE build of the very same source (profiles/r05_wg_timeline_slots.log) —
// a schedule that fragile is not worth 0.1 ms.)
template <int C, int K, int D>
__global__ __launch_bounds__(256, 2) void conv_f16x3_wide_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int WM = 2, WN = 2;                   // 128 output channels x 256 positions per workgroup (grid.y = C / 128)
    constexpr int MI = 2, NJ = 4, NT = WN * NJ * 32;
    constexpr int SPAN = NT + (K - 1) * D;          // staged positions per channel ("same" padding: halo (K-1)*D)
    constexpr int SPANP = (SPAN + 63) & ~63;
    constexpr int BUFSZ = 4 * SPAN + 2;             // items per activation buffer: 4 planes (h, hi|lo) + a dump slot pair
    constexpr int NCHUNK = C / 16, COTN = C / 32;
    constexpr int AITEMS = WM * MI * 2 * 64;        // weight items of one (tap, chunk) for the workgroup's 128 rows (8 KB)
    half8* Xp = reinterpret_cast<half8*>(smem_raw);   // [2 buffers][plane (h, pl)][SPAN] 16-byte items
    half8* Aw = Xp + 2 * BUFSZ;                       // [2 slots][AITEMS] weight fragments
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int half = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.z;
    const int q0 = blockIdx.x * NT;
    const int lin = a.in_len ? a.in_len[b] : a.Lin;
    if (a.out_len && q0 >= a.out_len[b]) return;
    const int cotg = blockIdx.y * (WM * MI);        // first 32-row tile of the workgroup
#ifdef TTSC_ABLATE
    const unsigned wg_lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
#endif
    TTSC_STAMP(a, wg_lin, 0);
    TTSC_STAMP_HWID(a, wg_lin, 15);

    // Accumulators.  With `acc_init` they START at (bias + residual + running sum) / w_unscale (w_unscale is a power of two: exact), so the
    // epilogue is 128 independent stores per lane.  Why: as epilogue operands the residual tile was fetched four rows at a time, load ->
    // wait -> store, 32 dependent round trips per wave with 1 KB in flight each — the workgroup timeline (tools/wg_timeline.py, round 5) put
    // 41-45 % of a workgroup's life into that epilogue for the K = 3 / K = 7 layers (56 us for 256 KB), and since the workgroups of a launch
    // run in step, the whole chip sat in it together.  Here all of a lane's 128 (256 with the running sum) loads leave back to back at the top
    // of the kernel, in front of the first activation chunk's, and land while the prologue waits for that chunk anyway.
    f32x16 acc[MI][NJ];
    // acc_init: operand sources and this lane's element offsets (32-bit: B * C * L < 2^32 checked by the host)
    const float* ai_first = a.acc_init ? (a.resid ? a.resid : (a.accumulate ? a.y : nullptr)) : nullptr;
    const float* ai_second = (a.acc_init && a.resid && a.accumulate) ? a.y : nullptr;
    const unsigned ai_L = (unsigned)a.Lout;
    const unsigned ai_row0 = ((unsigned)b * a.Cout + (cotg + wm * MI) * 32 + 4 * half) * ai_L;
    const int ai_qw = q0 + wn * (NJ * 32) + l31;
    // row tile i of the accumulators <- first operand (or zero); the loads of a row tile leave back to back
    auto acc_load_first = [&](int i) __attribute__((always_inline)) {
        if (ai_first) {
#pragma unroll
            for (int n = 0; n < NJ; ++n) {
                const int q = ai_qw + n * 32;
                const unsigned o = ai_row0 + (unsigned)(i * 32) * ai_L + (unsigned)(q < a.Lout ? q : 0);   // (columns beyond the row are never stored)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    unsigned og = o + (unsigned)(8 * g) * ai_L;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[i][n][4 * g + e] = ai_first[og];
                        og += ai_L;
                    }
                }
            }
        } else {
#pragma unroll
            for (int n = 0; n < NJ; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][n][r] = 0.f;
        }
    };
    // the rest of the initial value: + running sum (when both operands are present), + bias, / weight scale
    auto acc_init_finish = [&]() __attribute__((always_inline)) {
        const float inv = 1.f / a.w_unscale;
        if (ai_second) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int n0 = 0; n0 < NJ; n0 += 2) {
                    float t[2][16];
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        const int q = ai_qw + (n0 + n) * 32;
                        const unsigned o = ai_row0 + (unsigned)(i * 32) * ai_L + (unsigned)(q < a.Lout ? q : 0);
#pragma unroll
                        for (int r = 0; r < 16; ++r) t[n][r] = ai_second[o + (unsigned)((r & 3) + 8 * (r >> 2)) * ai_L];
                    }
#pragma unroll
                    for (int n = 0; n < 2; ++n)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][n0 + n][r] += t[n][r];
                }
        }
        if (a.bias) {
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                float bv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) bv[r] = a.bias[(cotg + wm * MI + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half];
#pragma unroll
                for (int n = 0; n < NJ; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][n][r] += bv[r];
            }
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int n = 0; n < NJ; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][n][r] *= inv;
    };
    if (a.acc_init) {
        // (filled in the prologue below, between the activation loads: see acc_load_first / acc_init_finish)
    } else if (a.epi_prefetch && a.bias && (a.resid || a.accumulate)) {
        // prefetched epilogue: the bias starts the sum (divided by the weight scale, a power of two: exact), so the epilogue holds no bias registers
        const float inv = 1.f / a.w_unscale;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float bv = a.bias[(cotg + wm * MI + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half] * inv;
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j][r] = bv;
            }
    } else {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }

    const float* xb = a.x + (size_t)b * C * a.Lin;
    const int lo = q0 - D * ((K - 1) / 2);
    const half8* wsrc = reinterpret_cast<const half8*>(a.wph) + (size_t)cotg * 128 + lane;
    // A launch is only 4-6 rounds of workgroups, two resident per CU, all started together: both residents reach their
    // memory-bound epilogue at the same moment and the matrix pipe idles.  The second resident of every CU (dispatch order:
    // workgroups 256..511) therefore starts ~24 us late, so that one partner's epilogue falls into the other's tap loop
    // for the rest of the launch (measured -2 .. -9 % per layer, tools/ablate.py; a wrong guess about the placement only
    // costs the delay).
    if (a.skew) {
        const unsigned lin_id = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        if (lin_id >= 256u && lin_id < 512u)
            for (int z = 0; z < a.skew; ++z) __builtin_amdgcn_s_sleep(127);
    }

    // ---- weights: global -> LDS by LDS-DMA, one (tap, chunk) step ahead; the two waves that share a row tile (and, at
    // 256 channels, nobody else) read them back as ds_read_b128.  Fetched per wave straight into registers (the first
    // version of this kernel) the fragments cost ~30 % of the run time in the CU's vector-memory path.
    auto stage_A = [&](int c, int j, int slot) __attribute__((always_inline)) {
        if (TTSC_DBG(a, 8)) return;
        const half8* wj = wsrc + (size_t)(j * NCHUNK + c) * (COTN * 128);
#pragma unroll
        for (int i = 0; i < AITEMS / 64 / 4; ++i) {
            const int blk = wave + i * 4;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wj + blk * 64),
                                             (__attribute__((address_space(3))) void*)(Aw + slot * AITEMS + blk * 64), 16, 0, 0);
        }
    };

    // ---- activation staging: work item = (channel half h, position p) = 8 fp32 channels -> one (hi, lo) pair of items
    constexpr int XIT = (2 * SPANP + 255) / 256;
    static_assert(XIT <= K, "one staging item per tap");
    float xr[XIT][8];
    unsigned xoff[XIT];
    int xslot[XIT];
    bool xok[XIT];
#pragma unroll
    for (int e = 0; e < XIT; ++e) {
        const int i = tid + e * 256;
        const int h = i >= SPANP ? 1 : 0;
        const int p = i - h * SPANP;
        const int pos = lo + p;
        xok[e] = pos >= 0 && pos < lin;
        int pc = pos > lin - 1 ? lin - 1 : pos;
        pc = pc < 0 ? 0 : pc;
        xoff[e] = (unsigned)(h * 8 * a.Lin + pc);   // channel half folded into the offset
        // lanes beyond the window write to a dump slot behind the planes (no divergent branch around the LDS stores)
        xslot[e] = (p < SPAN && i < 2 * SPANP) ? (h * 2) * SPAN + p : 4 * SPAN;
    }
    auto x_issue_item = [&](int e, int c) __attribute__((always_inline)) {
        const float* rc = xb + (size_t)(c * 16) * a.Lin;
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) xr[e][ch] = rc[(size_t)ch * a.Lin + xoff[e]];
    };
    auto x_commit_item = [&](int e, half8* buf) __attribute__((always_inline)) {
        // (hi, lo) split of the 8 channels, pairwise: cvt_pk + two fma_mix + cvt_pk per pair (conv_internal.hpp::split2_f16; same bits as the
        // scalar convert / convert back / subtract / convert sequence, two instructions fewer per pair)
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        u32x4 uh, ul;
#pragma unroll
        for (int ch = 0; ch < 8; ch += 2) {
            float v0 = xok[e] ? xr[e][ch] * a.in_scale : 0.f, v1 = xok[e] ? xr[e][ch + 1] * a.in_scale : 0.f;
            v0 = fmaxf(v0, v0 * a.in_slope);
            v1 = fmaxf(v1, v1 * a.in_slope);
            unsigned h, l;
            split2_f16(v0, v1, h, l);
            uh[ch >> 1] = h;
            ul[ch >> 1] = l;
        }
        const half8 vh = __builtin_bit_cast(half8, uh), vl = __builtin_bit_cast(half8, ul);
        buf[xslot[e]] = vh;
        buf[xslot[e] + (xslot[e] < 4 * SPAN ? SPAN : 1)] = vl;   // (the dump slot's partner is the item right behind it)
    };

    // all LDS reads of a wave hang off two base registers: buffer, plane, slot and tap offsets are immediates (D, K, NT are
    // template parameters) — per-tap address registers were what pushed the first version of this kernel into scratch
    const half8* xbase = Xp + (unsigned)((half * 2) * SPAN + wn * (NJ * 32) + l31);
    const half8* abase = Aw + (unsigned)(wm * (MI * 128) + lane);
    // one channel chunk: K taps on activation buffer BUF; meanwhile chunk c+1 is converted into the other buffer and the
    // loads of chunk c+2 are issued (item e at tap K-1-e, so that a tap carries at most one item's conversion).  PAR =
    // weight slot of the chunk's first tap (K is odd: it flips from chunk to chunk).  One barrier per tap: it publishes the
    // next tap's weights and, after the last tap, the next chunk's activations.
    auto chunk = [&](int c, auto buf_tag, auto par_tag) __attribute__((always_inline)) {
        constexpr int BUF = decltype(buf_tag)::value;
        constexpr int PAR = decltype(par_tag)::value;
        half8* nxt = Xp + (1 - BUF) * BUFSZ;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const int slot = (PAR + j) & 1;
            // the next step's weights leave first, then (staging taps) the loads of chunk c+2 ...
            if (j + 1 < K)
                stage_A(c, j + 1, slot ^ 1);
            else
                stage_A(c + 1 < NCHUNK ? c + 1 : c, 0, slot ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            half8 ah[MI], al[MI], bh[NJ], bl[NJ];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                ah[i] = abase[slot * AITEMS + i * 128];
                al[i] = abase[slot * AITEMS + i * 128 + 64];
            }
#pragma unroll
            for (int n = 0; n < NJ; ++n) {
                bh[n] = xbase[BUF * BUFSZ + j * D + n * 32];
                bl[n] = xbase[BUF * BUFSZ + SPAN + j * D + n * 32];
            }
            // ... then, on the staging taps (the last XIT taps of a chunk, one item each), chunk c+1 is converted and
            // published and the staging registers are refilled with chunk c+2 — the compiler interleaves this VALU work
            // with the tap's MFMAs (past the last chunk it handles clamped garbage nobody reads: no branches)
            const int e = K - 1 - j;
            if (e < XIT && !TTSC_DBG(a, 1)) {
                x_commit_item(e, nxt);
                x_issue_item(e, c + 2 < NCHUNK ? c + 2 : NCHUNK - 1);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int n = 0; n < NJ; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[n], acc[i][n], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int n = 0; n < NJ; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[n], acc[i][n], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int n = 0; n < NJ; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[n], acc[i][n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (!TTSC_DBG(a, 2)) __syncthreads();
        }
    };

    // prologue: chunk 0 -> buffer 0, chunk 1 into the staging registers, first weight step
    stage_A(0, 0, 0);
#pragma unroll
    for (int e = 0; e < XIT; ++e) x_issue_item(e, 0);
    if (a.acc_init) {
        // the first row tile's operand loads leave BEHIND the first chunk's (loads return in order: the chunk's conversion below then waits for
        // its own loads only), the second row tile's behind the second chunk's; the arithmetic on them comes last
        __builtin_amdgcn_sched_barrier(0);
        acc_load_first(0);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int e = 0; e < XIT; ++e) x_commit_item(e, Xp);
#pragma unroll
    for (int e = 0; e < XIT; ++e) x_issue_item(e, 1);
    if (a.acc_init) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 1; i < MI; ++i) acc_load_first(i);
        acc_init_finish();
    }
    __syncthreads();
    TTSC_STAMP(a, wg_lin, 1);
    for (int c = 0; c < NCHUNK; c += 2) {
        chunk(c, std::integral_constant<int, 0>(), std::integral_constant<int, 0>());
        chunk(c + 1, std::integral_constant<int, 1>(), std::integral_constant<int, (K & 1)>());
    }
    TTSC_STAMP(a, wg_lin, 2);

    if (TTSC_DBG(a, 4)) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int n = 0; n < NJ; ++n) t += acc[i][n][0] + acc[i][n][7];
        if (t == 12345.678f) a.y[0] = 1.f;   // keep the accumulators alive
        return;
    }
    if (a.acc_init) {
        // everything but the weight scale is already in the sum: independent stores, nothing to wait for
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int n = 0; n < NJ; ++n) {
                const int q = q0 + wn * (NJ * 32) + n * 32 + l31;
                if (q < a.Lout) {
                    float* yp = a.y + ((size_t)b * a.Cout + (cotg + wm * MI + i) * 32 + 4 * half) * a.Lout + q;
#pragma unroll
                    for (int r = 0; r < 16; ++r) yp[(size_t)((r & 3) + 8 * (r >> 2)) * a.Lout] = acc[i][n][r] * a.w_unscale;
                }
            }
    } else if (a.epi_prefetch && (a.resid || a.accumulate)) {
        // Deep-prefetched epilogue.  Same arithmetic as epilogue_tile EXCEPT that the bias already sits in the accumulators (it started the sum, see their
        // initialisation: not bit-identical to the general kernel, which adds the bias after the sum — ADVICE r5): the residual (and running-sum) operands of the NEXT tiles
        // are in flight while a tile is finished and stored — four tiles ahead with one operand, two with both (64 registers: the fragment and
        // staging registers of the main loop are free now).  epilogue_tile fetched four rows at a time, load -> wait -> store: 32 dependent
        // round trips per wave with 1 KB in flight each; the workgroup timeline (tools/wg_timeline.py, round 5) put 41-45 % of a workgroup's
        // life into that for the K = 3 / K = 7 layers (56 us for 256 KB), and the workgroups of a launch run in step, so the chip sat in it together.
        const float* rsrc = a.resid;
        const float* ysrc = a.accumulate ? a.y : nullptr;
        constexpr int NTILE = MI * NJ;
        // 32-bit element offsets from the tensor base (B * C * L < 2^32 is checked by the host); the 16 rows of a tile follow from its first by adds
        const unsigned L1 = (unsigned)a.Lout;
        const unsigned row0 = ((unsigned)b * a.Cout + (cotg + wm * MI) * 32 + 4 * half) * L1;
        const int qw = q0 + wn * (NJ * 32) + l31;
        auto tile_off = [&](int t) __attribute__((always_inline)) -> unsigned {
            const int i = t / NJ, n = t % NJ;
            const int q = qw + n * 32;
            return row0 + (unsigned)(i * 32) * L1 + (unsigned)(q < a.Lout ? q : 0);
        };
        // MODE 0: residual only, 1: running sum only, 2: both (compile-time: no per-element selects)
        auto body = [&](auto mode_tag) __attribute__((always_inline)) {
            constexpr int MODE = decltype(mode_tag)::value;
            constexpr bool BOTH = MODE == 2;
            constexpr int DEPTH = BOTH ? 1 : 3;   // (register budget: 128 accumulators + 48 / 32 operand registers)
            const float* s0 = MODE == 1 ? ysrc : rsrc;
            float t0[DEPTH][16], t1[BOTH ? DEPTH : 1][16];
            auto issue = [&](int t) __attribute__((always_inline)) {
                unsigned o = tile_off(t);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    unsigned og = o + (unsigned)(8 * g) * L1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        t0[t % DEPTH][4 * g + e] = s0[og];
                        if constexpr (BOTH) t1[t % DEPTH][4 * g + e] = ysrc[og];
                        og += L1;
                    }
                }
            };
            // (the bias is already in the accumulators: see their initialisation)
#pragma unroll
            for (int t = 0; t < DEPTH; ++t) issue(t);
#pragma unroll
            for (int t = 0; t < NTILE; ++t) {
                const int i = t / NJ, n = t % NJ;
                float res[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float u = acc[i][n][r] * a.w_unscale;
                    if constexpr (MODE == 0) res[r] = (u + t0[t % DEPTH][r]) * a.out_scale + 0.f;
                    if constexpr (MODE == 1) res[r] = (u + 0.f) * a.out_scale + t0[t % DEPTH][r];
                    if constexpr (MODE == 2) res[r] = (u + t0[t % DEPTH][r]) * a.out_scale + t1[t % DEPTH][r];
                }
                if (qw + n * 32 < a.Lout) {
                    unsigned o = tile_off(t);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        unsigned og = o + (unsigned)(8 * g) * L1;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            a.y[og] = res[4 * g + e];
                            og += L1;
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);   // (the next tile's loads stay behind this tile's stores)
                if (t + DEPTH < NTILE) issue(t + DEPTH);
            }
        };
        if (rsrc && ysrc)
            body(std::integral_constant<int, 2>());
        else if (rsrc)
            body(std::integral_constant<int, 0>());
        else
            body(std::integral_constant<int, 1>());
    } else {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int n = 0; n < NJ; ++n) {
                const int q = q0 + wn * (NJ * 32) + n * 32 + l31;
                epilogue_tile(acc[i][n], a, b, (cotg + wm * MI + i) * 32, q, q < a.Lout, half, a.w_unscale);
            }
        }
    }
#ifdef TTSC_ABLATE
    if (a.prof) {
        __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0): the stores have been accepted
        TTSC_STAMP(a, wg_lin, 3);
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// Fused residual pair for 32-channel stages (HiFi-GAN ResBlock1, last upsample stage):
//     y = x + conv2_{d=1}( lrelu( conv1_{d}( lrelu(x) ) ) )          [ + running sum ]
// The 32-channel stage is HBM-bound when its two convolutions run as separate launches (24..88 FLOP per byte moved);
// here the inner activation never leaves the CU: conv1 is evaluated on a 512-column grid (480 outputs + 16 columns of
// margin on each side, >= conv2's halo), its result is leaky-relu'ed, split into fp16 hi/lo and parked in LDS as
// [position][32 ch], and conv2 runs straight out of LDS.  Per pair the HBM traffic drops from 5 tensor passes to ~3.2.
// 8 waves side by side along time (2 MFMA column tiles each); with only 32 output channels the weight fragments of a
// 16-channel chunk fit in registers (K taps x hi/lo x 4 VGPRs), so the tap loop has no barriers and reads only the
// activation fragments from LDS.  K (3/7/11) is a template parameter so that the register-resident fragments are
// statically indexed.
struct PairArgs {
    const float* x;      // [B, 32, L]   input AND residual
    float* y;            // [B, 32, L]   must not alias x (neighbouring tiles read x's halo)
    const void* w1;      // f16x3 fragments of conv1 / conv2: [tap][2 chunks][1][2][64][8 half]
    const void* w2;
    const float* b1;
    const float* b2;
    const int* len;      // [B] valid length or null
    float unscale1, unscale2;
    int L, d1, accumulate;
};

template <int K, int NW>
__global__ __launch_bounds__(64 * NW, (NW == 8 ? 2 : 2)) void respair32_f16x3_kernel(PairArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int C = 32, NCOL = 64 * NW, MARG = 16, NTO = NCOL - 2 * MARG, XTP = NCOL + 16, H2 = (K - 1) / 2, NTHR = 64 * NW;
    const int span1 = NCOL + (K - 1) * a.d1;
    // plane layouts as in conv_f16x3_kernel: X planes (h, pl) of span1 items; T planes (chunk c, h, pl) of XTP items
    half8* Xp = reinterpret_cast<half8*>(smem_raw);
    half8* Tp = Xp + (size_t)4 * span1;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.y;
    const int q0 = blockIdx.x * NTO;
    const int lin = a.len ? a.len[b] : a.L;
    if (q0 >= lin) return;
    const int h1 = a.d1 * (K - 1) / 2;
    const int lo = q0 - MARG - h1;  // x position of X-LDS column 0
    const float* xb = a.x + (size_t)b * C * a.L;

    f32x16 acc[2];
    half8 ah[K], al[K];

    constexpr int XIT = ((NCOL + 64) * 2 + NTHR - 1) / NTHR;
    const int spanp = (span1 + 63) & ~63;
    float xr[XIT][8];
    auto x_issue = [&](int c) {
#pragma unroll
        for (int e = 0; e < XIT; ++e) {
            const int i = tid + e * NTHR;
            const int h = i >= spanp ? 1 : 0;
            const int p = i - h * spanp;
            int pos = lo + p;
            pos = pos > lin - 1 ? lin - 1 : pos;
            pos = pos < 0 ? 0 : pos;
            const int cb = c * 16 + h * 8;
#pragma unroll
            for (int ch = 0; ch < 8; ++ch) xr[e][ch] = xb[(size_t)(cb + ch) * a.L + pos];
        }
    };
    auto x_commit = [&]() {
#pragma unroll
        for (int e = 0; e < XIT; ++e) {
            const int i = tid + e * NTHR;
            const int h = i >= spanp ? 1 : 0;
            const int p = i - h * spanp;
            const int pos = lo + p;
            const bool pok = pos >= 0 && pos < lin;
            if (p < span1 && i < 2 * spanp) {
                half8 vh, vl;
#pragma unroll
                for (int ch = 0; ch < 8; ++ch) {
                    float v = pok ? xr[e][ch] : 0.f;
                    v = fmaxf(v, v * 0.1f);
                    const _Float16 hh = (_Float16)v;
                    vh[ch] = hh;
                    vl[ch] = (_Float16)(v - (float)hh);
                }
                Xp[(size_t)(h * 2 + 0) * span1 + p] = vh;
                Xp[(size_t)(h * 2 + 1) * span1 + p] = vl;
            }
        }
    };
    auto load_a = [&](const void* w, int c) {
        const half8* src = reinterpret_cast<const half8*>(w);
#pragma unroll
        for (int j = 0; j < K; ++j) {
            ah[j] = src[(size_t)(j * 2 + c) * 128 + lane];
            al[j] = src[(size_t)(j * 2 + c) * 128 + 64 + lane];
        }
    };

    // ---------------- conv1 (dilated) over the 512-column grid ----------------
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    x_issue(0);
    for (int c = 0; c < 2; ++c) {
        load_a(a.w1, c);
        if (c) __syncthreads();  // everyone finished reading chunk 0 of the activation tile
        x_commit();
        __syncthreads();
        if (c == 0) x_issue(1);
        const half8* xh = Xp + (unsigned)((half * 2 + 0) * span1 + wv * 64 + l31);
        const half8* xl = Xp + (unsigned)((half * 2 + 1) * span1 + wv * 64 + l31);
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const half8 bh0 = xh[j * a.d1], bh1 = xh[j * a.d1 + 32];
            const half8 bl0 = xl[j * a.d1], bl1 = xl[j * a.d1 + 32];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[j], bh0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[j], bh1, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[j], bl0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[j], bl1, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[j], bh0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[j], bh1, acc[1], 0, 0, 0);
        }
    }
    // conv1 epilogue -> LDS: xt = lrelu(conv1 + b1), zero outside the sequence (conv2 pads with zeros), hi/lo split.
    // A lane holds channels {4*half + (r&3) + 8*(r>>2)} of its column: four groups of 4 consecutive channels.
    load_a(a.w2, 0);  // conv2's first weight chunk travels while the epilogue runs
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int col = wv * 64 + n * 32 + l31;
        const int pos = q0 - MARG + col;
        const bool pok = pos >= 0 && pos < lin;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            typedef _Float16 half4 __attribute__((ext_vector_type(4)));
            half4 vh, vl;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ch = 8 * g + 4 * half + e;
                float v = acc[n][4 * g + e] * a.unscale1 + a.b1[ch];
                v = fmaxf(v, v * 0.1f);
                v = pok ? v : 0.f;
                const _Float16 hh = (_Float16)v;
                vh[e] = hh;
                vl[e] = (_Float16)(v - (float)hh);
            }
            // channel group g (8 channels) = chunk g/2, channel-half g%2; this lane owns 4 of its 8 channels
            _Float16* th = reinterpret_cast<_Float16*>(Tp + (size_t)(g * 2 + 0) * XTP + (col + 8)) + 4 * half;
            _Float16* tl = reinterpret_cast<_Float16*>(Tp + (size_t)(g * 2 + 1) * XTP + (col + 8)) + 4 * half;
            *reinterpret_cast<half4*>(th) = vh;
            *reinterpret_cast<half4*>(tl) = vl;
        }
    }
    __syncthreads();
    // ---------------- conv2 (dilation 1) straight out of LDS ----------------
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    for (int c = 0; c < 2; ++c) {
        if (c) load_a(a.w2, 1);
        const half8* th = Tp + (unsigned)(((c * 2 + half) * 2 + 0) * XTP + wv * 64 + l31 + 8 - H2);
        const half8* tl = Tp + (unsigned)(((c * 2 + half) * 2 + 1) * XTP + wv * 64 + l31 + 8 - H2);
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const half8 bh0 = th[j], bh1 = th[j + 32];
            const half8 bl0 = tl[j], bl1 = tl[j + 32];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[j], bh0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[j], bh1, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[j], bl0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[j], bl1, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[j], bh0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[j], bh1, acc[1], 0, 0, 0);
        }
    }
    // conv2 epilogue: + b2 + x (residual) [+ running sum]; only the 480 central columns are outputs
    ConvArgs ea;
    ea.y = a.y;
    ea.resid = a.x;
    ea.bias = a.b2;
    ea.Cout = C;
    ea.Lout = a.L;
    ea.out_scale = 1.f;
    ea.out_act = TTSC_ACT_NONE;
    ea.accumulate = a.accumulate;
    ea.dbg = 0;
    ea.skew = 0;
    ea.acc_init = 0;
    ea.epi_prefetch = 0;
    ea.swz_nx = ea.swz_ny = 0;
    ea.gate = nullptr;
    ea.gate_slope = 1.f;
    ea.vphase = 0;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int col = wv * 64 + n * 32 + l31;
        const int q = q0 + col - MARG;
        const bool qok = col >= MARG && col < MARG + NTO && q < lin;
        epilogue_tile(acc[n], ea, b, 0, q, qok, half, a.unscale2);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Single-output-channel Conv1d (HiFi-GAN conv_post: 32 -> 1, K = 7, tanh): an M = 1 problem wastes 31/32 of an MFMA
// tile and its time goes into staging, so it runs on the vector ALU as an fp32 fmaf chain (ci-major, tap-minor) and is
// bound by the one HBM read of its input.  1024 outputs per workgroup; 8-channel chunks of the activated input window
// go through LDS (coalesced dword loads in, conflict-free ds_read_b128 out); a thread owns 4 consecutive outputs.
__global__ __launch_bounds__(256) void conv_cout1_kernel(ConvArgs a) {
    constexpr int NT = 1024, CH = 8, HALO = 16;          // HALO >= (K - 1) * dilation, K <= 16, dilation 1
    __shared__ __attribute__((aligned(16))) float xs[CH][NT + 2 * HALO];
    __shared__ float ws[64 * 16];
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, b = blockIdx.y;
    const int q0 = blockIdx.x * NT;
    const int lin = a.in_len ? a.in_len[b] : a.Lin;
    if (a.out_len && q0 >= a.out_len[b]) return;
    const int K = a.ntaps;
    for (int i = tid; i < a.Cin * K; i += 256) ws[i] = a.wp[i];     // plain [Cin][K] fp32 weights (see conv_repack)
    const float* xb = a.x + (size_t)b * a.Cin * a.Lin;
    const int lo = q0 + a.tap_base;                                  // x position of LDS column 0
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c0 = 0; c0 < a.Cin; c0 += CH) {
        __syncthreads();
        // stage 8 channel rows: every thread issues 8 x 5 independent, clamped, unconditional loads (no integer division, no
        // branch around a load — the first version of this loop ran one dependent load at a time), then activates and stores
        {
            constexpr int PER = (NT + 2 * HALO + 255) / 256;   // 5 columns per thread and row
            float v[CH][PER];
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int ci = c0 + c < a.Cin ? c0 + c : a.Cin - 1;
                const float* row = xb + (size_t)ci * a.Lin;
#pragma unroll
                for (int k = 0; k < PER; ++k) {
                    int pos = lo + tid + k * 256;
                    pos = pos < 0 ? 0 : (pos > a.Lin - 1 ? a.Lin - 1 : pos);
                    v[c][k] = row[pos];
                }
            }
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
                for (int k = 0; k < PER; ++k) {
                    const int p = tid + k * 256, pos = lo + p;
                    float t = v[c][k] * a.in_scale;
                    t = fmaxf(t, t * a.in_slope);
                    if (p < NT + 2 * HALO) xs[c][p] = (c0 + c < a.Cin && pos >= 0 && pos < lin) ? t : 0.f;
                }
        }
        __syncthreads();
        const int nc = a.Cin - c0 < CH ? a.Cin - c0 : CH;
        for (int c = 0; c < nc; ++c) {
            // outputs 4*tid .. 4*tid+3 read window columns 4*tid .. 4*tid + 3 + K - 1  (<= 4*tid + 18)
            float w[20];
#pragma unroll
            for (int v4 = 0; v4 < 5; ++v4) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(&xs[c][4 * tid + 4 * v4]);
                w[4 * v4] = t[0]; w[4 * v4 + 1] = t[1]; w[4 * v4 + 2] = t[2]; w[4 * v4 + 3] = t[3];
            }
            const float* wk = ws + (c0 + c) * K;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (j < K) {
                    const float wj = wk[j];
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] = fmaf(wj, w[e + j], acc[e]);
                }
            }
        }
    }
    const float bv = a.bias ? a.bias[0] : 0.f;
    float* yb = a.y + (size_t)b * a.Lout;
    const float* rb = a.resid ? a.resid + (size_t)b * a.Lout : nullptr;
    const int q = q0 + 4 * tid;
    float res[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const bool ok = q + e < a.Lout;
        float t = (acc[e] + bv + (rb && ok ? rb[q + e] : 0.f)) * a.out_scale;
        t = apply_act(t, a.out_act);
        res[e] = t + (a.accumulate && ok ? yb[q + e] : 0.f);
    }
    if (a.nf_flag) {
        // range guard of the split-precision generator: an fp16 overflow anywhere upstream reaches the waveform as NaN (the hi and
        // lo halves of an overflowed value are +inf and -inf, their products cancel to NaN), so one test per output sample on this
        // HBM-bound kernel covers every layer (hifigan.cpp re-calibrates and reruns when the word is set)
        bool bad = false;
#pragma unroll
        for (int e = 0; e < 4; ++e) bad = bad || (q + e < a.Lout && !(fabsf(res[e]) <= 3.0e38f));
        if (bad) atomicOr(a.nf_flag, 1u);
    }
    if (q + 3 < a.Lout && (((uintptr_t)(yb + q)) & 15) == 0) {
        const f32x4 o = {res[0], res[1], res[2], res[3]};
        *reinterpret_cast<f32x4*>(yb + q) = o;
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (q + e < a.Lout) yb[q + e] = res[e];
    }
}

template <int MI, int NJ, int TMAX>
static int launch_f16_t(const ConvArgs& a, int B, hipStream_t s) {
    constexpr int NT = 4 * NJ * 32;
    constexpr int MT = MI * 32;
    dim3 grid((unsigned)ceil_div(a.q_cnt, NT), (unsigned)(a.CoutP / MT), (unsigned)B);
    const size_t lds = (size_t)a.span_pad * 4 * 16 + (size_t)a.ntaps * MI * 2 * 64 * 16;
    if (int rc = ensure_full_lds((const void*)conv_f16x3_kernel<MI, NJ, TMAX>)) return rc;   // once per (device, kernel)
    hipLaunchKernelGGL((conv_f16x3_kernel<MI, NJ, TMAX>), grid, dim3(256), lds, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("conv_f16x3_kernel launch failed: %s", hipGetErrorString(e));
        return TTSC_EHIP;
    }
    return TTSC_OK;
}

template <int C, int K, int D>
static int launch_f16_wide(const ConvArgs& a0, int B, hipStream_t s) {
    constexpr int NT = 256;
    constexpr int SPAN = NT + (K - 1) * D;
    dim3 grid((unsigned)ceil_div(a0.Lout, NT), (unsigned)(C / 128), (unsigned)B);
    ConvArgs a = a0;
    // start skew of the second resident workgroup per CU (see the kernel): only when the launch has several full rounds
    static const int skew_env = getenv("TTSC_CONV_SKEW") ? atoi(getenv("TTSC_CONV_SKEW")) : 0;   // (round 5: off — with the short epilogue the delay no longer pays: 42.87 vs 43.0 ms per forward)
    a.skew = ((size_t)grid.x * grid.y * grid.z >= 1024) ? skew_env : 0;
    // epilogue operands as the accumulators' initial value (see the kernel) whenever the epilogue is the plain affine one; TTSC_CONV_ACC_INIT=0
    // keeps the operand loads in the epilogue (measurement switch)
    // (a.acc_init — the operands as the accumulators' initial value, see the kernel — is decided by the caller for the LAYER, not per kernel: the
    // general kernel evaluates the same layer with the same arithmetic when the machine-fill rule sends it there)
    // deep-prefetched epilogue operands (bit-identical to the plain epilogue; TTSC_CONV_EPI_PREFETCH=0 restores the four-rows-at-a-time one)
    static const int epi_env = getenv("TTSC_CONV_EPI_PREFETCH") ? atoi(getenv("TTSC_CONV_EPI_PREFETCH")) : 1;
    a.epi_prefetch = (epi_env && !a.gate && a.out_act == TTSC_ACT_NONE && a.Cout == C && (size_t)B * C * a.Lout < (1ull << 32)) ? 1 : 0;   // (32-bit element offsets)
    constexpr size_t lds = (size_t)2 * (4 * SPAN + 2) * 16 + (size_t)2 * (2 * 2 * 2 * 64) * 16;   // activations + 2 weight slots
    if (int rc = ensure_full_lds((const void*)conv_f16x3_wide_kernel<C, K, D>)) return rc;   // once per (device, kernel)
    hipLaunchKernelGGL((conv_f16x3_wide_kernel<C, K, D>), grid, dim3(256), lds, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("conv_f16x3_wide_kernel launch failed: %s", hipGetErrorString(e));
        return TTSC_EHIP;
    }
    return TTSC_OK;
}


// ---------------------------------------------------------------------------------------------------------------
// Tall-tile variant for the first two upsamplers (ConvTranspose1d 512 -> 256, k16 s5 and 256 -> 128, k16 s3).  As GEMMs they are 1280 / 768
// "virtual rows" (phase, channel) x CIN * J deep over the INPUT positions, J = ceil(K / stride) taps reading x[q - j].  The general kernel
// runs them as 20 / 12 M tiles of 64 rows that each load, leaky-relu and split the same input window: the ablation (tools/ablate.py --ups)
// puts 45 % of their time into that staging.  Here a workgroup owns 256 rows x 128 input positions (2 x 2 waves of 4 x 2 MFMA tiles: the
// same 24 MFMAs per 12 fragment reads as the wide kernel), so the window is staged 5 / 3 times instead of 20 / 12, on the wide kernel's
// skeleton: weights by LDS-DMA one tap ahead, activations double-buffered, conversion of chunk c+1 spread over the last taps of chunk c.
// (MI, NJ) = (4, 2): 256 rows x 128 positions (ups.0: 1280 rows); (2, 4): 128 rows x 256 positions (ups.1: 384 rows = 3 tiles instead of 6).
template <int CIN, int J, int MI, int NJ>
__global__ __launch_bounds__(256, 2) void conv_f16x3_tall_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int WM = 2, WN = 2, NT = WN * NJ * 32;
    constexpr int SPAN = NT + (J - 1);
    constexpr int SPANP = (SPAN + 63) & ~63;
    constexpr int BUFSZ = 4 * SPAN + 2;
    constexpr int NCHUNK = CIN / 16;
    constexpr int AITEMS = WM * MI * 2 * 64;        // weight items of one (tap, chunk) for the workgroup's 256 rows (16 KB)
    static_assert(J % 2 == 0 && NCHUNK % 2 == 0, "slot parity: an even number of taps per chunk, chunks in pairs");
    half8* Xp = reinterpret_cast<half8*>(smem_raw);   // [2 buffers][plane (h, pl)][SPAN]
    half8* Aw = Xp + 2 * BUFSZ;                       // [2 slots][AITEMS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int half = lane >> 5, l31 = lane & 31;
    // Which tile?  The row tiles of a transposed convolution are its PHASES: they write interleaved samples of the same output rows, every
    // 128-byte line of the output gets a fifth (stride 5) / a third (stride 3) of its bytes from each of them.  Dispatched as a 3-D grid the
    // phases of a q tile land on different XCDs — workgroup i runs on XCD i mod 8 (tools/probes/xcc_probe.hip) — whose L2s cannot merge the
    // partial lines: the 512 -> 256, k16 s5 upsampler wrote 1311 MB for a 262 MB tensor (round-4 PMC).  As a 1-D launch the id is decoded so
    // that the phases of one q tile are consecutive workgroups of ONE XCD: its L2 sees all pieces of a line within a few microseconds.
    int bx, by, bz;
    if (a.swz_nx > 0) {
        const unsigned id = blockIdx.x, xcd = id & 7u, u = id >> 3;
        by = (int)(u % (unsigned)a.swz_ny);
        const unsigned t = (u / (unsigned)a.swz_ny) * 8u + xcd;      // q tile x utterance, linear
        bx = (int)(t % (unsigned)a.swz_nx);
        bz = (int)(t / (unsigned)a.swz_nx);
        if (bz >= a.fold_B) return;   // (padding of the launch to a multiple of 8 tiles; fold_B carries the batch size here)
    } else {
        bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    }
    const int b = bz;
    const int q0 = a.q_lo + bx * NT;          // first INPUT position of the tile
    const int lin = a.in_len ? a.in_len[b] : a.Lin;
    if (a.out_len && (long)q0 * a.out_stride + a.out_off >= a.out_len[b]) return;
    const int cotg = by * (WM * MI);
    const int cotN = a.CoutP >> 5;

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const float* xb = a.x + (size_t)b * CIN * a.Lin;
    const int lo = q0 - (J - 1);                      // x position of LDS column 0 (tap j reads column (q - q0) + J - 1 - j)
    const half8* wsrc = reinterpret_cast<const half8*>(a.wph) + (size_t)cotg * 128 + lane;
    auto stage_A = [&](int c, int j, int slot) __attribute__((always_inline)) {
        const half8* wj = wsrc + (size_t)(j * NCHUNK + c) * ((size_t)cotN * 128);
#pragma unroll
        for (int i = 0; i < AITEMS / 64 / 4; ++i) {
            const int blk = wave + i * 4;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wj + blk * 64),
                                             (__attribute__((address_space(3))) void*)(Aw + slot * AITEMS + blk * 64), 16, 0, 0);
        }
    };
    constexpr int XIT = (2 * SPANP + 255) / 256;
    static_assert(XIT <= J, "one staging item per tap");
    float xr[XIT][8];
    unsigned xoff[XIT];
    int xslot[XIT];
    bool xok[XIT];
#pragma unroll
    for (int e = 0; e < XIT; ++e) {
        const int i = tid + e * 256;
        const int h = i >= SPANP ? 1 : 0;
        const int p = i - h * SPANP;
        const int pos = lo + p;
        xok[e] = pos >= 0 && pos < lin;
        int pc = pos > lin - 1 ? lin - 1 : pos;
        pc = pc < 0 ? 0 : pc;
        xoff[e] = (unsigned)(h * 8 * a.Lin + pc);
        xslot[e] = (p < SPAN && i < 2 * SPANP) ? (h * 2) * SPAN + p : 4 * SPAN;
    }
    auto x_issue_item = [&](int e, int c) __attribute__((always_inline)) {
        const float* rc = xb + (size_t)(c * 16) * a.Lin;
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) xr[e][ch] = rc[(size_t)ch * a.Lin + xoff[e]];
    };
    auto x_commit_item = [&](int e, half8* buf) __attribute__((always_inline)) {
        // (hi, lo) split of the 8 channels, pairwise: cvt_pk + two fma_mix + cvt_pk per pair (conv_internal.hpp::split2_f16; same bits as the
        // scalar convert / convert back / subtract / convert sequence, two instructions fewer per pair)
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        u32x4 uh, ul;
#pragma unroll
        for (int ch = 0; ch < 8; ch += 2) {
            float v0 = xok[e] ? xr[e][ch] * a.in_scale : 0.f, v1 = xok[e] ? xr[e][ch + 1] * a.in_scale : 0.f;
            v0 = fmaxf(v0, v0 * a.in_slope);
            v1 = fmaxf(v1, v1 * a.in_slope);
            unsigned h, l;
            split2_f16(v0, v1, h, l);
            uh[ch >> 1] = h;
            ul[ch >> 1] = l;
        }
        const half8 vh = __builtin_bit_cast(half8, uh), vl = __builtin_bit_cast(half8, ul);
        buf[xslot[e]] = vh;
        buf[xslot[e] + (xslot[e] < 4 * SPAN ? SPAN : 1)] = vl;
    };
    const half8* xbase = Xp + (unsigned)((half * 2) * SPAN + wn * (NJ * 32) + l31);
    const half8* abase = Aw + (unsigned)(wm * (MI * 128) + lane);
    auto chunk = [&](int c, auto buf_tag) __attribute__((always_inline)) {
        constexpr int BUF = decltype(buf_tag)::value;
        half8* nxt = Xp + (1 - BUF) * BUFSZ;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int slot = j & 1;   // (J is even: every chunk starts on slot 0)
            if (j + 1 < J)
                stage_A(c, j + 1, slot ^ 1);
            else
                stage_A(c + 1 < NCHUNK ? c + 1 : c, 0, slot ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            half8 ah[MI], al[MI], bh[NJ], bl[NJ];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                ah[i] = abase[slot * AITEMS + i * 128];
                al[i] = abase[slot * AITEMS + i * 128 + 64];
            }
#pragma unroll
            for (int n = 0; n < NJ; ++n) {
                bh[n] = xbase[BUF * BUFSZ + (J - 1 - j) + n * 32];
                bl[n] = xbase[BUF * BUFSZ + SPAN + (J - 1 - j) + n * 32];
            }
            const int e = J - 1 - j;
            if (e < XIT) {
                x_commit_item(e, nxt);
                x_issue_item(e, c + 2 < NCHUNK ? c + 2 : NCHUNK - 1);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int n = 0; n < NJ; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[n], acc[i][n], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int n = 0; n < NJ; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[n], acc[i][n], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int n = 0; n < NJ; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[n], acc[i][n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
        }
    };
    stage_A(0, 0, 0);
#pragma unroll
    for (int e = 0; e < XIT; ++e) x_issue_item(e, 0);
#pragma unroll
    for (int e = 0; e < XIT; ++e) x_commit_item(e, Xp);
#pragma unroll
    for (int e = 0; e < XIT; ++e) x_issue_item(e, 1);
    __syncthreads();
    for (int c = 0; c < NCHUNK; c += 2) {
        chunk(c, std::integral_constant<int, 0>());
        chunk(c + 1, std::integral_constant<int, 1>());
    }
    // epilogue: virtual row tile -> (phase r, real channel tile); output position o = q * stride + r - padding
    const int q_hi = a.q_lo + a.q_cnt;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int n = 0; n < NJ; ++n) {
            const int q = q0 + wn * (NJ * 32) + n * 32 + l31;
            int cb = (cotg + wm * MI + i) * 32;
            const int r = cb / a.vphase;
            cb -= r * a.vphase;
            const long oo = (long)q * a.out_stride + a.out_off + r;
            const bool ok = (q < q_hi) && (oo >= 0) && (oo < a.Lout) && (r < a.out_stride);
            epilogue_tile(acc[i][n], a, b, cb, oo, ok, half, a.w_unscale);
        }
    }
}

template <int CIN, int J, int MI, int NJ>
static int launch_f16_tall(const ConvArgs& a0, int B, hipStream_t s) {
    constexpr int NT = 2 * NJ * 32, SPAN = NT + (J - 1), MT = 2 * MI * 32;
    ConvArgs a = a0;
    dim3 grid((unsigned)ceil_div(a.q_cnt, NT), (unsigned)(a.CoutP / MT), (unsigned)B);
    static const int swz_env = getenv("TTSC_TALL_SWIZZLE") ? atoi(getenv("TTSC_TALL_SWIZZLE")) : 1;
    if (swz_env && grid.y > 1) {   // XCD-aware 1-D launch (see the kernel): the row tiles of a q tile on one XCD, back to back
        a.swz_nx = (int)grid.x;
        a.swz_ny = (int)grid.y;
        a.fold_B = B;
        const unsigned tiles = (unsigned)round_up((int64_t)grid.x * B, 8);
        grid = dim3(tiles * grid.y, 1, 1);
    }
    constexpr size_t lds = (size_t)2 * (4 * SPAN + 2) * 16 + (size_t)2 * (2 * MI * 2 * 64) * 16;
    if (int rc = ensure_full_lds((const void*)conv_f16x3_tall_kernel<CIN, J, MI, NJ>)) return rc;
    hipLaunchKernelGGL((conv_f16x3_tall_kernel<CIN, J, MI, NJ>), grid, dim3(256), lds, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("conv_f16x3_tall_kernel launch failed: %s", hipGetErrorString(e));
        return TTSC_EHIP;
    }
    return TTSC_OK;
}

template <int C, int K>
static int launch_f16_wide_d(const ConvArgs& a, int B, int d, hipStream_t s) {
    if (d == 1) return launch_f16_wide<C, K, 1>(a, B, s);
    if (d == 3) return launch_f16_wide<C, K, 3>(a, B, s);
    return launch_f16_wide<C, K, 5>(a, B, s);
}

template <int C>
static int launch_f16_wide_k(const ConvArgs& a, int B, int d, hipStream_t s) {
    if (a.ntaps == 3) return launch_f16_wide_d<C, 3>(a, B, d, s);
#ifdef TTSC_PROBE_EVENK
    if (a.ntaps == 4) return launch_f16_wide<C, 4, 1>(a, B, s);
    if (a.ntaps == 6) return launch_f16_wide<C, 6, 1>(a, B, s);
#endif
    if (a.ntaps == 7) return launch_f16_wide_d<C, 7>(a, B, d, s);
    return launch_f16_wide_d<C, 11>(a, B, d, s);
}

template <int MI, int NJ>
static int launch_f16(const ConvArgs& a, int B, hipStream_t s) {
    if (a.ntaps <= 3) return launch_f16_t<MI, NJ, 3>(a, B, s);
    if (a.ntaps <= 7) return launch_f16_t<MI, NJ, 7>(a, B, s);
    if (a.ntaps <= 11) return launch_f16_t<MI, NJ, 11>(a, B, s);
    return launch_f16_t<MI, NJ, 16>(a, B, s);
}

template <int MI, int NJ, int WM, int WN>
static int launch_cfg(const ConvArgs& a, int B, hipStream_t s) {
    constexpr int NT = WN * NJ * 32;
    constexpr int MT = WM * MI * 32;
    dim3 grid((unsigned)ceil_div(a.q_cnt, NT), (unsigned)(a.CoutP / MT), (unsigned)B);
    dim3 block(WM * WN * 64);
    size_t lds = (size_t)2 * KC * a.span_pad * sizeof(float);
    TTSC_REQUIRE(lds <= 160 * 1024, "conv_mfma_kernel: receptive field too large for LDS (%zu bytes)", lds);
    if (lds > 64 * 1024)
        if (int rc = ensure_full_lds(reinterpret_cast<const void*>(conv_mfma_kernel<MI, NJ, WM, WN>))) return rc;
    hipLaunchKernelGGL((conv_mfma_kernel<MI, NJ, WM, WN>), grid, block, lds, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("conv_mfma_kernel launch failed: %s", hipGetErrorString(e));
        return TTSC_EHIP;
    }
    return TTSC_OK;
}

static int pick_mt(int Cout) {
    int best = 32, best_pad = (int)round_up(Cout, 32);
    for (int mt : {64, 128}) {
        int pad = (int)round_up(Cout, mt);
        if (pad <= best_pad) {
            best = mt;
            best_pad = pad;
        }
    }
    return best;
}

}  // namespace ttsc

using namespace ttsc;

#include "conv_internal.hpp"

extern "C" int ttsc_conv1d_create(const ttsc_conv1d_cfg* cfg, ttsc_conv1d** out) {
    TTSC_REQUIRE(cfg && out, "ttsc_conv1d_create: null argument");
    TTSC_REQUIRE(cfg->in_channels > 0 && cfg->out_channels > 0 && cfg->kernel_size > 0,
                 "ttsc_conv1d_create: bad channels/kernel (%d,%d,%d)", cfg->in_channels, cfg->out_channels,
                 cfg->kernel_size);
    TTSC_REQUIRE(cfg->stride >= 1 && cfg->dilation >= 1 && cfg->padding >= 0, "ttsc_conv1d_create: bad stride/dilation/padding");
    if (cfg->transposed) {
        TTSC_REQUIRE(cfg->dilation == 1, "ttsc_conv1d_create: ConvTranspose1d supports dilation 1 only");
    } else {
        TTSC_REQUIRE(cfg->stride == 1, "ttsc_conv1d_create: Conv1d supports stride 1 only");
    }
    const int groups = cfg->groups > 1 ? cfg->groups : 1;
    if (groups > 1) {
        TTSC_REQUIRE(!cfg->transposed, "ttsc_conv1d_create: groups need a Conv1d");
        TTSC_REQUIRE(cfg->in_channels % groups == 0 && cfg->out_channels % groups == 0, "ttsc_conv1d_create: channels (%d, %d) not divisible by groups %d",
                     cfg->in_channels, cfg->out_channels, groups);
    }
    ttsc_conv1d* c = new ttsc_conv1d();
    c->cfg = *cfg;
    c->cfg.groups = groups;
    c->vfused = cfg->transposed && (cfg->out_channels % 32 == 0) && cfg->stride > 1;
    c->CoutV = c->vfused ? cfg->stride * cfg->out_channels : cfg->out_channels;
    c->MT = pick_mt(c->CoutV);
    c->groups = groups;
    c->cin_g = cfg->in_channels / groups;
    c->cout_g = cfg->out_channels / groups;
    c->cin_tile = cfg->in_channels;
    if (groups > 1) {
        // an M tile must not straddle a group boundary unless it holds whole groups
        c->MT = c->cout_g >= 128 && c->cout_g % 128 == 0 ? 128 : (c->cout_g >= 64 && c->cout_g % 64 == 0 ? 64 : 32);
        if (!(c->cout_g % c->MT == 0 || c->MT % c->cout_g == 0)) {
            delete c;
            set_error("ttsc_conv1d_create: %d output channels per group do not tile into 32-row blocks", cfg->out_channels / groups);
            return TTSC_EINVAL;
        }
        c->cin_tile = (c->MT > c->cout_g ? c->MT / c->cout_g : 1) * c->cin_g;
    }
    c->NT = c->MT == 128 ? 128 : (c->MT == 64 ? 256 : 512);
    c->CinP = (int)round_up(c->cin_tile, KC);
    c->CoutP = (int)round_up(c->CoutV, c->MT);
    *out = c;
    return TTSC_OK;
}

static void free_phases(ttsc_conv1d* c) {
    for (auto& p : c->phases) {
        if (p.wp_dev) (void)hipFree(p.wp_dev);
        if (p.wph_dev) (void)hipFree(p.wph_dev);
    }
    c->phases.clear();
}

extern "C" void ttsc_conv1d_destroy(ttsc_conv1d* c) {
    if (!c) return;
    free_phases(c);
    if (c->bias_dev) (void)hipFree(c->bias_dev);
    if (c->w_plain_dev) (void)hipFree(c->w_plain_dev);
    delete c;
}

extern "C" int64_t ttsc_conv1d_out_len(const ttsc_conv1d* c, int64_t Lin) {
    if (!c) return TTSC_EINVAL;
    const auto& g = c->cfg;
    if (g.transposed) return (Lin - 1) * g.stride - 2 * g.padding + g.kernel_size;
    return Lin + 2 * g.padding - g.dilation * (g.kernel_size - 1);
}

// Pack W into MFMA A-fragment order: [tap][CinP/2][CoutP/32][lane], lane -> (co = cot*32 + (lane&31), ci = 2*cip + (lane>>5))
static void pack_phase(const ttsc_conv1d* c, const float* w, const std::vector<int>& taps_k, std::vector<float>& out) {
    const auto& g = c->cfg;
    const int Cin = g.in_channels, Cout = g.out_channels, K = g.kernel_size;
    const int cipN = c->CinP / 2, cotN = c->CoutP / 32;
    out.assign((size_t)taps_k.size() * cipN * cotN * 64, 0.f);
    for (size_t j = 0; j < taps_k.size(); ++j) {
        const int k = taps_k[j];
        for (int cip = 0; cip < cipN; ++cip)
            for (int cot = 0; cot < cotN; ++cot)
                for (int lane = 0; lane < 64; ++lane) {
                    int co = cot * 32 + (lane & 31);
                    const int ci = 2 * cip + (lane >> 5);
                    int kk = k;
                    bool ok = co < c->CoutV && ci < Cin;
                    if (c->vfused) {   // virtual row = phase r * Cout + co, tap index k -> kernel tap r + k*stride
                        const int r = co / Cout;
                        co -= r * Cout;
                        kk = r + k * g.stride;
                        ok = ok && kk < K;
                    }
                    float v = 0.f;
                    if (c->groups > 1) {   // ci counts from the first input channel of the row tile's group(s); weight [Cout, cin_g, K]
                        const int gco = co / c->cout_g, cig = ((co / c->MT) * c->MT / c->cout_g) * c->cin_g + ci;
                        ok = co < Cout && ci < c->cin_tile && cig / c->cin_g == gco;
                        if (ok) v = w[((size_t)co * c->cin_g + (cig - gco * c->cin_g)) * K + kk];
                    } else if (ok) v = g.transposed ? w[((size_t)ci * Cout + co) * K + kk] : w[((size_t)co * Cin + ci) * K + kk];
                    out[(((size_t)j * cipN + cip) * cotN + cot) * 64 + lane] = v;
                }
    }
}

// f16x3 packing: [tap][CinP/16][CoutP/32][2 (hi,lo)][64 lanes][8 half]; lane -> co = cot*32 + (lane&31),
// ci = 16*chunk + 8*(lane>>5) + e.  Weights are multiplied by `scale` (a power of two) before the split.
static void pack_phase_f16(const ttsc_conv1d* c, const float* w, const std::vector<int>& taps_k, float scale,
                           std::vector<_Float16>& out) {
    const auto& g = c->cfg;
    const int Cin = g.in_channels, Cout = g.out_channels, K = g.kernel_size;
    const int nch = c->CinP / 16, cotN = c->CoutP / 32;
    out.assign((size_t)taps_k.size() * nch * cotN * 2 * 64 * 8, (_Float16)0.f);
    for (size_t j = 0; j < taps_k.size(); ++j) {
        const int k = taps_k[j];
        for (int ch = 0; ch < nch; ++ch)
            for (int cot = 0; cot < cotN; ++cot)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 8; ++e) {
                        int co = cot * 32 + (lane & 31);
                        const int ci = ch * 16 + 8 * (lane >> 5) + e;
                        int kk = k;
                        bool ok = co < c->CoutV && ci < Cin;
                        if (c->vfused && c->vrow4) {   // rows interleaved co * 4 + r (see epilogue_tile_v4)
                            const int r = co & 3;
                            co >>= 2;
                            kk = r + k * g.stride;
                            ok = ok && kk < K;
                        } else if (c->vfused) {
                            const int r = co / Cout;
                            co -= r * Cout;
                            kk = r + k * g.stride;
                            ok = ok && kk < K;
                        }
                        float v = 0.f;
                        if (ok) v = g.transposed ? w[((size_t)ci * Cout + co) * K + kk] : w[((size_t)co * Cin + ci) * K + kk];
                        v *= scale;
                        const _Float16 hi = (_Float16)v;
                        const _Float16 lo = (_Float16)(v - (float)hi);
                        const size_t base = ((((size_t)j * nch + ch) * cotN + cot) * 2) * 64 * 8;
                        out[base + (size_t)lane * 8 + e] = hi;
                        out[base + (size_t)64 * 8 + (size_t)lane * 8 + e] = lo;
                    }
    }
}

static int conv_repack(ttsc_conv1d* c);

extern "C" int ttsc_conv1d_set_precision(ttsc_conv1d* c, int32_t precision) {
    TTSC_REQUIRE(c, "ttsc_conv1d_set_precision: null argument");
    TTSC_REQUIRE(precision == TTSC_PREC_FP32 || precision == TTSC_PREC_F16X3, "ttsc_conv1d_set_precision: unknown precision %d", precision);
    // the split kernel prefetches a fixed-size register window: tile + 64 positions of halo.  Layers with a larger
    // receptive field (none on the hot path) silently stay on the exact fp32 kernel.
    const int halo = c->cfg.transposed ? (c->cfg.kernel_size + c->cfg.stride - 1) / c->cfg.stride - 1
                                       : (c->cfg.kernel_size - 1) * c->cfg.dilation;
    const int taps = c->cfg.transposed ? (c->cfg.kernel_size + c->cfg.stride - 1) / c->cfg.stride : c->cfg.kernel_size;
    if (precision == TTSC_PREC_F16X3 && (halo > 64 || taps > 16 || c->groups > 1)) precision = TTSC_PREC_FP32;   // (grouped layers: fp32 kernel only)
    if (precision == c->precision) return TTSC_OK;
    TTSC_REQUIRE(!c->dev_weights, "ttsc_conv1d_set_precision: weights were set from device memory; switch the precision first");
    c->precision = precision;
    return c->has_weight ? conv_repack(c) : TTSC_OK;
}

extern "C" int ttsc_conv1d_set_weight(ttsc_conv1d* c, const float* w, const float* bias) {
    TTSC_REQUIRE(c && w, "ttsc_conv1d_set_weight: null argument");
    const auto& g0 = c->cfg;
    c->w_host.assign(w, w + (size_t)c->cin_g * g0.out_channels * g0.kernel_size);   // torch layout: [Cout, Cin / groups, K]
    c->has_bias = bias != nullptr;
    if (bias) c->b_host.assign(bias, bias + g0.out_channels);
    c->dev_weights = false;
    return conv_repack(c);
}

// Device-side packing for training: the weights change every optimizer step, so the fragment order is produced by a
// gather kernel straight from the torch parameter (same mapping as pack_phase above), no host round trip.
namespace ttsc {
struct PackArgs {
    const float* w;
    float* out;
    int Cin, Cout, K, CoutV, cipN, cotN, ntaps, k0, kstep, transposed, vfused, stride;
    int flipT;   // source is the forward weight [Cin(this), Cout(this), K] of the layer this handle differentiates
    int groups, cin_g, cout_g, cin_tile, MT;   // grouped Conv1d: see ttsc_conv1d (conv_internal.hpp)
};
__global__ void pack_w_kernel(PackArgs p) {
    const long total = (long)p.ntaps * p.cipN * p.cotN * 64;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        long t = i >> 6;
        const int cot = (int)(t % p.cotN);
        t /= p.cotN;
        const int cip = (int)(t % p.cipN);
        const int j = (int)(t / p.cipN);
        int co = cot * 32 + (lane & 31);
        const int ci = 2 * cip + (lane >> 5);
        int kk = p.k0 + j * p.kstep;
        bool ok = co < p.CoutV && ci < p.Cin;
        if (p.vfused) {
            const int r = co / p.Cout;
            co -= r * p.Cout;
            kk = r + (p.k0 + j * p.kstep) * p.stride;
            ok = ok && kk < p.K;
        }
        float v = 0.f;
        if (p.groups > 1) {
            // ci counts from the first input channel of the row tile's group(s).  Forward weight [Cout, cin_g, K]; as the data
            // gradient of a grouped layer the source is THAT layer's weight [Cin(this), cout_g(this), K], flipped in k.
            const int gco = co / p.cout_g, cig = ((co / p.MT) * p.MT / p.cout_g) * p.cin_g + ci;
            if (co < p.Cout && ci < p.cin_tile && cig / p.cin_g == gco)
                v = p.flipT ? p.w[((size_t)cig * p.cout_g + (co - gco * p.cout_g)) * p.K + (p.K - 1 - kk)]
                            : p.w[((size_t)co * p.cin_g + (cig - gco * p.cin_g)) * p.K + kk];
        } else if (ok) {
            if (p.flipT)
                v = p.w[((size_t)ci * p.Cout + co) * p.K + (p.K - 1 - kk)];
            else
                v = p.transposed ? p.w[((size_t)ci * p.Cout + co) * p.K + kk] : p.w[((size_t)co * p.Cin + ci) * p.K + kk];
        }
        p.out[i] = v;
    }
}
}  // namespace ttsc

static int set_weight_device_impl(ttsc_conv1d* c, const float* w_dev, const float* bias_dev, int flipT, void* stream);

extern "C" int ttsc_conv1d_set_weight_device(ttsc_conv1d* c, const float* w_dev, const float* bias_dev, void* stream) {
    return set_weight_device_impl(c, w_dev, bias_dev, 0, stream);
}

extern "C" int ttsc_conv1d_set_weight_device_dgrad(ttsc_conv1d* c, const float* fwd_weight_dev, void* stream) {
    TTSC_REQUIRE(c && !c->cfg.transposed, "ttsc_conv1d_set_weight_device_dgrad: the data-gradient handle must be a Conv1d");
    return set_weight_device_impl(c, fwd_weight_dev, nullptr, 1, stream);
}

static int set_weight_device_impl(ttsc_conv1d* c, const float* w_dev, const float* bias_dev, int flipT, void* stream) {
    TTSC_REQUIRE(c && w_dev, "ttsc_conv1d_set_weight_device: null argument");
    TTSC_REQUIRE(c->precision == TTSC_PREC_FP32, "ttsc_conv1d_set_weight_device: only TTSC_PREC_FP32 handles take device weights");
    const auto& g = c->cfg;
    if (c->phases.empty() || (bias_dev != nullptr) != (c->bias_dev != nullptr)) {
        // first call: lay out the phase buffers (zero weights), later calls only overwrite them
        c->w_host.assign((size_t)c->cin_g * g.out_channels * g.kernel_size, 0.f);   // (cin_g == in_channels unless grouped)
        c->has_bias = bias_dev != nullptr;
        c->b_host.assign(g.out_channels, 0.f);
        int rc = conv_repack(c);
        if (rc) return rc;
    }
    hipStream_t s = (hipStream_t)stream;
    for (auto& ph : c->phases) {
        PackArgs p;
        p.w = w_dev;
        p.out = ph.wp_dev;
        p.Cin = g.in_channels;
        p.Cout = g.out_channels;
        p.K = g.kernel_size;
        p.CoutV = c->CoutV;
        p.cipN = c->CinP / 2;
        p.cotN = c->CoutP / 32;
        p.ntaps = ph.ntaps;
        p.k0 = ph.taps.empty() ? 0 : ph.taps[0];
        p.kstep = ph.taps.size() > 1 ? ph.taps[1] - ph.taps[0] : 1;
        p.transposed = g.transposed;
        p.vfused = c->vfused ? 1 : 0;
        p.stride = g.stride;
        p.flipT = flipT;
        p.groups = c->groups;
        p.cin_g = c->cin_g;
        p.cout_g = c->cout_g;
        p.cin_tile = c->cin_tile;
        p.MT = c->MT;
        const long total = (long)p.ntaps * p.cipN * p.cotN * 64;
        const int blocks = (int)std::min<long>((total + 255) / 256, 2048);
        hipLaunchKernelGGL(pack_w_kernel, dim3(blocks), dim3(256), 0, s, p);
    }
    c->bias_ext = bias_dev;   // read straight from the caller's tensor by the launches that follow (no copy)
    c->w_plain_ext = flipT ? nullptr : w_dev;   // torch layout, used as is by the single-output-channel kernel
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("pack_w_kernel launch failed: %s", hipGetErrorString(e));
        return TTSC_EHIP;
    }
    c->dev_weights = true;
    return TTSC_OK;
}

static int conv_repack(ttsc_conv1d* c) {
    const float* w = c->w_host.data();
    const float* bias = c->has_bias ? c->b_host.data() : nullptr;
    TTSC_REQUIRE(c && w, "ttsc_conv1d_set_weight: null argument");
    const auto& g = c->cfg;
    free_phases(c);
    std::vector<float> packed;
    std::vector<_Float16> packed_h;
    // power-of-two weight scale for the split path: max|w| * 2^s in [512, 1024) keeps the low halves normal in fp16
    float wscale = 1.f;
    if (c->precision == TTSC_PREC_F16X3) {
        float mx = 0.f;
        for (float v : c->w_host) mx = fmaxf(mx, fabsf(v));
        int e = 0;
        if (mx > 0.f && std::isfinite(mx)) {
            (void)frexpf(mx, &e);   // mx = m * 2^e, m in [0.5, 1)
            e = 10 - e;             // mx * 2^e in [512, 1024)
            if (e > 24) e = 24;
            if (e < -8) e = -8;
        }
        wscale = ldexpf(1.f, e);
        c->w_unscale = ldexpf(1.f, -e);
    }
    // kernel_size == stride == 4 (the last two upsamplers of HiFi-GAN V1): interleave the virtual rows so that a lane's
    // accumulator registers are four consecutive output samples (epilogue_tile_v4)
    c->vrow4 = c->vfused && c->precision == TTSC_PREC_F16X3 && g.stride == 4 && g.kernel_size == 4 && g.padding == 0;
    std::vector<int> cur_taps;
    auto upload = [&](ConvPhase& ph) -> int {
        if (c->precision == TTSC_PREC_F16X3) {
            pack_phase_f16(c, w, cur_taps, wscale, packed_h);
            TTSC_HIP_CHECK(hipMalloc((void**)&ph.wph_dev, packed_h.size() * sizeof(_Float16)));
            TTSC_HIP_CHECK(hipMemcpy(ph.wph_dev, packed_h.data(), packed_h.size() * sizeof(_Float16), hipMemcpyHostToDevice));
            return TTSC_OK;
        }
        pack_phase(c, w, cur_taps, packed);
        TTSC_HIP_CHECK(hipMalloc((void**)&ph.wp_dev, packed.size() * sizeof(float)));
        TTSC_HIP_CHECK(hipMemcpy(ph.wp_dev, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice));
        return TTSC_OK;
    };
    if (!g.transposed) {
        ConvPhase ph;
        std::vector<int> taps;
        for (int k = 0; k < g.kernel_size; ++k) taps.push_back(k);
        ph.ntaps = g.kernel_size;
        ph.tap_base = -g.padding;
        ph.tap_step = g.dilation;
        ph.out_stride = 1;
        ph.out_off = 0;
        cur_taps = taps;
        ph.taps = taps;
        int rc = upload(ph);
        if (rc) return rc;
        c->phases.push_back(ph);
    } else if (c->vfused) {
        ConvPhase ph;
        std::vector<int> taps;
        const int J = (g.kernel_size + g.stride - 1) / g.stride;
        for (int j = 0; j < J; ++j) taps.push_back(j);   // tap INDEX; the packers map (phase, index) -> kernel tap
        ph.r = -1;
        ph.ntaps = J;
        ph.tap_base = 0;
        ph.tap_step = -1;
        ph.out_stride = g.stride;
        ph.out_off = -g.padding;
        cur_taps = taps;
        ph.taps = taps;
        int rc = upload(ph);
        if (rc) return rc;
        c->phases.push_back(ph);
    } else {
        for (int r = 0; r < g.stride; ++r) {
            ConvPhase ph;
            std::vector<int> taps;
            for (int k = r; k < g.kernel_size; k += g.stride) taps.push_back(k);
            if (taps.empty()) continue;  // K < stride: this phase only gets the bias (handled below)
            ph.r = r;
            ph.ntaps = (int)taps.size();
            ph.tap_base = 0;
            ph.tap_step = -1;
            ph.out_stride = g.stride;
            ph.out_off = r - g.padding;
            cur_taps = taps;
            ph.taps = taps;
            int rc = upload(ph);
            if (rc) return rc;
            c->phases.push_back(ph);
        }
        TTSC_REQUIRE((int)c->phases.size() == g.stride, "ConvTranspose1d with kernel_size < stride is not supported");
    }
    if (g.transposed) {
        TTSC_REQUIRE(g.kernel_size >= g.stride, "ConvTranspose1d with kernel_size < stride is not supported");
    }
    if (c->bias_dev) {
        (void)hipFree(c->bias_dev);
        c->bias_dev = nullptr;
    }
    if (bias) {
        TTSC_HIP_CHECK(hipMalloc((void**)&c->bias_dev, g.out_channels * sizeof(float)));
        TTSC_HIP_CHECK(hipMemcpy(c->bias_dev, bias, g.out_channels * sizeof(float), hipMemcpyHostToDevice));
    }
    if (c->w_plain_dev) {
        (void)hipFree(c->w_plain_dev);
        c->w_plain_dev = nullptr;
    }
    if (!g.transposed && g.out_channels == 1) {   // conv_cout1_kernel reads the weights in torch layout [1][Cin][K]
        TTSC_HIP_CHECK(hipMalloc((void**)&c->w_plain_dev, c->w_host.size() * sizeof(float)));
        TTSC_HIP_CHECK(hipMemcpy(c->w_plain_dev, c->w_host.data(), c->w_host.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    c->w_plain_ext = nullptr;
    c->has_weight = true;
    return TTSC_OK;
}

extern "C" int ttsc_conv1d_forward(const ttsc_conv1d* c, const float* x, int32_t B, int64_t Lin, float* y,
                                   const float* resid, const ttsc_conv1d_epilogue* ep, void* stream) {
    return ttsc_conv1d_forward_ragged(c, x, B, Lin, y, resid, ep, nullptr, nullptr, stream);
}

extern "C" int32_t ttsc_conv1d_in_channels(const ttsc_conv1d* c) { return c ? c->cfg.in_channels : 0; }

// Split precision carries an fp32 value as two fp16 halves, so an activation tensor must sit inside fp16's range: with |x|
// below ~2^-3 the low half goes subnormal (the accuracy degrades towards fp16's), above 65504 the high half overflows.  The
// layer's input is therefore multiplied by a power of two while it is staged (exact) and the accumulated result by its
// inverse in the epilogue (exact; leaky-relu is positively homogeneous).  scale = 1 restores the plain behaviour.
extern "C" int ttsc_conv1d_set_activation_scale(ttsc_conv1d* c, float scale) {
    TTSC_REQUIRE(c, "ttsc_conv1d_set_activation_scale: null argument");
    int e = 0;
    TTSC_REQUIRE(scale > 0.f && std::isfinite(scale) && frexpf(scale, &e) == 0.5f, "ttsc_conv1d_set_activation_scale: scale must be a power of two, got %g", scale);
    c->act_scale = scale;
    return TTSC_OK;
}
extern "C" float ttsc_conv1d_get_activation_scale(const ttsc_conv1d* c) { return c ? c->act_scale : 0.f; }

extern "C" int ttsc_conv1d_set_nonfinite_flag(ttsc_conv1d* c, uint32_t* flag_dev) {
    TTSC_REQUIRE(c, "ttsc_conv1d_set_nonfinite_flag: null argument");
    c->nf_flag = flag_dev;
    return TTSC_OK;
}

namespace ttsc {
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long n, unsigned* __restrict__ out) {
    __shared__ float red[4];
    float m = 0.f;
    const long stride = (long)gridDim.x * blockDim.x;
    const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if ((reinterpret_cast<uintptr_t>(x) & 15) == 0) {   // 16-byte loads over the aligned body, the tail element-wise
        const long n4 = n >> 2;
        const float4* x4 = reinterpret_cast<const float4*>(x);
        for (long i = i0; i < n4; i += stride) {
            const float4 v = x4[i];
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
        for (long i = (n4 << 2) + i0; i < n; i += stride) m = fmaxf(m, fabsf(x[i]));
    } else {
        for (long i = i0; i < n; i += stride) m = fmaxf(m, fabsf(x[i]));
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    // ONE atomic per workgroup (the first version sent one per wave from up to 4096 workgroups to the same word: 0.19 ms for a 16 MB tensor)
    if (threadIdx.x == 0) atomicMax(out, __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));   // non-negative floats order like their bit patterns
}
}  // namespace ttsc

// out_dev (one float, device) = max(out_dev, max_i |x_i|); the caller zeroes it first.  Non-finite inputs end up as a NaN/inf
// bit pattern, which compares above every finite value: the caller sees it.
extern "C" int ttsc_absmax(const float* x_dev, int64_t n, float* out_dev, void* stream) {
    TTSC_REQUIRE(x_dev && out_dev && n > 0, "ttsc_absmax: bad argument");
    const int blocks = (int)std::min<int64_t>((n / 4 + 255) / 256 + 1, 1024);
    hipLaunchKernelGGL(absmax_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x_dev, (long)n, reinterpret_cast<unsigned*>(out_dev));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("absmax_kernel launch failed: %s", hipGetErrorString(e));
        return TTSC_EHIP;
    }
    return TTSC_OK;
}

extern "C" int ttsc_conv1d_forward_ragged(const ttsc_conv1d* c, const float* x, int32_t B, int64_t Lin, float* y,
                                          const float* resid, const ttsc_conv1d_epilogue* ep, const int32_t* in_len_dev,
                                          const int32_t* out_len_dev, void* stream) {
    return ttsc_conv1d_forward_pitched(c, x, B, Lin, y, resid, ep, in_len_dev, out_len_dev, 0, stream);
}

extern "C" int ttsc_conv1d_forward_pitched(const ttsc_conv1d* c, const float* x, int32_t B, int64_t Lin, float* y,
                                           const float* resid, const ttsc_conv1d_epilogue* ep, const int32_t* in_len_dev,
                                           const int32_t* out_len_dev, int64_t Lout_pitch, void* stream) {
    TTSC_REQUIRE(c && x && y, "ttsc_conv1d_forward: null argument");
    TTSC_REQUIRE(Lout_pitch == 0 || (Lout_pitch > 0 && in_len_dev && out_len_dev),
                 "ttsc_conv1d_forward_pitched: an output pitch needs per-utterance lengths (in_len_dev, out_len_dev)");
    if (!c->has_weight) {
        set_error("ttsc_conv1d_forward: weights not set");
        return TTSC_ESTATE;
    }
    TTSC_REQUIRE(B > 0 && Lin > 0, "ttsc_conv1d_forward: bad B/Lin (%d, %lld)", B, (long long)Lin);
    TTSC_REQUIRE(!ep || (ep->in_slope >= 0.f && ep->in_slope <= 1.f), "ttsc_conv1d_forward: in_slope must be in [0,1]");
    const auto& g = c->cfg;
    // (with a pitch, positions between an utterance's real length and the pitch are computed from zero-padded input and land in the row's padding)
    const int64_t Lout = Lout_pitch > 0 ? Lout_pitch : ttsc_conv1d_out_len(c, Lin);
    TTSC_REQUIRE(Lout > 0, "ttsc_conv1d_forward: output length %lld <= 0", (long long)Lout);
    TTSC_REQUIRE(Lin < (1ll << 30) && Lout < (1ll << 30), "ttsc_conv1d_forward: length too large");
    hipStream_t s = (hipStream_t)stream;
    for (const auto& ph : c->phases) {
        ConvArgs a;
        a.x = x;
        a.y = y;
        a.resid = resid;
        a.nf_flag = nullptr;
        a.wp = ph.wp_dev;
        a.wph = ph.wph_dev;
        a.w_unscale = c->w_unscale;
        a.bias = (c->dev_weights && c->bias_ext) ? c->bias_ext : c->bias_dev;
        a.in_len = in_len_dev;
        a.out_len = out_len_dev;
        a.dbg = 0;
        a.skew = 0;
        a.acc_init = 0;
        a.epi_prefetch = 0;
        a.swz_nx = a.swz_ny = 0;
        a.fold_S = a.fold_B = 0;
        a.amax_x = a.amax_w = nullptr;
        a.amax_out = nullptr;
#ifdef TTSC_ABLATE
        if (const char* ev = getenv("TTSC_CONV_DBG")) a.dbg = atoi(ev);
        a.prof = nullptr;
        if (const char* ev = getenv("TTSC_PROF_PTR")) a.prof = reinterpret_cast<unsigned long long*>(strtoull(ev, nullptr, 0));
#endif
        a.vphase = c->vfused ? ((c->vrow4 && c->precision == TTSC_PREC_F16X3) ? -4 : g.out_channels) : 0;
        if (a.vphase == -4) {
            TTSC_REQUIRE(!(ep && ep->gate_dev), "ttsc_conv1d_forward: the gate epilogue is not available on this transposed layer");
            TTSC_REQUIRE(((uintptr_t)y % 16 == 0) && (!resid || (uintptr_t)resid % 16 == 0), "ttsc_conv1d_forward: y / resid must be 16-byte aligned for ConvTranspose1d(k=4, s=4)");
        }
        a.Cin = c->groups > 1 ? c->cin_tile : g.in_channels;
        a.CinTot = g.in_channels;
        a.groups = c->groups;
        a.cin_g = c->cin_g;
        a.cout_g = c->cout_g;
        a.CinP = c->CinP;
        a.Cout = g.out_channels;
        a.CoutP = c->CoutP;
        a.Lin = (int)Lin;
        a.Lout = (int)Lout;
        a.ntaps = ph.ntaps;
        a.tap_base = ph.tap_base;
        a.tap_step = ph.tap_step;
        a.out_stride = ph.out_stride;
        a.out_off = ph.out_off;
        if (!g.transposed) {
            a.q_lo = 0;
            a.q_cnt = (int)Lout;
        } else if (c->vfused) {
            // union over the phases r = 0..s-1 of  o = q*s + r - p in [0, Lout); per-element validity is checked in the epilogue
            int64_t qlo = -floor_div((g.stride - 1) - g.padding, g.stride);
            int64_t qhi = floor_div(Lout - 1 + g.padding, g.stride) + 1;
            if (qhi <= qlo) continue;
            a.q_lo = (int)qlo;
            a.q_cnt = (int)(qhi - qlo);
        } else {
            // o = q*s + r - p in [0, Lout)
            int64_t qlo = -floor_div(ph.r - g.padding, g.stride);  // ceil((p - r) / s)
            int64_t qhi = floor_div(Lout - 1 - ph.r + g.padding, g.stride) + 1;      // exclusive
            if (qhi <= qlo) continue;
            a.q_lo = (int)qlo;
            a.q_cnt = (int)(qhi - qlo);
        }
        const int last = (ph.ntaps - 1) * ph.tap_step;
        a.min_shift = ph.tap_base + (last < 0 ? last : 0);
        a.span = c->NT + (last < 0 ? -last : last);
        a.span_pad = a.span + 1;
        a.in_scale = ep ? ep->in_scale : 1.f;
        const float* w_plain = c->dev_weights ? c->w_plain_ext : c->w_plain_dev;
        const bool use_cout1 = !g.transposed && c->groups == 1 && g.out_channels == 1 && g.in_channels <= 64 && g.kernel_size <= 16 && g.dilation == 1 &&
                               w_plain && !(ep && ep->gate_dev);
        if (c->precision == TTSC_PREC_F16X3 && !use_cout1) {   // activation pre-scale (exact powers of two, see ttsc_conv1d_set_activation_scale)
            a.in_scale *= c->act_scale;
            a.w_unscale = c->w_unscale / c->act_scale;
        }
        a.in_slope = ep ? ep->in_slope : 1.f;
        a.out_scale = ep ? ep->out_scale : 1.f;
        a.out_act = ep ? ep->out_act : TTSC_ACT_NONE;
        a.accumulate = ep ? ep->accumulate : 0;
        a.gate = ep ? ep->gate_dev : nullptr;
        a.gate_slope = ep ? ep->gate_slope : 1.f;
        int rc;
        if (use_cout1) {
            // one output channel (conv_post): vector-ALU kernel, bound by the single read of its input
            a.wp = w_plain;
            a.nf_flag = c->nf_flag;
            dim3 grid((unsigned)ceil_div(Lout, 1024), (unsigned)B);
            hipLaunchKernelGGL(conv_cout1_kernel, grid, dim3(256), 0, s, a);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) {
                set_error("conv_cout1_kernel launch failed: %s", hipGetErrorString(e));
                return TTSC_EHIP;
            }
            rc = TTSC_OK;
        } else if (c->precision == TTSC_PREC_F16X3) {
            // the split kernel's N tile is fixed by its 4-waves-along-N shape
            const int nt = c->MT >= 64 ? 256 : 512;
            a.span = nt + (last < 0 ? -last : last);
            a.span_pad = a.span;
            TTSC_REQUIRE(ph.ntaps <= 16, "f16x3 path supports at most 16 taps per phase (got %d)", ph.ntaps);
            // tile by machine fill (same rule as the fp32 path): a single short utterance (the reference API's B=1 case)
            // has only a few thousand positions per layer, so the big tiles would occupy a fraction of the 256 CUs
            auto wgs16 = [&](int mt, int nt) { return (long)ceil_div(a.q_cnt, nt) * (c->CoutP / mt) * B; };
            auto set_nt16 = [&](int n) {
                a.span = n + (last < 0 ? -last : last);
                a.span_pad = a.span;
            };
            static const bool big_only = getenv("TTSC_F16_TILE") && atoi(getenv("TTSC_F16_TILE")) == 0;
            const char* wide_ev = getenv("TTSC_CONV_WIDE");
            const int wide_env = wide_ev ? atoi(wide_ev) : 1;   // 0 = off, 2 = also for problems too small to fill the chip (tests)
            const long want16 = 512;
            // square "same"-padded layers of the wide stages (the generator's ResBlock convolutions at 128 / 256 channels):
            // the wide-tile kernel (activation window staged once for all output channels)
            const bool wide_shape = !g.transposed && !a.gate && g.in_channels == g.out_channels &&
                                    (g.out_channels == 128 || g.out_channels == 256) &&
                                    ((g.kernel_size == 3 || g.kernel_size == 7 || g.kernel_size == 11)
#ifdef TTSC_PROBE_EVENK
                                     || ((g.kernel_size == 4 || g.kernel_size == 6) && g.dilation == 1)
#endif
                                     ) &&
                                    (g.dilation == 1 || g.dilation == 3 || g.dilation == 5) &&
                                    g.padding == g.dilation * (g.kernel_size - 1) / 2;
            // these layers start their sums at (residual + running sum + bias) / w_unscale in BOTH kernels that may run them (same bits whichever
            // the fill rule picks); TTSC_CONV_ACC_INIT=0 puts the operands back into the epilogue (measurement switch)
            static const int acc_init_env = getenv("TTSC_CONV_ACC_INIT") ? atoi(getenv("TTSC_CONV_ACC_INIT")) : 1;
            a.acc_init = (acc_init_env && wide_env && wide_shape && a.out_act == TTSC_ACT_NONE && a.out_scale == 1.f && c->CoutP == g.out_channels &&
                          (size_t)B * g.out_channels * Lout < (1ull << 32)) ? 1 : 0;
            // the first upsamplers of the V1 generator: tall tiles (256 virtual rows x 128 input positions)
            static const bool tall_on = !(getenv("TTSC_CONV_TALL") && atoi(getenv("TTSC_CONV_TALL")) == 0);
            const bool tall0 = g.in_channels == 512 && ph.ntaps == 4 && c->CoutP % 256 == 0;    // ups.0: 256-row tiles
            const bool tall1 = g.in_channels == 256 && ph.ntaps == 6 && c->CoutP % 128 == 0;    // ups.1: 128-row tiles
            const bool tall_shape = tall_on && c->vfused && a.vphase > 0 && !a.gate && c->CinP == g.in_channels && (tall0 || tall1);
            if (tall_shape && (wide_env == 2 || (long)ceil_div(a.q_cnt, tall0 ? 128 : 256) * (c->CoutP / (tall0 ? 256 : 128)) * B >= want16)) {
                rc = tall0 ? launch_f16_tall<512, 4, 4, 2>(a, B, s) : launch_f16_tall<256, 6, 2, 4>(a, B, s);
            } else if (wide_env && wide_shape && (wide_env == 2 || (long)ceil_div(a.Lout, 256) * (g.out_channels / 128) * B >= want16)) {
                if (g.out_channels == 256)
                    rc = launch_f16_wide_k<256>(a, B, g.dilation, s);
                else
                    rc = launch_f16_wide_k<128>(a, B, g.dilation, s);
            } else if (c->MT >= 64) {
                if (big_only || wgs16(64, 256) >= want16)
                    rc = launch_f16<2, 2>(a, B, s);   // 64 x 256 tile (MT=128 layers run as two M tiles)
                else if (wgs16(64, 128) >= want16) {
                    set_nt16(128);
                    rc = launch_f16<2, 1>(a, B, s);
                } else {
                    set_nt16(128);
                    rc = launch_f16<1, 1>(a, B, s);
                }
            } else {
                if (big_only || wgs16(32, 512) >= want16)
                    rc = launch_f16<1, 4>(a, B, s);   // 32 x 512 tile
                else if (wgs16(32, 256) >= want16) {
                    set_nt16(256);
                    rc = launch_f16<1, 2>(a, B, s);
                } else {
                    set_nt16(128);
                    rc = launch_f16<1, 1>(a, B, s);
                }
            }
        } else {
            // fp32 tile by machine fill: the largest tile that still gives >= 2 workgroups per CU (training crops and
            // single short utterances are small problems: a 128 x 128 tile would leave most of the 256 CUs idle)
            auto wgs = [&](int mt, int nt) { return (long)ceil_div(a.q_cnt, nt) * (c->CoutP / mt) * B; };
            auto set_nt = [&](int nt) {
                a.span = nt + (last < 0 ? -last : last);
                a.span_pad = a.span + 1;
            };
            static const long want_env = getenv("TTSC_CONV_WANT") ? atol(getenv("TTSC_CONV_WANT")) : 512;
            static const int narrow_env = getenv("TTSC_CONV_NARROW") ? atoi(getenv("TTSC_CONV_NARROW")) : 1;
            const long want = want_env;
            // ... and, among the tiles that fill the machine, not one that is mostly padding: the deep layers of the discriminators see
            // sequences of 50-200 positions (a 128-column tile over 149 positions computes 256), so a narrower tile is taken whenever it
            // cuts the padded columns by more than an eighth (this kernel is bound by its MFMA count, not by operand reuse)
            auto padded = [&](int nt) { return (long)ceil_div(a.q_cnt, nt) * nt; };
            auto narrower_pays = [&](int nt_big, int nt_small) { return narrow_env && padded(nt_small) * 8 < padded(nt_big) * 7; };
            if (c->MT == 128) {
                if (wgs(128, 128) >= want && !narrower_pays(128, 64)) { set_nt(128); rc = launch_cfg<2, 2, 2, 2>(a, B, s); }
                else if (wgs(64, 128) >= want && !narrower_pays(128, 64)) { set_nt(128); rc = launch_cfg<1, 2, 2, 2>(a, B, s); }
                else { set_nt(64); rc = launch_cfg<1, 1, 2, 2>(a, B, s); }
            } else if (c->MT == 64) {
                if (wgs(64, 256) >= want && !narrower_pays(256, 128)) { set_nt(256); rc = launch_cfg<2, 2, 1, 4>(a, B, s); }
                else if (wgs(64, 128) >= want && !narrower_pays(128, 64)) { set_nt(128); rc = launch_cfg<1, 2, 2, 2>(a, B, s); }
                else { set_nt(64); rc = launch_cfg<1, 1, 2, 2>(a, B, s); }
            } else {
                if (wgs(32, 512) >= want && !narrower_pays(512, 256)) { set_nt(512); rc = launch_cfg<1, 4, 1, 4>(a, B, s); }
                else if (wgs(32, 256) >= want && !narrower_pays(256, 128)) { set_nt(256); rc = launch_cfg<1, 2, 1, 4>(a, B, s); }
                else { set_nt(128); rc = launch_cfg<1, 1, 1, 4>(a, B, s); }
            }
        }
        if (rc) return rc;
    }
    return TTSC_OK;
}


// Fused  y = x + conv2(lrelu(conv1(lrelu(x))))  for 32-channel ResBlock1 pairs (see respair32_f16x3_kernel).
extern "C" int ttsc_respair_supported(const ttsc_conv1d* c1, const ttsc_conv1d* c2) {
    if (!c1 || !c2) return 0;
    const auto &g1 = c1->cfg, &g2 = c2->cfg;
    const int k = g1.kernel_size;
    if (g1.transposed || g2.transposed) return 0;
    if (g1.in_channels != 32 || g1.out_channels != 32 || g2.in_channels != 32 || g2.out_channels != 32) return 0;
    if (g2.kernel_size != k || !(k == 3 || k == 7 || k == 11)) return 0;
    if (g2.dilation != 1 || g2.padding != (k - 1) / 2 || g1.padding != g1.dilation * (k - 1) / 2) return 0;
    if ((k - 1) * g1.dilation > 64) return 0;
    if (c1->precision != TTSC_PREC_F16X3 || c2->precision != TTSC_PREC_F16X3) return 0;
    if (!c1->has_weight || !c2->has_weight || !c1->bias_dev || !c2->bias_dev) return 0;
    return 1;
}

extern "C" int ttsc_respair_forward(const ttsc_conv1d* c1, const ttsc_conv1d* c2, const float* x, int32_t B, int64_t L, float* y,
                                    int32_t accumulate, const int32_t* len_dev, void* stream) {
    TTSC_REQUIRE(c1 && c2 && x && y, "ttsc_respair_forward: null argument");
    TTSC_REQUIRE(ttsc_respair_supported(c1, c2), "ttsc_respair_forward: this pair of layers is not eligible for the fused kernel");
    TTSC_REQUIRE(x != y, "ttsc_respair_forward: y must not alias x");
    TTSC_REQUIRE(B > 0 && L > 0 && L < (1ll << 30), "ttsc_respair_forward: bad B/L");
    PairArgs a;
    a.x = x;
    a.y = y;
    a.w1 = c1->phases[0].wph_dev;
    a.w2 = c2->phases[0].wph_dev;
    a.b1 = c1->bias_dev;
    a.b2 = c2->bias_dev;
    a.len = len_dev;
    a.unscale1 = c1->w_unscale;
    a.unscale2 = c2->w_unscale;
    a.L = (int)L;
    a.d1 = c1->cfg.dilation;
    a.accumulate = accumulate;
    const int k = c1->cfg.kernel_size;
    // 4 waves / 256 columns (224 outputs) per workgroup: two workgroups share a CU and hide each other's load latency
    constexpr int NW = 4, NCOL = 64 * NW, NTO = NCOL - 32;
    const int span1 = NCOL + (k - 1) * a.d1;
    const size_t lds = (size_t)span1 * 16 * 2 * 2 + (size_t)(NCOL + 16) * 32 * 2 * 2;
    dim3 grid((unsigned)ceil_div(L, NTO), (unsigned)B);
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipSuccess;
    if (int rc = ensure_full_lds((const void*)respair32_f16x3_kernel<3, NW>)) return rc;
    if (int rc = ensure_full_lds((const void*)respair32_f16x3_kernel<7, NW>)) return rc;
    if (int rc = ensure_full_lds((const void*)respair32_f16x3_kernel<11, NW>)) return rc;
    if (k == 3)
        hipLaunchKernelGGL((respair32_f16x3_kernel<3, NW>), grid, dim3(64 * NW), lds, s, a);
    else if (k == 7)
        hipLaunchKernelGGL((respair32_f16x3_kernel<7, NW>), grid, dim3(64 * NW), lds, s, a);
    else
        hipLaunchKernelGGL((respair32_f16x3_kernel<11, NW>), grid, dim3(64 * NW), lds, s, a);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("respair32_f16x3_kernel launch failed: %s", hipGetErrorString(e));
        return TTSC_EHIP;
    }
    return TTSC_OK;
}
