// Fused HiFi-GAN ResBlock1 chain for the narrow stages (32 / 64 channels), split-precision (f16x3) MFMA, gfx950.
//
//     for p in 0..npairs-1:   x = x + conv2_p( lrelu( conv1_p( lrelu(x) ) ) )           [ y (+)= x at the end ]
//
// (hifigan.models.ResBlock1.forward [EXTERNAL]; reference call sites cube/networks/cubegan.py:72,83 through
// Generator.forward.)  At 32 / 64 channels every convolution of the block is HBM-bound when it runs as its own
// launch (24..140 FLOP per byte moved against a ridge of ~130), so the whole chain is evaluated per time tile:
//
//   * the fp32 residual stream of the tile lives in REGISTERS, in the MFMA C/D layout (the wave that owns a group of
//     columns owns them for every convolution of the chain), so the residual add never touches memory;
//   * ONE activation image in LDS, as fp16 (hi, lo) planes [8-channel group][hi|lo][column] of 16-byte items — the B
//     fragment of any tap is a single conflict-free ds_read_b128.  conv1 reads lrelu(x) from it, its epilogue
//     overwrites it in place with lrelu(conv1 + b1) (one extra barrier instead of a second image), conv2 reads that;
//   * weight fragments (pre-packed in A-fragment order, see conv1d.hip::pack_phase_f16) are shared by all waves of the
//     workgroup: they are copied global -> LDS by LDS-DMA (global_load_lds, no VGPRs) one group of two (tap, chunk)
//     steps ahead of the MFMAs and read as ds_read_b128 like the activations; one barrier per group publishes them;
//   * the tile carries `halo` extra columns per side (sum over the chain of every convolution's half receptive field);
//     columns whose receptive field leaves the tile go stale and are never stored.
//
// HBM traffic per chain: read x once (+ halo), write y once (read-modify-write when accumulating the block sum) —
// 2-3 tensor passes instead of 15 for three unfused pairs.
#include "conv_internal.hpp"
#include "conv_kernels.hpp"

namespace ttsc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

constexpr int RB_MAXP = 3;

// ablation switches: measurement build only (-DTTSC_ABLATE, tools/ablate.cpp), constants in the product library
#ifdef TTSC_ABLATE
#define TTSC_DBG(args, bit) (((args).dbg & (bit)) != 0)
#else
#define TTSC_DBG(args, bit) false
#endif

struct ChainArgs {
    const float* x;       // [B, C, L] chain input
    float* y;             // [B, C, L] chain output (must not alias x: neighbouring tiles read x's halo)
    const half8* w1[RB_MAXP];   // f16x3 fragments [tap][C/16][C/32][hi|lo][64 lanes][8 half]
    const half8* w2[RB_MAXP];
    const float* b1[RB_MAXP];
    const float* b2[RB_MAXP];
    float us1[RB_MAXP], us2[RB_MAXP];   // epilogue factors: conv1 -> image  t' = fma(acc, us1, b1 * bs1);  conv2 -> residual  x += fma(acc, us2, b2)
    float xs[RB_MAXP], bs1[RB_MAXP];    // activation pre-scales (powers of two): image <- xs * lrelu(x);  bs1 = pre-scale of conv2's input
    int d1[RB_MAXP];      // dilation of conv1 (conv2 is undilated)
    const int* len;       // [B] valid length or null
    int L, npairs, accumulate;
    int halo, nto;        // columns of halo per side, output columns per tile (NCOL - 2*halo)
    // POST instantiations (last block of the last stage): y is only READ (the sum of the blocks before this one, when `accumulate`), and
    //     wav = act((conv_post(lrelu((y + chain(x)) * post_in_scale, post_slope)) + bias) * post_out_scale)
    // leaves the kernel instead of the block sum — the same ci-major, tap-minor fmaf chain as conv_cout1_kernel (conv1d.hip): bit-identical
    float* wav;           // [B, 1, L]
    const float* wpost;   // conv_post weights, torch layout [1][C][7]
    const float* bpost;   // [1] or null
    float post_in_scale, post_slope, post_out_scale;
    int post_act;
    unsigned* nf_flag;    // range-guard word (conv_cout1_kernel) or null
#ifdef TTSC_ABLATE
    unsigned long long* prof;   // workgroup phase timeline (TTSC_STAMP) or null; env TTSC_PROF_PTR
#endif
    int dbg;              // -DTTSC_ABLATE builds: 1 skip the epilogue -> image conversions, 2 skip barriers, 4 skip the final store, 8 skip the x load, 16 skip weight loads in the loop
};

typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float2v __attribute__((ext_vector_type(2)));

// MI = C/32 row tiles, K taps, CT 32-column tiles per wave, NW waves per workgroup, WPS = waves per SIMD the register
// allocation has to leave room for (2: two workgroups of 4 waves or one of 8 per CU).  GRP (tap, chunk) steps form one weight
// group = one LDS slot = one barrier interval; NSLOT slots form a ring.  Two slots of two steps at 32 / 64 channels; at 128
// channels the image takes 136 KiB of the 160, so the ring is three slots of ONE step (8 KiB each): the group after next travels
// while the next one — published a whole step earlier — is already being read ahead.
// WM = waves along the channel axis: a wave owns MI / WM row tiles of its CT column tiles (WM = 1: every wave owns all channels of its
// columns).  At 128 channels 2 x 4 waves of (2 row tiles x 2 column tiles) read 8 fragments per 12 MFMAs instead of 10 and hold
// half the weight fragments in registers.
//
// IL ("interleaved columns"): MFMA column tile ct, lane l of a wave stand for tile column  colw + CT * l + ct  instead of
// colw + 32 * ct + l.  A lane's CT accumulator registers of one channel are then CT CONSECUTIVE samples of a [B, C, L] row: the x
// tile comes in and the y tile goes out (and comes in again when it is accumulated) as 16 dwordx4 (CT = 4) / dwordx2 (CT = 2)
// accesses per lane and row tile instead of 64 / 32 dword accesses — the prologue and the read-modify-write epilogue of a workgroup
// that owns its whole CU are exposed, issue-bound time (round-2 ablation: 15 % + 8 % of the K = 11 launch at 32 channels, 26 % + 23 %
// of the K = 3 one).  The LDS image becomes phase-major: plane row = CT phase rows [q = column / CT], so that the B fragment of any
// (tap, tile) is still one conflict-free ds_read_b128 over 32 consecutive q; which phase row and which q shift a tap needs depends on
// the dilation, so the IL convolutions take the dilation as a compile-time constant (1, 3 or 5) and every offset is an immediate.
template <int K, int CT>
struct ChainGeo {
    static constexpr int OMAX = (CT - 1) + 5 * ((K - 1) / 2);       // largest |column offset| a B fragment is read at (dilation <= 5)
    static constexpr int MARGQ = (OMAX + CT - 1) / CT + 1;          // margin of a phase row, in q
    static constexpr int MARG = (K == 3) ? 8 : (K == 7 ? 16 : 26);  // margin of a plain row, in columns
};
template <int V> struct IntTag { static constexpr int value = V; };

template <int MI, int K, int CT, int NW, int WPS, int GRP = 2, int NSLOT = 2, int WM = 1, bool IL = false, bool POST = false>
__global__ __launch_bounds__(64 * NW, WPS) void rbchain_f16x3_kernel(ChainArgs a) {
    static_assert(!POST || (IL && MI == 1 && WM == 1 && CT == 4), "conv_post epilogue: 32 channels, interleaved columns");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int C = 32 * MI, NCH = 2 * MI, NG = 4 * MI;
    constexpr int WN = NW / WM, MIW = MI / WM;    // waves along time; row tiles per wave
    static_assert(NW % WM == 0 && MI % WM == 0, "wave grid");
    constexpr int NCOL = WN * CT * 32;
    constexpr int MARG = ChainGeo<K, CT>::MARG;   // >= the largest tap offset (dilation 5: 5*(K-1)/2)
    constexpr int NQ = NCOL / CT, MARGQ = ChainGeo<K, CT>::MARGQ, PQ = NQ + 2 * MARGQ;   // IL: phase row = MARGQ | NQ | MARGQ items
    constexpr int PW = IL ? CT * PQ : NCOL + 2 * MARG;
    constexpr int H = (K - 1) / 2;
    typedef float fvecT __attribute__((ext_vector_type(CT == 3 ? 4 : CT)));
    static_assert(!IL || CT == 2 || CT == 4, "interleaved columns: 2 or 4 column tiles per wave");
    constexpr int NTHR = 64 * NW;
    constexpr int STEP_ITEMS = MI * 2 * 64;       // 16-byte items of the weight fragments of one (tap, chunk) step
    constexpr int GRP_ITEMS = GRP * STEP_ITEMS;
    constexpr int AHEAD = NSLOT - 1;              // weight groups in flight ahead of the one being multiplied
    constexpr int NS = K * NCH, NGRP = (NS + GRP - 1) / GRP;   // the last group may be short
    half8* P = reinterpret_cast<half8*>(smem_raw);   // plane (group g, pl) at P + (g*2 + pl) * PW, tile column c at + MARG + c
    half8* Aw = P + (size_t)NG * 2 * PW;             // weight fragments: [NSLOT slots][GRP_ITEMS]

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.y;
    const int q0 = blockIdx.x * a.nto;
    const int lin = a.len ? a.len[b] : a.L;
    if (q0 >= lin) return;
#ifdef TTSC_ABLATE
    const unsigned wg_lin = blockIdx.x + gridDim.x * blockIdx.y;
#endif
    TTSC_STAMP(a, wg_lin, 0);
    TTSC_STAMP_HWID(a, wg_lin, 15);
    const int wm = wv / WN;                          // this wave's row-tile group
    const int mi0 = wm * MIW;                        // its first row tile
    const int colw = (wv % WN) * (CT * 32);          // first tile column of this wave
    const int pos_w = q0 - a.halo + colw + (IL ? CT * l31 : l31);   // sequence position of this lane's column in column tile 0
    constexpr int CSTEP = IL ? 1 : 32;               // column distance between the lane's columns of consecutive column tiles
    const int qw = (wv % WN) * 32;                   // IL: first q of this wave
    const float* xb = a.x + (size_t)b * C * a.L;
    // IL: a lane's CT columns move as ONE vector when every group of CT columns is wholly inside or wholly outside the sequence and
    // the rows are vector-aligned (workgroup-uniform); otherwise column by column like the plain layout
    const bool vec = IL && (lin % CT == 0) && (a.L % CT == 0) && ((q0 - a.halo) % CT == 0) && (a.nto % CT == 0) &&
                     ((reinterpret_cast<uintptr_t>(a.x) | reinterpret_cast<uintptr_t>(a.y)) % (4 * CT) == 0);

    // Weight fragments reach the MFMAs through LDS: one group (GRP steps, 4*MI KB) is copied global -> LDS by the LDS-DMA
    // path (no VGPRs, one 1-KB instruction per wave) while the previous group is multiplied; the barrier that ends a group
    // publishes the next one.  Every wave of the workgroup needs the same fragments: fetched per wave straight into registers
    // (the first version of this kernel) they cost a quarter to a third of the run time in the CU's vector-memory path.
    auto stage_group = [&](const half8* w, int g, int slot) __attribute__((always_inline)) {
        constexpr int NIMAX = GRP_ITEMS / 64;   // 1-KB wave instructions per (full) group
        const int NI = ((NS - g * GRP < GRP ? NS - g * GRP : GRP) * STEP_ITEMS) / 64;
        if (TTSC_DBG(a, 16)) return;
#pragma unroll
        for (int i = 0; i < (NIMAX + NW - 1) / NW; ++i) {
            const int blk = wv + i * NW;     // wave-uniform
            if (blk < NI)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w + (size_t)g * GRP_ITEMS + blk * 64 + lane),
                                                 (__attribute__((address_space(3))) void*)(Aw + slot * GRP_ITEMS + blk * 64), 16, 0, 0);
        }
    };
    // the first AHEAD groups of a convolution leave together (into slots 0 .. AHEAD-1, all retired by the barrier that ended the
    // previous convolution); the barrier in front of the convolution publishes them
    auto stage_first = [&](const half8* w) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < AHEAD; ++g) stage_group(w, g, g);
    };
    stage_first(a.w1[0]);

    // the margins only feed columns that are never stored, but they must hold finite numbers
    if constexpr (IL) {
        for (int i = tid; i < NG * 2 * CT * 2 * MARGQ; i += NTHR) {
            const int row = i / (2 * MARGQ), m = i - row * (2 * MARGQ);   // row = (plane, phase)
            half8 z;
#pragma unroll
            for (int e = 0; e < 8; ++e) z[e] = (_Float16)0.f;
            P[(size_t)row * PQ + (m < MARGQ ? m : NQ + m)] = z;
        }
    } else {
        for (int i = tid; i < NG * 2 * 2 * MARG; i += NTHR) {
            const int pl = i / (2 * MARG), m = i - pl * (2 * MARG);
            half8 z;
#pragma unroll
            for (int e = 0; e < 8; ++e) z[e] = (_Float16)0.f;
            P[(size_t)pl * PW + (m < MARG ? m : NCOL + m)] = z;
        }
    }

    // residual stream of the tile: C/D layout, channel = 32*mi + (r & 3) + 8*(r >> 2) + 4*half, column = lane & 31
    f32x16 xres[MIW][CT];
    bool pok[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int pos = pos_w + ct * CSTEP;
        pok[ct] = pos >= 0 && pos < lin;
    }
    if (IL && vec) {
        int p0 = pos_w < lin - CT ? pos_w : lin - CT;
        p0 = p0 < 0 ? 0 : p0;
        const unsigned voff = (unsigned)(4 * half * a.L + p0);
#pragma unroll
        for (int mi = 0; mi < MIW; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float* row = xb + (size_t)(32 * (mi0 + mi) + (r & 3) + 8 * (r >> 2)) * a.L;
                const fvecT v = TTSC_DBG(a, 8) ? fvecT(1e-3f) : *reinterpret_cast<const fvecT*>(row + voff);   // (a non-temporal hint on these loads measured nothing: round 5)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) xres[mi][ct][r] = v[ct];
            }
    } else {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int pos = pos_w + ct * CSTEP;
            int pc = pos < lin - 1 ? pos : lin - 1;
            pc = pc < 0 ? 0 : pc;
#pragma unroll
            for (int mi = 0; mi < MIW; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ch = 32 * (mi0 + mi) + (r & 3) + 8 * (r >> 2) + 4 * half;
                    xres[mi][ct][r] = TTSC_DBG(a, 8) ? (float)(ch + pc) * 1e-3f : xb[(size_t)ch * a.L + pc];
                }
        }
    }

#ifdef TTSC_ABLATE
    if (a.prof) {
        TTSC_STAMP(a, wg_lin, 16);           // x loads issued
        __builtin_amdgcn_s_waitcnt(0);
        TTSC_STAMP(a, wg_lin, 17);           // ... and landed (this wave's)
    }
#endif
    // four channels (one lane's share of 8-channel group 4*mi + gi) of one column -> (hi, lo) halves in the image.
    // Written pairwise so that the conversions are packed: cvt_pk (hi), two cvt back, two subtractions, cvt_pk (lo).
    // The image columns outside the sequence hold zeros (the convolutions pad with zeros): they are zeroed ONCE below and never written
    // again — every later store is predicated on the lane's column being inside (an exec-masked ds_write instead of one v_cndmask per value).
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    auto image_ptr = [&](int mi, int ct, int gi) __attribute__((always_inline)) {
        return reinterpret_cast<_Float16*>(P + (size_t)(((mi0 + mi) * 4 + gi) * 2) * PW + (IL ? ct * PQ + MARGQ + qw + l31 : MARG + colw + ct * 32 + l31)) + 4 * half;
    };
    auto store_split = [&](int mi, int ct, int gi, float v0, float v1, float v2, float v3) __attribute__((always_inline)) {
        if (TTSC_DBG(a, 1)) return;
        unsigned h0, l0, h1, l1;
        split2_f16(v0, v1, h0, l0);
        split2_f16(v2, v3, h1, l1);
        const u32x2 vh = {h0, h1}, vl = {l0, l1};
        _Float16* ph = image_ptr(mi, ct, gi);
        if (pok[ct]) {
            *reinterpret_cast<u32x2*>(ph) = vh;
            *reinterpret_cast<u32x2*>(ph + (size_t)PW * 8) = vl;
        }
    };
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
        if (!pok[ct]) {
#pragma unroll
            for (int mi = 0; mi < MIW; ++mi)
#pragma unroll
                for (int gi = 0; gi < 4; ++gi) {
                    _Float16* ph = image_ptr(mi, ct, gi);
                    const u32x2 z = {0u, 0u};
                    *reinterpret_cast<u32x2*>(ph) = z;
                    *reinterpret_cast<u32x2*>(ph + (size_t)PW * 8) = z;
                }
        }
    // image <- split(s * lrelu(x)), zero outside the sequence (the convolutions pad with zeros); s = power-of-two pre-scale
    auto xres_to_image = [&](float s) __attribute__((always_inline)) {
#pragma unroll
        for (int mi = 0; mi < MIW; ++mi)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int gi = 0; gi < 4; ++gi) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float t = xres[mi][ct][4 * gi + e] * s;
                        v[e] = fmaxf(t, t * 0.1f);
                    }
                    store_split(mi, ct, gi, v[0], v[1], v[2], v[3]);
                }
    };
    // Both convolutions' biases ride in the accumulators: the first MFMA of a tile takes `bias * factor` as its C operand (the factor undoes the
    // epilogue's power-of-two scale: exact).  The 32 * MI values come through the SCALAR cache (wave-uniform addresses; a lane picks its
    // half's four of every eight with v_cndmask): a vector load at this point queues behind the weight group that stage_first has just sent by
    // LDS-DMA — loads return in order — and the round-5 workgroup timeline showed ~1 us of that at each of the six places (the first MFMA of
    // conv2, the epilogue of conv1); scalar loads have their own path and counter, and no vector register is held across an epilogue.
    const int mi0u = __builtin_amdgcn_readfirstlane(mi0);
    // the 16 initial values of row tile mi's accumulators, built where the first MFMA of that row tile is issued (16 registers live for CT MFMAs)
    auto bias_tile = [&](const float* bias, float factor, int mi) __attribute__((always_inline)) -> f32x16 {
        f32x16 c;
#pragma unroll
        for (int gi = 0; gi < 4; ++gi)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float lo = bias[32 * (mi0u + mi) + 8 * gi + e], hi = bias[32 * (mi0u + mi) + 8 * gi + 4 + e];
                c[4 * gi + e] = (half ? hi : lo) * factor;
            }
        return c;
    };
    // acc = sum over taps j and 16-channel chunks c of  W[j][c] x image[c][column + (j - (K-1)/2) * d]   (three split products).
    // Precondition: weight group 0 of `w` sits in slot 0, published by a barrier.  Ends with a barrier (every wave has
    // finished reading the image and the weight slots).
    // c0: initial value of every accumulator tile of row tile mi — the first MFMA of a tile reads it as its C operand, so a per-channel
    // constant (the bias, pre-divided by the epilogue factor) joins the sum without a single extra instruction
    auto conv = [&](const half8* w, auto dtag, int d, f32x16 (&acc)[MIW][CT], const float* bias, float bfac) __attribute__((always_inline)) {
        constexpr int D = decltype(dtag)::value;   // IL: the dilation (compile time); plain layout: unused (d is a run-time value)
        const half8* base = IL ? P + (size_t)(half * 2) * PW + MARGQ + qw + l31 : P + (size_t)(half * 2) * PW + MARG + colw + l31 - d * ((K - 1) / 2);
        // B fragment of step s (tap s / NCH, chunk s % NCH), plane pl (0 hi, 1 lo), column tile ct
        auto bfrag = [&](int s, int pl, int ct) __attribute__((always_inline)) -> const half8* {
            const int j = s / NCH, cn = s % NCH;
            if constexpr (IL) {
                const int o = ct + (j - H) * D;                   // column offset of the lane's tile-ct column
                const int ph = ((o % CT) + CT) % CT, qs = (o - ph) / CT;
                return base + (size_t)(cn * 4 + pl) * PW + ph * PQ + qs;
            } else {
                return base + (size_t)(cn * 4 + pl) * PW + j * d + ct * 32;
            }
        };
        // Software pipeline, written out and pinned with scheduling barriers because hipcc will not build it (it sinks
        // every LDS read to just before its first use): the activation fragments of step s+1 are read between the MFMAs
        // of step s; the weight fragments of the group's second step are read during its first.
        half8 Af[2][MIW][2];
        half8 Bf[2][2][CT];
        auto readA = [&](half8 (&Aq)[MIW][2], int s) __attribute__((always_inline)) {
            const half8* ap = Aw + ((s / GRP) % NSLOT) * GRP_ITEMS + (s % GRP) * STEP_ITEMS + mi0 * 128 + lane;
#pragma unroll
            for (int mi = 0; mi < MIW; ++mi) {
                Aq[mi][0] = ap[(mi * 2 + 0) * 64];
                Aq[mi][1] = ap[(mi * 2 + 1) * 64];
            }
        };
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            Bf[0][0][ct] = *bfrag(0, 0, ct);
            Bf[0][1][ct] = *bfrag(0, 1, ct);
        }
        constexpr int NM = 3 * MIW * CT;   // MFMAs per step (and wave)
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (s % GRP == 0) {
                // group g + AHEAD leaves now, behind this group's MFMAs: its slot was read last by group g - 1 (retired by the previous barrier)
                if (s / GRP + AHEAD < NGRP) stage_group(w, s / GRP + AHEAD, (s / GRP + AHEAD) % NSLOT);
                if (s == 0 || NSLOT < 3) readA(Af[s & 1], s);   // (with three slots the next group was read ahead during the previous step)
            }
            __builtin_amdgcn_sched_barrier(0);
            // issue order of a step, pinned: (MFMA, one LDS read for the next step) pairs, then the rest of the MFMAs.
            // Term order lo_w*hi_x, hi_w*lo_x, hi_w*hi_x; consecutive MFMAs go to different accumulators.
            f32x16 c0;
#pragma unroll
            for (int q = 0; q < NM; ++q) {
                const int term = q / (MIW * CT), mi = (q / CT) % MIW, ct = q % CT;
                if (s == 0 && term == 0 && ct == 0) c0 = bias_tile(bias, bfac, mi);
                acc[mi][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Af[s & 1][mi][term == 0 ? 1 : 0], Bf[s & 1][term == 1 ? 1 : 0][ct],
                                                                   (s == 0 && term == 0) ? c0 : acc[mi][ct], 0, 0, 0);
                if (q < 2 * CT && s + 1 < NS) Bf[(s + 1) & 1][q / CT][q % CT] = *bfrag(s + 1, q / CT, q % CT);
                // weights of the next step: same group, or (three slots) the next group, published one barrier ago
                if (q >= 2 * CT && q < 2 * CT + 2 * MIW && s + 1 < NS && ((s + 1) % GRP != 0 || NSLOT >= 3)) {
                    const int i = q - 2 * CT;
                    Af[(s + 1) & 1][i >> 1][i & 1] = Aw[(((s + 1) / GRP) % NSLOT) * GRP_ITEMS + ((s + 1) % GRP) * STEP_ITEMS + mi0 * 128 + i * 64 + lane];
                }
                if (q < 2 * CT + 2 * MIW) __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (((s + 1) % GRP == 0 || s + 1 == NS) && !TTSC_DBG(a, 2)) __syncthreads();   // publishes the next weight group, retires this one
        }
    };

    xres_to_image(a.xs[0]);
    TTSC_STAMP(a, wg_lin, 18);               // image written by this wave
    if (!TTSC_DBG(a, 2)) __syncthreads();
    TTSC_STAMP(a, wg_lin, 1);
    for (int p = 0; p < a.npairs; ++p) {
        f32x16 acc[MIW][CT];
        const float bf1 = a.bs1[p] / a.us1[p];
        if constexpr (IL) {                    // (ends with a barrier: the image may be overwritten in place)
            if (a.d1[p] == 1) conv(a.w1[p], IntTag<1>(), 1, acc, a.b1[p], bf1);
            else if (a.d1[p] == 3) conv(a.w1[p], IntTag<3>(), 3, acc, a.b1[p], bf1);
            else conv(a.w1[p], IntTag<5>(), 5, acc, a.b1[p], bf1);
        } else {
            conv(a.w1[p], IntTag<0>(), a.d1[p], acc, a.b1[p], bf1);
        }
        TTSC_STAMP(a, wg_lin, 2 + 4 * p);
        stage_first(a.w2[p]);                  // conv2's first weight group(s) travel while the epilogue runs
        {
            const float us = a.us1[p];
#pragma unroll
            for (int mi = 0; mi < MIW; ++mi)
#pragma unroll
                for (int gi = 0; gi < 4; ++gi) {
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float t = acc[mi][ct][4 * gi + e] * us;
                            v[e] = fmaxf(t, t * 0.1f);
                        }
                        store_split(mi, ct, gi, v[0], v[1], v[2], v[3]);
                    }
                }
        }
        if (!TTSC_DBG(a, 2)) __syncthreads();
        TTSC_STAMP(a, wg_lin, 3 + 4 * p);
        conv(a.w2[p], IntTag<1>(), 1, acc, a.b2[p], 1.f / a.us2[p]);   // (the residual add is the epilogue's fma)
        TTSC_STAMP(a, wg_lin, 4 + 4 * p);
        if (p + 1 < a.npairs) stage_first(a.w1[p + 1]);
        {
            const float us = a.us2[p];
#pragma unroll
            for (int mi = 0; mi < MIW; ++mi)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int r = 0; r < 16; ++r) xres[mi][ct][r] = __builtin_fmaf(acc[mi][ct][r], us, xres[mi][ct][r]);
        }
        if (p + 1 < a.npairs) {
            xres_to_image(a.xs[p + 1]);
            if (!TTSC_DBG(a, 2)) __syncthreads();
        }
        TTSC_STAMP(a, wg_lin, 5 + 4 * p);
    }

    // store the nto central columns (all loads of a 32x32 tile before its stores)
    if (TTSC_DBG(a, 4)) {
        float t = 0.f;
#pragma unroll
        for (int mi = 0; mi < MIW; ++mi)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) t += xres[mi][ct][0] + xres[mi][ct][9];
        if (t == 12345.678f) a.y[0] = 1.f;
        return;
    }
    // The lane geometry of the final store is derived AFRESH from the thread index (made opaque so that it is not merged with the copies at the
    // top of the kernel): kept live across the whole chain these values were the last registers in scratch (one store before the pair loop, one
    // reload after it) of the 8-wave variants.
    int tid_f = threadIdx.x;
    asm volatile("" : "+v"(tid_f));
    const int lane_f = tid_f & 63, wv_f = tid_f >> 6;
    const int half_f = lane_f >> 5, l31_f = lane_f & 31;
    const int mi0_f = (wv_f / WN) * MIW;
    const int colw_f = (wv_f % WN) * (CT * 32);
    const int pos_w_f = q0 - a.halo + colw_f + (IL ? CT * l31_f : l31_f);
    float* yb = a.y + (size_t)b * C * a.L;
    if constexpr (POST) {
        // ---- conv_post + activation on the tile (host guarantees the vector path: L and all lengths multiples of CT) ----
        // block sum of the lane's four columns (one group further out than the stored range on each side: conv_post's taps reach 3 columns)
        constexpr int FOFF = 3, PWF = NCOL + 8;   // fp32 staging [C][PWF] in the (now idle) image area, tile column c at index c + FOFF
        static_assert((size_t)C * PWF * 4 <= (size_t)NG * 2 * PW * 16, "conv_post staging exceeds the image area");
        float* F = reinterpret_cast<float*>(smem_raw);
        const int col0 = colw_f + CT * l31_f;
        const bool need = col0 >= a.halo - CT && col0 + CT <= a.halo + a.nto + CT;
        const bool inside = pos_w_f >= 0 && pos_w_f + CT <= lin;
        const unsigned voff = (unsigned)(4 * half_f * a.L + (inside ? pos_w_f : 0));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ch0 = (r & 3) + 8 * (r >> 2);
            fvecT yv = fvecT(0.f);
            if (a.accumulate && need && inside) yv = *reinterpret_cast<const fvecT*>(yb + (size_t)ch0 * a.L + voff);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                float t = (xres[0][ct][r] + yv[ct]) * a.post_in_scale;
                t = fmaxf(t, t * a.post_slope);
                F[(size_t)(ch0 + 4 * half_f) * PWF + FOFF + col0 + ct] = inside ? t : 0.f;
            }
        }
        __syncthreads();
        if (4 * tid_f < a.nto) {
            const float* win = F + FOFF + a.halo - 3 + 4 * tid_f;   // 16-byte aligned: halo % 4 == 0
            float acc4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
            for (int ci = 0; ci < C; ++ci) {
                float w[12];
#pragma unroll
                for (int v4 = 0; v4 < 3; ++v4) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(win + (size_t)ci * PWF + 4 * v4);
                    w[4 * v4] = t[0]; w[4 * v4 + 1] = t[1]; w[4 * v4 + 2] = t[2]; w[4 * v4 + 3] = t[3];
                }
                const float* wk = a.wpost + ci * 7;
#pragma unroll
                for (int j = 0; j < 7; ++j) {
                    const float wj = wk[j];
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc4[e] = fmaf(wj, w[e + j], acc4[e]);
                }
            }
            const float bv = a.bpost ? a.bpost[0] : 0.f;
            float* wb = a.wav + (size_t)b * a.L;
            const int q = q0 + 4 * tid_f;
            float res[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) res[e] = apply_act((acc4[e] + bv + 0.f) * a.post_out_scale, a.post_act) + 0.f;
            if (a.nf_flag) {
                bool bad = false;
#pragma unroll
                for (int e = 0; e < 4; ++e) bad = bad || (q + e < a.L && !(fabsf(res[e]) <= 3.0e38f));
                if (bad) atomicOr(a.nf_flag, 1u);
            }
            if (q + 3 < a.L && (((uintptr_t)(wb + q)) & 15) == 0) {
                const f32x4 o = {res[0], res[1], res[2], res[3]};
                *reinterpret_cast<f32x4*>(wb + q) = o;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (q + e < a.L) wb[q + e] = res[e];
            }
        }
        return;
    }
    if (IL && vec) {
        // the lane's CT columns of a channel are CT consecutive samples of the row: one vector access per channel
        const int col0 = colw_f + CT * l31_f;
        const bool ok = col0 >= a.halo && col0 + CT <= a.halo + a.nto && pos_w_f + CT <= lin;
        const unsigned voff = (unsigned)(4 * half_f * a.L + (ok ? pos_w_f : 0));
#pragma unroll
        for (int mi = 0; mi < MIW; ++mi) {
            fvecT yv[16];
            if (a.accumulate) {
#pragma unroll
                for (int r = 0; r < 16; ++r) yv[r] = *reinterpret_cast<const fvecT*>(yb + (size_t)(32 * (mi0_f + mi) + (r & 3) + 8 * (r >> 2)) * a.L + voff);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) yv[r] = fvecT(0.f);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                fvecT o;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) o[ct] = xres[mi][ct][r] + yv[r][ct];
                if (ok) *reinterpret_cast<fvecT*>(yb + (size_t)(32 * (mi0_f + mi) + (r & 3) + 8 * (r >> 2)) * a.L + voff) = o;
            }
        }
#ifdef TTSC_ABLATE
        if (a.prof) {
            __builtin_amdgcn_s_waitcnt(0);
            TTSC_STAMP(a, wg_lin, 14);
        }
#endif
        return;
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int col = colw_f + (IL ? CT * l31_f + ct : ct * 32 + l31_f);
        const int pos = q0 - a.halo + col;
        const bool ok = col >= a.halo && col < a.halo + a.nto && pos < lin;
        const int pc = ok ? pos : 0;
#pragma unroll
        for (int mi = 0; mi < MIW; ++mi) {
            float yv[16];
            if (a.accumulate) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ch = 32 * (mi0_f + mi) + (r & 3) + 8 * (r >> 2) + 4 * half_f;
                    yv[r] = yb[(size_t)ch * a.L + pc];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) yv[r] = 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ch = 32 * (mi0_f + mi) + (r & 3) + 8 * (r >> 2) + 4 * half_f;
                if (ok) yb[(size_t)ch * a.L + pc] = xres[mi][ct][r] + yv[r];
            }
        }
    }
#ifdef TTSC_ABLATE
    if (a.prof) {
        __builtin_amdgcn_s_waitcnt(0);
        TTSC_STAMP(a, wg_lin, 14);
    }
#endif
}

#ifdef TTSC_RB_PROBE   // development: compile ONE instantiation (tools/isa_stats.py -DTTSC_RB_PROBE=1,11,4,4,2)
template __global__ void rbchain_f16x3_kernel<TTSC_RB_PROBE>(ChainArgs);
}  // namespace ttsc
#else
template <int MI, int K, int CT, int NW, int WPS, int GRP = 2, int NSLOT = 2, int WM = 1, bool IL = false, bool POST = false>
static int launch_chain(ChainArgs& a, int B, hipStream_t s) {
    constexpr int NCOL = (NW / WM) * CT * 32;
    constexpr int PW = IL ? CT * (NCOL / CT + 2 * ChainGeo<K, CT>::MARGQ) : NCOL + 2 * ChainGeo<K, CT>::MARG;
    constexpr size_t lds = (size_t)(4 * MI) * 2 * PW * 16 + (size_t)NSLOT * GRP * (MI * 2 * 64) * 16;   // image + weight ring
    static_assert(lds <= 160 * 1024, "activation image exceeds the LDS");
    if (POST) a.halo = (a.halo + 3 + CT - 1) / CT * CT;   // + conv_post's three columns, kept a multiple of the vector width
    a.nto = NCOL - 2 * a.halo;
    if (IL && a.nto > 256) a.nto &= ~31;   // tile stores start on 128-byte boundaries of the row
    TTSC_REQUIRE(a.nto >= 64, "rbchain: halo %d leaves no output columns in a %d-column tile", a.halo, NCOL);
    if (int rc = ensure_full_lds((const void*)rbchain_f16x3_kernel<MI, K, CT, NW, WPS, GRP, NSLOT, WM, IL, POST>)) return rc;   // once per (device, kernel)
    dim3 grid((unsigned)ceil_div(a.L, a.nto), (unsigned)B);
    hipLaunchKernelGGL((rbchain_f16x3_kernel<MI, K, CT, NW, WPS, GRP, NSLOT, WM, IL, POST>), grid, dim3(64 * NW), lds, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("rbchain_f16x3_kernel launch failed: %s", hipGetErrorString(e));
        return TTSC_EHIP;
    }
    return TTSC_OK;
}

// tile shapes: 0 = 4 waves x 128 columns (two workgroups per CU at 32 channels), 1 = 8 waves x 128 columns (32 channels) /
// 8 waves x 64 columns (64 channels), one workgroup per CU
template <int MI, int K>
static int launch_chain_k(ChainArgs& a, int B, int shape, hipStream_t s) {
    if constexpr (MI == 4) {   // 128 channels: 2 x 4 waves of (64 channels x 64 columns), the whole LDS (image 136 KiB + three 8-KiB weight slots)
        if (shape >= 10) return launch_chain<4, K, 2, 8, 2, 1, 3, 2, true>(a, B, s);
        return launch_chain<4, K, 2, 8, 2, 1, 3, 2>(a, B, s);
    }
    // shapes 2 .. 4: experiments with longer weight groups (fewer barriers per convolution) and 768-column tiles;
    // shapes 10 / 11: the small / large tile with interleaved columns (vector loads and stores of the tile); 12: 11 with 6-step groups
    constexpr int G768 = K == 3 ? 6 : (K == 7 ? 7 : 11), G768W = K == 3 ? 6 : (K == 7 ? 14 : 11);
    if constexpr (MI == 1) {
        if (shape == 22) return launch_chain<1, K, 4, 8, 2, 6, 2, 1, true, true>(a, B, s);   // (internal) + conv_post epilogue
        // (round 5, measured and dropped: 256-column tiles with three / four workgroups per CU for the K = 3 block — 1.81 / 1.76 ms against 1.65 ms: the block
        // is bound by how fast a CU can fill a tile (~13.5 GB/s per CU: ~64 outstanding 128-byte lines x ~600 ns), not by how many workgroups take turns)
        if (shape == 12) return launch_chain<1, K, 4, 8, 2, 6, 2, 1, true>(a, B, s);
        if (shape == 11) return launch_chain<1, K, 4, 8, 2, 2, 2, 1, true>(a, B, s);
        if (shape == 10) return launch_chain<1, K, 4, 4, 2, 2, 2, 1, true>(a, B, s);
        if (shape == 4) return launch_chain<1, K, 3, 8, 2, G768W, 2>(a, B, s);
        if (shape == 3) return launch_chain<1, K, 3, 8, 2, G768, 2>(a, B, s);
        if (shape == 2) return launch_chain<1, K, 4, 8, 2, 6, 2>(a, B, s);
        if (shape == 1) return launch_chain<1, K, 4, 8, 2>(a, B, s);
        return launch_chain<1, K, 4, 4, 2>(a, B, s);
    }
    if constexpr (MI == 2 && K == 3) {
        if (shape == 2) return launch_chain<2, K, 3, 8, 2, 6, 2, 2>(a, B, s);
    }
    if (shape == 11 || shape == 12) return launch_chain<2, K, 2, 8, 2, 2, 2, 1, true>(a, B, s);
    if (shape == 10) return launch_chain<2, K, 2, 4, 2, 2, 2, 1, true>(a, B, s);
    if (shape == 1) return launch_chain<2, K, 2, 8, 2>(a, B, s);
    return launch_chain<2, K, 2, 4, 2>(a, B, s);
}

}  // namespace ttsc

using namespace ttsc;

static int chain_pair_ok(const ttsc_conv1d* c1, const ttsc_conv1d* c2, int C, int k) {
    if (!c1 || !c2) return 0;
    const auto &g1 = c1->cfg, &g2 = c2->cfg;
    if (g1.transposed || g2.transposed) return 0;
    if (g1.in_channels != C || g1.out_channels != C || g2.in_channels != C || g2.out_channels != C) return 0;
    if (g1.kernel_size != k || g2.kernel_size != k) return 0;
    if (g2.dilation != 1 || g2.padding != (k - 1) / 2 || g1.padding != g1.dilation * (k - 1) / 2) return 0;
    if (g1.dilation < 1 || g1.dilation > 5) return 0;
    if (c1->precision != TTSC_PREC_F16X3 || c2->precision != TTSC_PREC_F16X3) return 0;
    if (!c1->has_weight || !c2->has_weight || !c1->bias_dev || !c2->bias_dev || c1->dev_weights || c2->dev_weights) return 0;
    if (c1->phases.size() != 1 || c2->phases.size() != 1 || !c1->phases[0].wph_dev || !c2->phases[0].wph_dev) return 0;
    return 1;
}

extern "C" int ttsc_rbchain_supported(const ttsc_conv1d* const* convs1, const ttsc_conv1d* const* convs2, int32_t npairs) {
    if (!convs1 || !convs2 || npairs < 1 || npairs > RB_MAXP || !convs1[0]) return 0;
    const int C = convs1[0]->cfg.in_channels, k = convs1[0]->cfg.kernel_size;
    if (!(C == 32 || C == 64 || C == 128) || !(k == 3 || k == 7 || k == 11)) return 0;
    if (C == 128 && k != 3) return 0;   // the 128-channel image + margins only fits the LDS with K = 3 (halo 24 of 256 columns)
    int halo = 0;
    for (int p = 0; p < npairs; ++p) {
        if (!chain_pair_ok(convs1[p], convs2[p], C, k)) return 0;
        halo += (convs1[p]->cfg.dilation + 1) * (k - 1) / 2;
    }
    // the smallest tile (256 columns at 64 channels, 512 at 32) must keep a useful share of output columns
    const int ncol = C == 32 ? 512 : 256;   // (128 channels: the only tile is 256 columns)
    return ncol - 2 * halo >= ncol / 2 ? 1 : 0;
}

static int chain_forward(const ttsc_conv1d* const* convs1, const ttsc_conv1d* const* convs2, int32_t npairs, const float* x, int32_t B, int64_t L, float* y,
                         int32_t accumulate, const int32_t* len_dev, int32_t shape, void* stream, const ttsc_conv1d* post, const ttsc_conv1d_epilogue* post_ep,
                         float* wav);

extern "C" int ttsc_rbchain_forward(const ttsc_conv1d* const* convs1, const ttsc_conv1d* const* convs2, int32_t npairs, const float* x,
                                    int32_t B, int64_t L, float* y, int32_t accumulate, const int32_t* len_dev, int32_t shape,
                                    void* stream) {
    TTSC_REQUIRE(convs1 && convs2 && x && y, "ttsc_rbchain_forward: null argument");
    return chain_forward(convs1, convs2, npairs, x, B, L, y, accumulate, len_dev, shape, stream, nullptr, nullptr, nullptr);
}

// 1 when ttsc_rbchain_post_forward takes these layers: a 32-channel chain with dilations in {1, 3, 5} followed by conv_post (32 -> 1, k = 7, padding 3,
// host-set weights); L must be a multiple of 4 (checked by the forward)
extern "C" int ttsc_rbchain_post_supported(const ttsc_conv1d* const* convs1, const ttsc_conv1d* const* convs2, int32_t npairs, const ttsc_conv1d* post) {
    if (!post || !ttsc_rbchain_supported(convs1, convs2, npairs)) return 0;
    if (convs1[0]->cfg.in_channels != 32) return 0;
    for (int p = 0; p < npairs; ++p) {
        const int d = convs1[p]->cfg.dilation;
        if (!(d == 1 || d == 3 || d == 5)) return 0;
    }
    const auto& g = post->cfg;
    return !g.transposed && post->groups == 1 && g.in_channels == 32 && g.out_channels == 1 && g.kernel_size == 7 && g.dilation == 1 && g.padding == 3 &&
           g.stride == 1 && post->w_plain_dev && !post->dev_weights;
}

extern "C" int ttsc_rbchain_post_forward(const ttsc_conv1d* const* convs1, const ttsc_conv1d* const* convs2, int32_t npairs, const float* x, int32_t B,
                                         int64_t L, const float* ysum, const ttsc_conv1d* post, const ttsc_conv1d_epilogue* post_ep, float* wav,
                                         const int32_t* len_dev, void* stream) {
    TTSC_REQUIRE(convs1 && convs2 && x && post && wav, "ttsc_rbchain_post_forward: null argument");
    TTSC_REQUIRE(ttsc_rbchain_post_supported(convs1, convs2, npairs, post), "ttsc_rbchain_post_forward: these layers are not eligible");
    TTSC_REQUIRE(L % 4 == 0 && ((uintptr_t)x % 16 == 0) && (!ysum || (uintptr_t)ysum % 16 == 0),
                 "ttsc_rbchain_post_forward: L must be a multiple of 4 and x / ysum 16-byte aligned (got L = %lld)", (long long)L);
    TTSC_REQUIRE(!(post_ep && (post_ep->accumulate || post_ep->gate_dev)), "ttsc_rbchain_post_forward: conv_post epilogue options not available when fused");
    return chain_forward(convs1, convs2, npairs, x, B, L, const_cast<float*>(ysum), ysum ? 1 : 0, len_dev, 22, stream, post, post_ep, wav);
}

static int chain_forward(const ttsc_conv1d* const* convs1, const ttsc_conv1d* const* convs2, int32_t npairs, const float* x, int32_t B, int64_t L, float* y,
                         int32_t accumulate, const int32_t* len_dev, int32_t shape, void* stream, const ttsc_conv1d* post, const ttsc_conv1d_epilogue* post_ep,
                         float* wav) {
    TTSC_REQUIRE(ttsc_rbchain_supported(convs1, convs2, npairs), "ttsc_rbchain_forward: these layers are not eligible for the fused chain");
    TTSC_REQUIRE(x != y, "ttsc_rbchain_forward: y must not alias x");
    TTSC_REQUIRE(B > 0 && L > 0 && L < (1ll << 30), "ttsc_rbchain_forward: bad B/L");
    ChainArgs a;
    memset(&a, 0, sizeof(a));
    if (post) {
        a.wav = wav;
        a.wpost = post->w_plain_dev;
        a.bpost = post->bias_dev;
        a.post_in_scale = post_ep ? post_ep->in_scale : 1.f;
        a.post_slope = post_ep ? post_ep->in_slope : 1.f;
        a.post_out_scale = post_ep ? post_ep->out_scale : 1.f;
        a.post_act = post_ep ? post_ep->out_act : TTSC_ACT_NONE;
        a.nf_flag = post->nf_flag;
    }
    a.x = x;
    a.y = y;
    a.len = len_dev;
    a.L = (int)L;
    a.npairs = npairs;
    a.accumulate = accumulate;
#ifdef TTSC_ABLATE
    if (const char* ev = getenv("TTSC_CHAIN_DBG")) a.dbg = atoi(ev);
    if (const char* ev = getenv("TTSC_PROF_PTR")) a.prof = reinterpret_cast<unsigned long long*>(strtoull(ev, nullptr, 0));
#endif
    const int C = convs1[0]->cfg.in_channels, k = convs1[0]->cfg.kernel_size;
    for (int p = 0; p < npairs; ++p) {
        a.w1[p] = reinterpret_cast<const half8*>(convs1[p]->phases[0].wph_dev);
        a.w2[p] = reinterpret_cast<const half8*>(convs2[p]->phases[0].wph_dev);
        a.b1[p] = convs1[p]->bias_dev;
        a.b2[p] = convs2[p]->bias_dev;
        // activation pre-scales of the two layers (ttsc_conv1d_set_activation_scale): s1 multiplies lrelu(x) on its way into the
        // image, conv1's accumulator comes out s1 too large, conv2's input is written s2 * lrelu(t), conv2's accumulator s2 too large
        const float s1 = convs1[p]->act_scale, s2 = convs2[p]->act_scale;
        a.xs[p] = s1;
        a.us1[p] = convs1[p]->w_unscale * (s2 / s1);
        a.bs1[p] = s2;
        a.us2[p] = convs2[p]->w_unscale / s2;
        a.d1[p] = convs1[p]->cfg.dilation;
        a.halo += (a.d1[p] + 1) * (k - 1) / 2;
    }
    hipStream_t s = (hipStream_t)stream;
    bool auto_shape = false;
    if (shape < 0) {
        // default: the big tile once the halo would eat more than ~1/6 of the small one; at 64 channels the big tile also wins
        // for K = 3 (measured 1.54 vs 1.81 ms per ResBlock at config[1]) as long as it still fills the chip
        const int small = C == 32 ? 512 : 256;
        shape = (2 * a.halo * 6 > small) ? 1 : 0;
        if (C == 64 && (int64_t)B * ceil_div(L, 2 * small - 2 * a.halo) >= 256) shape = 1;
        auto_shape = true;
    }
    // interleaved columns (vector loads / stores of the tile) for the two standard shapes unless TTSC_CHAIN_IL=0; needs dilations 1 / 3 / 5
    const char* il_ev = getenv("TTSC_CHAIN_IL");
    const int il_env = il_ev ? atoi(il_ev) : 1;
    bool il_ok = true;
    for (int p = 0; p < npairs; ++p) il_ok = il_ok && (a.d1[p] == 1 || a.d1[p] == 3 || a.d1[p] == 5);
    if (shape >= 10 && shape < 20 && !il_ok) shape -= 10;
    // (measured at config[1], round 4, y += chain(x): K = 3 small tile 1.84 ms interleaved vs 1.92 plain; K = 7 small tile 3.43 plain vs 3.50
    // interleaved vs 3.49 large interleaved; K = 11 large tile 4.90 interleaved with 6-step weight groups vs 5.12 plain — the K = 7 block at
    // 32 channels keeps the plain small tile)
    if (auto_shape && C == 32 && k == 7 && shape == 0) il_ok = false;
    if (shape >= 0 && shape <= 1 && il_env && il_ok) shape += 10;
    if (auto_shape && shape == 11 && C == 32) shape = 12;   // large tile at 32 channels: 6-step weight groups fit beside the image (one barrier per 6 k-steps)
    if (shape == 12 && !(C == 32)) shape = 11;
    if (C == 128) return launch_chain_k<4, 3>(a, B, shape, s);
    if (C == 32) {
        if (k == 3) return launch_chain_k<1, 3>(a, B, shape, s);
        if (k == 7) return launch_chain_k<1, 7>(a, B, shape, s);
        return launch_chain_k<1, 11>(a, B, shape, s);
    }
    if (k == 3) return launch_chain_k<2, 3>(a, B, shape, s);
    if (k == 7) return launch_chain_k<2, 7>(a, B, shape, s);
    return launch_chain_k<2, 11>(a, B, shape, s);
}
#endif  // TTSC_RB_PROBE
