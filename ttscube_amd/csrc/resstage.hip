// One launch per generator STAGE of the narrow (32-channel) part of HiFi-GAN: the three ResBlock1 chains of a time tile back to
// back, their sum kept in registers, and — for the last stage — conv_post + tanh in the epilogue.  Split-precision (f16x3) MFMA, gfx950.
//
//     xs = 0;  for j in 0..2:  x = x0;  for p in 0..2:  x = x + conv2_jp( lrelu( conv1_jp( lrelu(x) ) ) );   xs += x
//     POST:  wav = tanh( conv_post( lrelu(xs / 3, 0.01) ) )           otherwise:  y = xs
//
// (hifigan.models.Generator.forward [EXTERNAL]: `xs += resblocks[i*nk+j](x)`, `x = xs / nk`, `conv_post`, `tanh`; reference call
// sites cube/networks/cubegan.py:72,83,131 and cube/io_utils/runtime.py:78.)
//
// Why: as three rbchain launches + conv_post (resblock.hip, conv1d.hip) the stage reads x0 three times, read-modify-writes the
// block sum twice and reads it once more for conv_post — eight passes over a 1.57 GB tensor at BASELINE config[1] — and every one of
// those passes is an exposed prologue / epilogue of a workgroup that owns its whole CU (the image fills the LDS).  Here a tile's
// x0 comes from HBM once (the re-reads for the second and third chain hit L2), the block sum never leaves the registers and
// the waveform is the only thing written.
//
// Layout inside the workgroup (same scheme as rbchain_f16x3_kernel):
//   * residual stream `xres` and block sum `ysum` in REGISTERS in the MFMA C/D layout;
//   * ONE activation image in LDS, fp16 (hi, lo) planes [8-channel group][hi|lo][column], rewritten in place between convolutions;
//   * weights by LDS-DMA into a two-slot ring whose groups are HALF A CONVOLUTION OR A WHOLE ONE (K = 3: 6 steps, K = 7: 14 steps,
//     K = 11: 11 steps): at most one barrier inside a convolution instead of one every two k-steps, and the next group always has a
//     whole group's worth of MFMAs to land behind;
//   * tile = NW * CT * 32 columns; `lhalo` columns on the left and NCOL - lhalo - nto on the right are halo (>= the K = 11 chain's
//     60 + conv_post's 3).
#include "conv_internal.hpp"
#include "conv_kernels.hpp"

namespace ttsc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float2v __attribute__((ext_vector_type(2)));

constexpr int RS_NB = 3, RS_NP = 3;   // ResBlocks per stage, pairs per ResBlock

struct StageBlock {
    const half8* w1[RS_NP];   // f16x3 fragments [tap][C/16][C/32][hi|lo][64 lanes][8 half]
    const half8* w2[RS_NP];
    const float* b1[RS_NP];
    const float* b2[RS_NP];
    float us1[RS_NP], us2[RS_NP];   // conv1 -> image  t' = fma(acc, us1, b1 * bs1);  conv2 -> residual  x += fma(acc, us2, b2)
    float xs[RS_NP], bs1[RS_NP];    // activation pre-scales (powers of two)
    int d1[RS_NP];                  // dilation of conv1 (conv2 is undilated)
};

struct StageArgs {
    const float* x;      // [B, C, L] stage input (output of the upsampler)
    float* y;            // [B, C, L] sum of the three blocks (POST = false)
    float* wav;          // [B, 1, L] (POST = true)
    StageBlock blk[RS_NB];
    const float* wpost;  // conv_post weights, torch layout [1][C][7]
    const float* bpost;  // [1] or null
    float post_in_scale, post_slope, post_out_scale;
    int post_act;
    unsigned* nf_flag;   // range guard word (see conv_cout1_kernel) or null
    const int* len;      // [B] valid length or null
    int L, nto, lhalo;
};

// weight group = steps between two barriers of a convolution (a step = one (tap, 16-channel chunk) = MI * 2 KiB of fragments)
__host__ __device__ constexpr int rs_grp(int MI, int K) { return MI == 1 ? (K == 3 ? 6 : (K == 7 ? 14 : 11)) : 2; }

template <int MI_, int CT_, int NW_>
struct StageGeo {
    static constexpr int MI = MI_, CT = CT_, NW = NW_;
    static constexpr int C = 32 * MI, NCH = 2 * MI, NG = 4 * MI;
    static constexpr int NCOL = NW * CT * 32;
    static constexpr int MARG = 26;               // >= the largest tap offset of the stage (K = 11, dilation 5: 25)
    static constexpr int PW = NCOL + 2 * MARG;
    static constexpr int NTHR = 64 * NW;
    static constexpr int STEP_ITEMS = MI * 2 * 64;   // 16-byte items of one (tap, chunk) step
    static constexpr int IMG_ITEMS = NG * 2 * PW;
    // fp32 staging of the block sum for conv_post: [C][PWF] floats inside the image area, column c at index c + FOFF
    static constexpr int FOFF = 3, PWF = NCOL + 8;
};

struct StageCtx {
    half8* P;      // image
    half8* Aw;     // weight ring
    int lane, wv, half, l31, colw;
};

template <class G, int K>
__device__ __forceinline__ void rs_stage_group(const StageCtx& c, const half8* w, int g, int slot) {
    constexpr int GRP = rs_grp(G::MI, K), NS = K * G::NCH, GRP_ITEMS = GRP * G::STEP_ITEMS, NIMAX = GRP_ITEMS / 64;
    const int steps = NS - g * GRP < GRP ? NS - g * GRP : GRP;
    const int NI = steps * G::STEP_ITEMS / 64;
#pragma unroll
    for (int i = 0; i < (NIMAX + G::NW - 1) / G::NW; ++i) {
        const int blk = c.wv + i * G::NW;   // wave-uniform
        if (blk < NI)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w + (size_t)g * GRP_ITEMS + blk * 64 + c.lane),
                                             (__attribute__((address_space(3))) void*)(c.Aw + slot * GRP_ITEMS + blk * 64), 16, 0, 0);
    }
}

// four channels (one lane's share of 8-channel group 4*mi + gi) of one column -> (hi, lo) halves in the image; `inside`: the lane's column
// lies inside the sequence — columns outside are zeroed once by the kernel and never written again (same scheme as rbchain_f16x3_kernel)
template <class G>
__device__ __forceinline__ _Float16* rs_image_ptr(const StageCtx& c, int mi, int ct, int gi) {
    return reinterpret_cast<_Float16*>(c.P + (size_t)((mi * 4 + gi) * 2) * G::PW + G::MARG + c.colw + ct * 32 + c.l31) + 4 * c.half;
}
template <class G>
__device__ __forceinline__ void rs_store_split(const StageCtx& c, int mi, int ct, int gi, bool inside, float v0, float v1, float v2, float v3) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    unsigned h0, l0, h1, l1;
    split2_f16(v0, v1, h0, l0);
    split2_f16(v2, v3, h1, l1);
    const u32x2 vh = {h0, h1}, vl = {l0, l1};
    _Float16* ph = rs_image_ptr<G>(c, mi, ct, gi);
    if (inside) {
        *reinterpret_cast<u32x2*>(ph) = vh;
        *reinterpret_cast<u32x2*>(ph + (size_t)G::PW * 8) = vl;
    }
}

// image <- split(s * lrelu(x)), zero outside the sequence
template <class G>
__device__ __forceinline__ void rs_xres_to_image(const StageCtx& c, const f32x16 (&xres)[G::MI][G::CT], const bool (&pok)[G::CT], float s) {
#pragma unroll
    for (int mi = 0; mi < G::MI; ++mi)
#pragma unroll
        for (int ct = 0; ct < G::CT; ++ct)
#pragma unroll
            for (int gi = 0; gi < 4; ++gi) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = xres[mi][ct][4 * gi + e] * s;
                    v[e] = fmaxf(t, t * 0.1f);
                }
                rs_store_split<G>(c, mi, ct, gi, pok[ct], v[0], v[1], v[2], v[3]);
            }
}

// acc = sum over taps j and 16-channel chunks of  W[j][chunk] x image[chunk][column + (j - (K-1)/2) * d]   (three split products).
// Weight groups of this convolution occupy ring slots (SLOT0 + g) & 1.  Precondition: group 0 sits in slot SLOT0, published by a
// barrier.  `wnext` (or null): the convolution that follows in the same chain — its group 0 leaves for slot (SLOT0 + NGRP) & 1 when this
// convolution's last group starts.  Ends with a barrier (every wave has finished reading the image and the weight slots).
template <class G, int K, int SLOT0>
__device__ __forceinline__ void rs_conv(const StageCtx& c, const half8* w, const half8* wnext, int d, f32x16 (&acc)[G::MI][G::CT], const f32x16* c0) {
    constexpr int MI = G::MI, CT = G::CT, NCH = G::NCH, PW = G::PW;
    constexpr int GRP = rs_grp(MI, K), NS = K * NCH, NGRP = (NS + GRP - 1) / GRP, GRP_ITEMS = GRP * G::STEP_ITEMS;
    if (!c0) {   // (c0: initial accumulator value per row tile, read by the first MFMA of every tile as its C operand — conv2's bias)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ct][r] = 0.f;
    }
    const half8* base = c.P + (size_t)(c.half * 2) * PW + G::MARG + c.colw + c.l31 - d * ((K - 1) / 2);
    half8 Af[2][MI][2];
    half8 Bh[2][CT];   // B fragments, hi plane: used by the first and the last product of a step -> double-buffered
    half8 Bl[CT];      // lo plane: used by the middle product only -> refilled in place right behind its MFMA
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        Bh[0][ct] = base[ct * 32];
        Bl[ct] = base[PW + ct * 32];
    }
    constexpr int NM = 3 * MI * CT;   // MFMAs per step (and wave)
    static_assert(MI == 1, "the in-place refill of the lo plane assumes one row tile per wave");
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int g = s / GRP, slot = (SLOT0 + g) & 1;
        if (s % GRP == 0) {
            // the next group leaves now, behind this group's MFMAs: its slot was read last by the previous group (retired by the barrier before this one)
            if (g + 1 < NGRP) rs_stage_group<G, K>(c, w, g + 1, slot ^ 1);
            else if (wnext) rs_stage_group<G, K>(c, wnext, 0, slot ^ 1);
            const half8* ap = c.Aw + slot * GRP_ITEMS + c.lane;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                Af[s & 1][mi][0] = ap[(mi * 2 + 0) * 64];
                Af[s & 1][mi][1] = ap[(mi * 2 + 1) * 64];
            }
        }
        const int jn = (s + 1) / NCH, cn = (s + 1) % NCH;
        const half8* bpn = base + (size_t)(cn * 4) * PW + jn * d;
        __builtin_amdgcn_sched_barrier(0);
        // issue order of a step, pinned: (MFMA, one LDS read for the next step) pairs.  Term order lo_w*hi_x, hi_w*lo_x, hi_w*hi_x (the same
        // as rbchain_f16x3_kernel: bit-identical sums); consecutive MFMAs go to different accumulators.
#pragma unroll
        for (int q = 0; q < NM; ++q) {
            const int term = q / CT, ct = q % CT;
            acc[0][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Af[s & 1][0][term == 0 ? 1 : 0], term == 1 ? Bl[ct] : Bh[s & 1][ct],
                                                              (s == 0 && term == 0 && c0) ? c0[0] : acc[0][ct], 0, 0, 0);
            if (s + 1 < NS) {
                if (term == 0) Bh[(s + 1) & 1][ct] = bpn[ct * 32];
                if (term == 1) Bl[ct] = bpn[PW + ct * 32];
                if (term == 2 && ct < 2 && (s + 1) % GRP != 0)   // weights of the next step of the same group
                    Af[(s + 1) & 1][0][ct] = c.Aw[slot * GRP_ITEMS + ((s + 1) % GRP) * G::STEP_ITEMS + ct * 64 + c.lane];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if ((s + 1) % GRP == 0 || s + 1 == NS) __syncthreads();   // publishes the next weight group, retires this one
    }
}

// one ResBlock1: three (conv1, conv2) pairs on the tile; xres is the residual stream (in: x0, out: the block's output)
template <class G, int K>
__device__ __forceinline__ void rs_block(const StageCtx& c, const StageBlock& bk, f32x16 (&xres)[G::MI][G::CT], const bool (&pok)[G::CT]) {
    constexpr int MI = G::MI, CT = G::CT;
    constexpr int NGRP = (K * G::NCH + rs_grp(MI, K) - 1) / rs_grp(MI, K);
    constexpr int SLOT1 = NGRP & 1;   // first slot of conv2 (conv1 starts in slot 0; two convolutions use an even number of groups)
    rs_xres_to_image<G>(c, xres, pok, bk.xs[0]);
    __syncthreads();    // (also publishes group 0 of w1[0], sent by the caller)
    for (int p = 0; p < RS_NP; ++p) {
        f32x16 acc[MI][CT];
        rs_conv<G, K, 0>(c, bk.w1[p], bk.w2[p], bk.d1[p], acc, nullptr);
        {
            const float us = bk.us1[p];
            const float* bias = bk.b1[p];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int gi = 0; gi < 4; ++gi) {
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + 32 * mi + 8 * gi + 4 * c.half) * bk.bs1[p];
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float t = __builtin_fmaf(acc[mi][ct][4 * gi + e], us, bv[e]);
                            v[e] = fmaxf(t, t * 0.1f);
                        }
                        rs_store_split<G>(c, mi, ct, gi, pok[ct], v[0], v[1], v[2], v[3]);
                    }
                }
        }
        __syncthreads();
        {
            f32x16 bias_c[MI];
            const float inv_us = 1.f / bk.us2[p];
            const float* bias = bk.b2[p];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int gi = 0; gi < 4; ++gi) {
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + 32 * mi + 8 * gi + 4 * c.half);
#pragma unroll
                    for (int e = 0; e < 4; ++e) bias_c[mi][4 * gi + e] = bv[e] * inv_us;
                }
            rs_conv<G, K, SLOT1>(c, bk.w2[p], p + 1 < RS_NP ? bk.w1[p + 1] : nullptr, 1, acc, bias_c);
        }
        {
            const float us = bk.us2[p];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int r = 0; r < 16; ++r) xres[mi][ct][r] = __builtin_fmaf(acc[mi][ct][r], us, xres[mi][ct][r]);
        }
        if (p + 1 < RS_NP) {
            rs_xres_to_image<G>(c, xres, pok, bk.xs[p + 1]);
            __syncthreads();
        }
    }
}

// x0 tile -> residual registers.  Addressing: wave-uniform row pointer (channel (r & 3) + 8 * (r >> 2) of row tile mi) + ONE 32-bit lane
// offset per column tile (the lane's half selects channel + 4): 16 scalar bases and CT vector offsets instead of 16 * CT 64-bit addresses.
template <class G>
__device__ __forceinline__ void rs_load_x(const float* xb, int L, int lin, int pos_w, int half, f32x16 (&xres)[G::MI][G::CT]) {
#pragma unroll
    for (int ct = 0; ct < G::CT; ++ct) {
        const int pos = pos_w + ct * 32;
        int pc = pos < lin - 1 ? pos : lin - 1;
        pc = pc < 0 ? 0 : pc;
        const unsigned voff = (unsigned)(4 * half * L + pc);
#pragma unroll
        for (int mi = 0; mi < G::MI; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float* row = xb + (size_t)(32 * mi + (r & 3) + 8 * (r >> 2)) * L;
                xres[mi][ct][r] = row[voff];
            }
    }
}

template <int MI, int CT, int NW, int K0, int K1, int K2, bool POST>
__global__ __launch_bounds__(64 * NW, NW / 4) void rbstage_f16x3_kernel(StageArgs a) {
    using G = StageGeo<MI, CT, NW>;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    StageCtx c;
    c.P = reinterpret_cast<half8*>(smem_raw);
    c.Aw = c.P + G::IMG_ITEMS;
    const int tid = threadIdx.x;
    c.lane = tid & 63;
    c.wv = tid >> 6;
    c.half = c.lane >> 5;
    c.l31 = c.lane & 31;
    c.colw = c.wv * (CT * 32);
    const int b = blockIdx.y;
    const int q0 = blockIdx.x * a.nto;
    const int lin = a.len ? a.len[b] : a.L;
    if (q0 >= lin) return;
    const int pos_w = q0 - a.lhalo + c.colw + c.l31;   // sequence position of this lane's column in column tile 0
    const float* xb = a.x + (size_t)b * G::C * a.L;

    rs_stage_group<G, K0>(c, a.blk[0].w1[0], 0, 0);

    // the margins only feed columns that are never stored, but they must hold finite numbers
    for (int i = tid; i < G::NG * 2 * 2 * G::MARG; i += G::NTHR) {
        const int pl = i / (2 * G::MARG), m = i - pl * (2 * G::MARG);
        half8 z;
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = (_Float16)0.f;
        c.P[(size_t)pl * G::PW + (m < G::MARG ? m : G::NCOL + m)] = z;
    }

    bool pok[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int pos = pos_w + ct * 32;
        pok[ct] = pos >= 0 && pos < lin;
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
        if (!pok[ct]) {   // columns outside the sequence: zero, once (every later image store is predicated on the column being inside)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int gi = 0; gi < 4; ++gi) {
                    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                    _Float16* ph = rs_image_ptr<G>(c, mi, ct, gi);
                    const u32x2 z = {0u, 0u};
                    *reinterpret_cast<u32x2*>(ph) = z;
                    *reinterpret_cast<u32x2*>(ph + (size_t)G::PW * 8) = z;
                }
        }
    f32x16 xres[MI][CT], ysum[MI][CT];
    rs_load_x<G>(xb, a.L, lin, pos_w, c.half, xres);
    rs_block<G, K0>(c, a.blk[0], xres, pok);
    rs_stage_group<G, K1>(c, a.blk[1].w1[0], 0, 0);   // (the ring was retired by the barrier that ended the last convolution)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) ysum[mi][ct] = xres[mi][ct];
    rs_load_x<G>(xb, a.L, lin, pos_w, c.half, xres);
    rs_block<G, K1>(c, a.blk[1], xres, pok);
    rs_stage_group<G, K2>(c, a.blk[2].w1[0], 0, 0);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) ysum[mi][ct] += xres[mi][ct];
    rs_load_x<G>(xb, a.L, lin, pos_w, c.half, xres);
    rs_block<G, K2>(c, a.blk[2], xres, pok);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) ysum[mi][ct] += xres[mi][ct];

    if constexpr (!POST) {
        float* yb = a.y + (size_t)b * G::C * a.L;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int col = c.colw + ct * 32 + c.l31;
            const int pos = q0 - a.lhalo + col;
            const bool ok = col >= a.lhalo && col < a.lhalo + a.nto && pos < lin;
            const unsigned voff = (unsigned)(4 * c.half * a.L + (ok ? pos : 0));
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float* row = yb + (size_t)(32 * mi + (r & 3) + 8 * (r >> 2)) * a.L;
                    if (ok) row[voff] = ysum[mi][ct][r];
                }
        }
    } else {
        // conv_post on the tile: lrelu(xs * in_scale, slope) as fp32 [C][PWF] in the (now idle) image area, then every thread of the
        // first nto / 4 owns four consecutive output samples — the same ci-major, tap-minor fmaf chain as conv_cout1_kernel
        // (conv1d.hip), so the fused stage is bit-identical to the layer-by-layer path.
        float* F = reinterpret_cast<float*>(smem_raw);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int col = c.colw + ct * 32 + c.l31;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ch = 32 * mi + (r & 3) + 8 * (r >> 2) + 4 * c.half;
                    float t = ysum[mi][ct][r] * a.post_in_scale;
                    t = fmaxf(t, t * a.post_slope);
                    F[(size_t)ch * G::PWF + G::FOFF + col] = pok[ct] ? t : 0.f;
                }
        }
        __syncthreads();
        if (4 * tid < a.nto) {
            const float* win = F + G::FOFF + a.lhalo - 3 + 4 * tid;   // 16-byte aligned: lhalo % 4 == 0
            float acc4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
            for (int ci = 0; ci < G::C; ++ci) {
                float w[12];
#pragma unroll
                for (int v4 = 0; v4 < 3; ++v4) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(win + (size_t)ci * G::PWF + 4 * v4);
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
            float* yb = a.wav + (size_t)b * a.L;
            const int q = q0 + 4 * tid;
            float res[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) res[e] = apply_act((acc4[e] + bv + 0.f) * a.post_out_scale, a.post_act) + 0.f;
            if (a.nf_flag) {
                bool bad = false;
#pragma unroll
                for (int e = 0; e < 4; ++e) bad = bad || (q + e < a.L && !(fabsf(res[e]) <= 3.0e38f));
                if (bad) atomicOr(a.nf_flag, 1u);
            }
            if (q + 3 < a.L && (((uintptr_t)(yb + q)) & 15) == 0) {
                const f32x4 o = {res[0], res[1], res[2], res[3]};
                *reinterpret_cast<f32x4*>(yb + q) = o;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (q + e < a.L) yb[q + e] = res[e];
            }
        }
    }
}

#ifdef TTSC_RS_PROBE   // development: compile ONE instantiation (tools/kernel_resources.py -DTTSC_RS_PROBE=1,3,8,3,7,11,true)
template __global__ void rbstage_f16x3_kernel<TTSC_RS_PROBE>(StageArgs);
}  // namespace ttsc
#else

template <int MI, int CT, int NW, int K0, int K1, int K2, bool POST>
static int launch_stage(StageArgs& a, int B, hipStream_t s) {
    using G = StageGeo<MI, CT, NW>;
    constexpr int rmax = (rs_grp(MI, K0) > rs_grp(MI, K1) ? rs_grp(MI, K0) : rs_grp(MI, K1)) > rs_grp(MI, K2)
                             ? (rs_grp(MI, K0) > rs_grp(MI, K1) ? rs_grp(MI, K0) : rs_grp(MI, K1))
                             : rs_grp(MI, K2);
    constexpr size_t lds = (size_t)G::IMG_ITEMS * 16 + (size_t)2 * rmax * G::STEP_ITEMS * 16;
    static_assert(lds <= 160 * 1024, "image + weight ring exceed the LDS");
    static_assert((size_t)G::C * G::PWF * 4 <= (size_t)G::IMG_ITEMS * 16, "conv_post staging exceeds the image area");
    auto kern = rbstage_f16x3_kernel<MI, CT, NW, K0, K1, K2, POST>;
    if (int rc = ensure_full_lds((const void*)kern)) return rc;
    dim3 grid((unsigned)ceil_div(a.L, a.nto), (unsigned)B);
    hipLaunchKernelGGL(kern, grid, dim3(64 * NW), lds, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("rbstage_f16x3_kernel launch failed: %s", hipGetErrorString(e));
        return TTSC_EHIP;
    }
    return TTSC_OK;
}

}  // namespace ttsc

using namespace ttsc;

static int stage_pair_ok(const ttsc_conv1d* c1, const ttsc_conv1d* c2, int C, int k) {
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

static int post_ok(const ttsc_conv1d* post, int C) {
    if (!post) return 1;
    const auto& g = post->cfg;
    return !g.transposed && post->groups == 1 && g.in_channels == C && g.out_channels == 1 && g.kernel_size == 7 && g.dilation == 1 &&
           g.padding == 3 && g.stride == 1 && post->w_plain_dev && !post->dev_weights;
}

// convs1 / convs2: [3 blocks][3 pairs] flattened; block j must have kernel size {3, 7, 11}[j]
extern "C" int ttsc_rbstage_supported(const ttsc_conv1d* const* convs1, const ttsc_conv1d* const* convs2, int32_t nblocks, int32_t npairs,
                                      const ttsc_conv1d* post) {
    if (!convs1 || !convs2 || nblocks != RS_NB || npairs != RS_NP || !convs1[0]) return 0;
    const int C = convs1[0]->cfg.in_channels;
    if (C != 32) return 0;
    static const int ks[RS_NB] = {3, 7, 11};
    for (int j = 0; j < RS_NB; ++j)
        for (int p = 0; p < RS_NP; ++p)
            if (!stage_pair_ok(convs1[j * RS_NP + p], convs2[j * RS_NP + p], C, ks[j])) return 0;
    return post_ok(post, C);
}

extern "C" int ttsc_rbstage_forward(const ttsc_conv1d* const* convs1, const ttsc_conv1d* const* convs2, int32_t nblocks, int32_t npairs,
                                    const float* x, int32_t B, int64_t L, float* y, const ttsc_conv1d* post, const ttsc_conv1d_epilogue* post_ep,
                                    float* wav, const int32_t* len_dev, int32_t shape, void* stream) {
    TTSC_REQUIRE(convs1 && convs2 && x, "ttsc_rbstage_forward: null argument");
    TTSC_REQUIRE(ttsc_rbstage_supported(convs1, convs2, nblocks, npairs, post), "ttsc_rbstage_forward: these layers are not eligible for the fused stage");
    TTSC_REQUIRE(post ? (wav != nullptr) : (y != nullptr && y != x), "ttsc_rbstage_forward: output missing (or y aliases x)");
    TTSC_REQUIRE(B > 0 && L > 0 && L < (1ll << 30), "ttsc_rbstage_forward: bad B/L");
    StageArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x;
    a.y = y;
    a.wav = wav;
    a.len = len_dev;
    a.L = (int)L;
    int halo = 0;
    for (int j = 0; j < RS_NB; ++j) {
        int hj = 0;
        for (int p = 0; p < RS_NP; ++p) {
            const ttsc_conv1d *c1 = convs1[j * RS_NP + p], *c2 = convs2[j * RS_NP + p];
            StageBlock& bk = a.blk[j];
            bk.w1[p] = reinterpret_cast<const half8*>(c1->phases[0].wph_dev);
            bk.w2[p] = reinterpret_cast<const half8*>(c2->phases[0].wph_dev);
            bk.b1[p] = c1->bias_dev;
            bk.b2[p] = c2->bias_dev;
            const float s1 = c1->act_scale, s2 = c2->act_scale;   // (same folding as ttsc_rbchain_forward)
            bk.xs[p] = s1;
            bk.us1[p] = c1->w_unscale * (s2 / s1);
            bk.bs1[p] = s2;
            bk.us2[p] = c2->w_unscale / s2;
            bk.d1[p] = c1->cfg.dilation;
            hj += (bk.d1[p] + 1) * (c1->cfg.kernel_size - 1) / 2;
        }
        halo = hj > halo ? hj : halo;
    }
    if (post) {
        halo += 3;
        a.wpost = post->w_plain_dev;
        a.bpost = post->bias_dev;
        a.post_in_scale = post_ep ? post_ep->in_scale : 1.f;
        a.post_slope = post_ep ? post_ep->in_slope : 1.f;
        a.post_out_scale = post_ep ? post_ep->out_scale : 1.f;
        a.post_act = post_ep ? post_ep->out_act : TTSC_ACT_NONE;
        TTSC_REQUIRE(!(post_ep && (post_ep->accumulate || post_ep->gate_dev)), "ttsc_rbstage_forward: conv_post epilogue options not available when fused");
        a.nf_flag = post->nf_flag;
    }
    a.lhalo = (int)round_up(halo, 32);   // tile stores start on a 128-byte boundary of the row
    hipStream_t s = (hipStream_t)stream;
    if (shape == 1) {
        a.nto = ((4 * 6 * 32 - a.lhalo - halo) / 32) * 32;
        return post ? launch_stage<1, 6, 4, 3, 7, 11, true>(a, B, s) : launch_stage<1, 6, 4, 3, 7, 11, false>(a, B, s);
    }
    a.nto = ((8 * 3 * 32 - a.lhalo - halo) / 32) * 32;
    return post ? launch_stage<1, 3, 8, 3, 7, 11, true>(a, B, s) : launch_stage<1, 3, 8, 3, 7, 11, false>(a, B, s);
}
#endif  // TTSC_RS_PROBE
