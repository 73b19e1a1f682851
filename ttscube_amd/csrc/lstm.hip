// LSTM / BiLSTM sequence recurrence as one persistent workgroup per (utterance tile, direction) — gfx950.
//
// torch.nn.LSTM semantics (gate order i,f,g,o; `_reverse` direction; stacked layers) as used by the mel decoders:
//   Languasito2  _char_rnn_{t,g}, _dur_rnn, _pitch_rnn, _cond_rnn, _lm_{t,g}   cube/networks/modules.py:873-905
//   CubenetTextcoder _rnn_char, _rnn_overlay, _dur_rnn, _pitch_rnn, _mel_rnn    cube/networks/textcoder.py:55-92
//
// Split of one layer:  (1) input projection for ALL time steps as one MFMA GEMM (gemm.hip): xg[b,t,dir,4H] =
// x[b,t,:] . W_ih^T + (b_ih + b_hh);  (2) this kernel: the sequential part  gates_t = xg_t + W_hh h_{t-1}, with thread j
// owning hidden unit j (its four gate rows are streamed from L2 as 16-byte packed loads, c_j stays in a register,
// h lives in LDS double-buffered) — no inter-workgroup communication, one workgroup barrier per step.
// Ragged batches: `lengths[b]` gives pack_padded_sequence semantics (reverse direction starts at len-1, outputs
// beyond len are zero), so a padded batch reproduces the per-utterance results exactly.
//
// Kernel choice (lstm_forward_impl):
//   H = 64 / 128        lstm_seq_resident_kernel      one thread per gate row, W_hh in registers, no stream at all
//   H = 256 / 512       lstm_seq_split_res_kernel     4 / 16 workgroups per (utterance, direction), 128 weights per thread in
//                                                      registers, tagged-granule hand-off
//                       lstm_seq_split_res_nb_kernel  the same member groups stepping 2 / 4 / 8 utterances of one direction together when the
//                                                      batch has more pairs than a launch holds (or the caller asks: ttsc_lstm_set_group_size);
//                                                      per utterance the same arithmetic; up to three consecutive launches
//   few sequences, other H   lstm_seq_split_kernel    <= 4 workgroups per sequence streaming their rows, counter hand-off
//   everything else     lstm_seq_kernel<BT>           one workgroup per (utterance tile, direction), rows streamed from L2
#include <algorithm>
#include <atomic>
#include <cstring>

#include "common.hpp"
#include "../../include/ttscube_math.h"
#include "rnn_chain.hpp"

namespace ttsc {

struct LstmArgs {
    const float* xg;     // [B, T, ndir*4H]
    const float* whh;    // [ndir][H/4][4H][4]
    float* y;            // [B, T, ldy], direction d writes columns [yoff + d*H, yoff + (d+1)*H)
    const int* lengths;  // [B] or null
    float* h_n;          // optional [ndir, B, H] final hidden state
    float* c_n;
    const float* h_0;    // optional initial state [ndir, B, H]
    const float* c_0;
    int B, T, H, ndir, ldy, yoff;
    float* gates_out;    // training: post-activation gates [B, T, ndir*4H] (i,f,g,o) saved for the backward kernel, or null
    float* c_out;        // training: cell states [B, T, ndir*H], or null
};

template <int BT>
__global__ __launch_bounds__(512) void lstm_seq_kernel(LstmArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];  // h[2][BT][H]
    const int H = a.H, H4 = 4 * H;
    const int j = threadIdx.x;
    const bool unit = j < H;
    const int dir = blockIdx.y;
    const float* whh = a.whh + (size_t)dir * H * H4;
    int len[BT], bi[BT];
    float c[BT];
#pragma unroll
    for (int u = 0; u < BT; ++u) {
        const int b = blockIdx.x * BT + u;
        bi[u] = b < a.B ? b : a.B - 1;
        len[u] = b < a.B ? (a.lengths ? a.lengths[bi[u]] : a.T) : 0;
        c[u] = (unit && a.c_0) ? a.c_0[((size_t)dir * a.B + bi[u]) * H + j] : 0.f;
        if (unit) sm[u * H + j] = a.h_0 ? a.h_0[((size_t)dir * a.B + bi[u]) * H + j] : 0.f;
    }
    __syncthreads();
    int cur = 0;
    for (int s = 0; s < a.T; ++s) {
        const float* hc = sm + cur * BT * H;
        float* hn = sm + (cur ^ 1) * BT * H;
        if (unit) {
            float acc[BT][4];
            int tpos[BT];
#pragma unroll
            for (int u = 0; u < BT; ++u) {
                tpos[u] = dir == 0 ? s : (len[u] - 1 - s);
                const bool ok = s < len[u];
                const float* xr = a.xg + ((size_t)bi[u] * a.T + (ok ? tpos[u] : 0)) * ((size_t)a.ndir * H4) + (size_t)dir * H4 + j;
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[u][g] = xr[g * H];
            }
            lstm_chain<BT, 4, 2>(acc, whh, H4, H, j, hc, H, H);
#pragma unroll
            for (int u = 0; u < BT; ++u) {
                const bool ok = s < len[u];
                const float ig = ttsc_sigmoidf(acc[u][0]);
                const float fg = ttsc_sigmoidf(acc[u][1]);
                const float gg = ttsc_tanhf(acc[u][2]);
                const float og = ttsc_sigmoidf(acc[u][3]);
                const float cn = fmaf(fg, c[u], ig * gg);
                const float hv = og * ttsc_tanhf(cn);
                if (ok) {
                    c[u] = cn;
                    hn[u * H + j] = hv;
                    a.y[((size_t)bi[u] * a.T + tpos[u]) * a.ldy + a.yoff + dir * H + j] = hv;
                    if (a.gates_out) {
                        float* gp = a.gates_out + ((size_t)bi[u] * a.T + tpos[u]) * ((size_t)a.ndir * H4) + (size_t)dir * H4 + j;
                        gp[0] = ig;
                        gp[H] = fg;
                        gp[2 * H] = gg;
                        gp[3 * H] = og;
                        a.c_out[((size_t)bi[u] * a.T + tpos[u]) * ((size_t)a.ndir * H) + (size_t)dir * H + j] = cn;
                    }
                } else {
                    hn[u * H + j] = hc[u * H + j];
                    // padded positions read as zeros (pad_packed_sequence); each padded t is written exactly once:
                    // position s itself is >= len for both directions
                    if (blockIdx.x * BT + u < a.B) a.y[((size_t)bi[u] * a.T + s) * a.ldy + a.yoff + dir * H + j] = 0.f;
                }
            }
        }
        __syncthreads();
        cur ^= 1;
    }
    if (unit) {
#pragma unroll
        for (int u = 0; u < BT; ++u) {
            if (blockIdx.x * BT + u < a.B) {
                if (a.h_n) a.h_n[((size_t)dir * a.B + bi[u]) * H + j] = sm[cur * BT * H + u * H + j];
                if (a.c_n) a.c_n[((size_t)dir * a.B + bi[u]) * H + j] = c[u];
            }
        }
    }
}


// Small hidden sizes (H <= 128: Languasito2's frame-level `_cond_rnn` has H = 64 and runs 300-700 steps per sentence): W_hh of
// one direction is only 64 / 256 KB, so nothing has to be streamed per step.  One workgroup per (utterance, direction), one THREAD
// PER GATE ROW (4H threads), each holding its whole row of W_hh in registers for the entire sequence.  A step is: broadcast reads
// of h from LDS, H fused multiply-adds per thread in four interleaved partial sums (a fixed order, the same for every batch
// size, so a padded batch still reproduces each utterance run alone bit for bit), gate pre-activations through LDS, the cell /
// hidden update by the H threads of gate i, two barriers.  The input-projection value of the NEXT step is loaded before the
// current step's chain.  lstm_seq_kernel (one thread per unit, rows streamed from L2 by a single wave at H = 64) took
// 3.4 us per step and layer at H = 64; this kernel: tools/bench_lstm.py.
typedef float lstm_v2 __attribute__((ext_vector_type(2)));

template <int HH>
__global__ __launch_bounds__(4 * HH) void lstm_seq_resident_kernel(LstmArgs a) {
    constexpr int H = HH, H4 = 4 * HH;
    __shared__ __attribute__((aligned(16))) float hbuf[2][HH];
    __shared__ float gbuf[4 * HH];
    const int r = threadIdx.x;            // gate row: gate r / H, unit r % H
    const int j = r % H;
    const bool upd = r < H;               // the threads of gate i also own the cell / hidden update of unit j
    const int b = blockIdx.x, dir = blockIdx.y;
    const int len = a.lengths ? a.lengths[b] : a.T;
    float w[HH];
    {
        const float4* w4 = reinterpret_cast<const float4*>(a.whh + (size_t)dir * H * H4) + r;   // packed [H/4][4H][4]
#pragma unroll
        for (int kb = 0; kb < H / 4; ++kb) {
            const float4 v = w4[(size_t)kb * H4];
            w[4 * kb] = v.x;
            w[4 * kb + 1] = v.y;
            w[4 * kb + 2] = v.z;
            w[4 * kb + 3] = v.w;
        }
    }
    float c = (upd && a.c_0) ? a.c_0[((size_t)dir * a.B + b) * H + j] : 0.f;
    if (upd) hbuf[0][j] = a.h_0 ? a.h_0[((size_t)dir * a.B + b) * H + j] : 0.f;
    const size_t xstride = (size_t)a.ndir * H4;
    const float* xb = a.xg + (size_t)b * a.T * xstride + (size_t)dir * H4 + r;
    auto tpos_of = [&](int s) { return dir == 0 ? s : (len - 1 - s); };
    float xnext = len > 0 ? xb[(size_t)tpos_of(0) * xstride] : 0.f;
    __syncthreads();
    int cur = 0;
    for (int s = 0; s < a.T; ++s) {
        const bool ok = s < len;
        const int tpos = tpos_of(s);
        const float xcur = xnext;
        if (s + 1 < len) xnext = xb[(size_t)tpos_of(s + 1) * xstride];
        if (ok) {
            const float4* h4 = reinterpret_cast<const float4*>(hbuf[cur]);
            // the four interleaved partial sums as two packed pairs: (p0, p1) and (p2, p3) advance with one v_pk_fma_f32 each (per lane the fused
            // multiply-add of the scalar form: same bits, half the vector instructions)
            lstm_v2 p01 = {xcur, 0.f}, p23 = {0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < H / 4; ++kb) {
                const float4 hv = h4[kb];
                p01 = __builtin_elementwise_fma((lstm_v2){w[4 * kb], w[4 * kb + 1]}, (lstm_v2){hv.x, hv.y}, p01);
                p23 = __builtin_elementwise_fma((lstm_v2){w[4 * kb + 2], w[4 * kb + 3]}, (lstm_v2){hv.z, hv.w}, p23);
            }
            // every thread applies ITS gate's activation (gate = r / H is uniform per wave): the four gates' transcendental functions run side by side on
            // four waves instead of in series on the update threads, which then only form c and h.  Same functions of the same values: same bits.
            const float pre = (p01.x + p01.y) + (p23.x + p23.y);
            gbuf[r] = (r / H == 2) ? ttsc_tanhf(pre) : ttsc_sigmoidf(pre);
        }
        __syncthreads();
        if (upd) {
            if (ok) {
                const float ig = gbuf[j];
                const float fg = gbuf[H + j];
                const float gg = gbuf[2 * H + j];
                const float og = gbuf[3 * H + j];
                const float cn = fmaf(fg, c, ig * gg);
                const float hv = og * ttsc_tanhf(cn);
                c = cn;
                hbuf[cur ^ 1][j] = hv;
                a.y[((size_t)b * a.T + tpos) * a.ldy + a.yoff + dir * H + j] = hv;
                if (a.gates_out) {
                    float* gp = a.gates_out + ((size_t)b * a.T + tpos) * xstride + (size_t)dir * H4 + j;
                    gp[0] = ig;
                    gp[H] = fg;
                    gp[2 * H] = gg;
                    gp[3 * H] = og;
                    a.c_out[((size_t)b * a.T + tpos) * ((size_t)a.ndir * H) + (size_t)dir * H + j] = cn;
                }
            } else {
                hbuf[cur ^ 1][j] = hbuf[cur][j];
                a.y[((size_t)b * a.T + s) * a.ldy + a.yoff + dir * H + j] = 0.f;   // padded positions read as zeros (each written once)
            }
        }
        __syncthreads();
        cur ^= 1;
    }
    if (upd) {
        if (a.h_n) a.h_n[((size_t)dir * a.B + b) * H + j] = hbuf[cur][j];
        if (a.c_n) a.c_n[((size_t)dir * a.B + b) * H + j] = c;
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Backward through time (training of the mel-decoder stacks): one persistent workgroup per
// (utterance, direction) walks the steps in reverse.  Thread k owns hidden unit k: it turns (dy_t + dh_rec, dc_next) into
// the four pre-activation gate gradients from the gates / cell states the forward saved, publishes them in LDS, and then
// evaluates its own dh_rec[k] = sum_r W_hh[r,k] * dgates[r] as one chain over the 4H gate rows (W_hh^T packed
// [4H/4][H][4], the same 16-byte streaming as the forward).  The gate gradients of all steps go to HBM; the weight /
// input gradients are three plain GEMMs over them afterwards (dW_ih = dG^T x, dW_hh = dG^T h_prev, dx = dG W_ih).
struct LstmBwdArgs {
    const float* dy;      // [B, T, ldy]; direction d reads columns [yoff + d*H, yoff + (d+1)*H)
    const float* gates;   // [B, T, ndir*4H]
    const float* cst;     // [B, T, ndir*H]
    const float* whhT;    // [ndir][4H/4][H][4]
    float* dgates;        // [B, T, ndir*4H]
    const int* lengths;
    int B, T, H, ndir, ldy, yoff;
};

__global__ __launch_bounds__(512) void lstm_bwd_kernel(LstmBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];  // dg[4H]
    const int H = a.H, H4 = 4 * H;
    const int j = threadIdx.x;
    const bool unit = j < H;
    const int b = blockIdx.x, dir = blockIdx.y;
    const int len = a.lengths ? a.lengths[b] : a.T;
    const float* whhT = a.whhT + (size_t)dir * H * H4;
    const size_t gstride = (size_t)a.ndir * H4, cstride = (size_t)a.ndir * H;
    const float* gb = a.gates + (size_t)b * a.T * gstride + (size_t)dir * H4 + j;
    const float* cb = a.cst + (size_t)b * a.T * cstride + (size_t)dir * H + j;
    const float* dyb = a.dy + (size_t)b * a.T * a.ldy + a.yoff + dir * H + j;
    float* dgb = a.dgates + (size_t)b * a.T * gstride + (size_t)dir * H4 + j;
    if (unit)
        for (int t = len; t < a.T; ++t) {   // padded positions contribute nothing to the GEMMs that follow
            float* p = dgb + (size_t)t * gstride;
            p[0] = 0.f;
            p[H] = 0.f;
            p[2 * H] = 0.f;
            p[3 * H] = 0.f;
        }
    float dh_rec = 0.f, dc_next = 0.f;
    // saved values of the step about to be processed (prefetched during the previous step's chain)
    float ig = 0.f, fg = 0.f, gg = 0.f, og = 0.f, ct = 0.f, cp = 0.f, dyv = 0.f;
    auto fetch = [&](int s) __attribute__((always_inline)) {
        const int tpos = dir == 0 ? s : (len - 1 - s);
        const int tprev = dir == 0 ? tpos - 1 : tpos + 1;
        const float* g = gb + (size_t)tpos * gstride;
        ig = g[0];
        fg = g[H];
        gg = g[2 * H];
        og = g[3 * H];
        ct = cb[(size_t)tpos * cstride];
        cp = s > 0 ? cb[(size_t)tprev * cstride] : 0.f;
        dyv = dyb[(size_t)tpos * a.ldy];
    };
    if (unit && len > 0) fetch(len - 1);
    for (int s = len - 1; s >= 0; --s) {
        if (unit) {
            const int tpos = dir == 0 ? s : (len - 1 - s);
            const float dh = dyv + dh_rec;
            const float tc = ttsc_tanhf(ct);
            const float d_o = dh * tc * og * (1.f - og);
            const float dc = dc_next + dh * og * (1.f - tc * tc);
            const float d_i = dc * gg * ig * (1.f - ig);
            const float d_g = dc * ig * (1.f - gg * gg);
            const float d_f = dc * cp * fg * (1.f - fg);
            dc_next = dc * fg;
            float* p = dgb + (size_t)tpos * gstride;
            p[0] = d_i;
            p[H] = d_f;
            p[2 * H] = d_g;
            p[3 * H] = d_o;
            sm[j] = d_i;
            sm[H + j] = d_f;
            sm[2 * H + j] = d_g;
            sm[3 * H + j] = d_o;
        }
        __syncthreads();
        if (unit) {
            if (s > 0) fetch(s - 1);
            float acc[1][1] = {{0.f}};
            lstm_chain<1, 1, 4>(acc, whhT, H, 0, j, sm, H4, H4);
            dh_rec = acc[0][0];
        }
        __syncthreads();
    }
}


// H = 64 / 128 (Languasito2's frame-level `_cond_rnn`: 300-700 steps per sentence): lstm_bwd_kernel above runs ONE wave per H = 64 sequence, each
// thread streaming all 4H rows of W_hh^T from L2 every step (3.2 us per step).  Here a workgroup is (unit j, k-slice) pairs — 4 slices of H gate rows
// — with the slice's H weights of column j in registers for the whole sequence; the four partial sums meet in LDS.
template <int H>
__global__ __launch_bounds__(4 * H) void lstm_bwd_resident_kernel(LstmBwdArgs a) {
    __shared__ __attribute__((aligned(16))) float dg[4 * H];
    __shared__ float part[4][H];
    constexpr int H4 = 4 * H;
    const int u = threadIdx.x % H, ks = threadIdx.x / H;
    const bool owner = ks == 0;
    const int b = blockIdx.x, dir = blockIdx.y;
    const int len = a.lengths ? a.lengths[b] : a.T;
    const size_t gstride = (size_t)a.ndir * H4, cstride = (size_t)a.ndir * H;
    const float* gb = a.gates + (size_t)b * a.T * gstride + (size_t)dir * H4 + u;
    const float* cb = a.cst + (size_t)b * a.T * cstride + (size_t)dir * H + u;
    const float* dyb = a.dy + (size_t)b * a.T * a.ldy + a.yoff + dir * H + u;
    float* dgb = a.dgates + (size_t)b * a.T * gstride + (size_t)dir * H4 + u;
    float w[H];
    {
        const float4* w4 = reinterpret_cast<const float4*>(a.whhT + (size_t)dir * H * H4 + (size_t)(ks * (H / 4)) * H * 4) + u;
#pragma unroll
        for (int kb = 0; kb < H / 4; ++kb) {
            const float4 v = w4[(size_t)kb * H];
            w[4 * kb] = v.x;
            w[4 * kb + 1] = v.y;
            w[4 * kb + 2] = v.z;
            w[4 * kb + 3] = v.w;
        }
    }
    if (owner)
        for (int t = len; t < a.T; ++t) {
            float* p = dgb + (size_t)t * gstride;
            p[0] = 0.f;
            p[H] = 0.f;
            p[2 * H] = 0.f;
            p[3 * H] = 0.f;
        }
    float dh_rec = 0.f, dc_next = 0.f;
    float ig = 0.f, fg = 0.f, gg = 0.f, og = 0.f, ct = 0.f, cp = 0.f, dyv = 0.f;
    auto fetch = [&](int s) __attribute__((always_inline)) {
        const int tpos = dir == 0 ? s : (len - 1 - s);
        const int tprev = dir == 0 ? tpos - 1 : tpos + 1;
        const float* g = gb + (size_t)tpos * gstride;
        ig = g[0];
        fg = g[H];
        gg = g[2 * H];
        og = g[3 * H];
        ct = cb[(size_t)tpos * cstride];
        cp = s > 0 ? cb[(size_t)tprev * cstride] : 0.f;
        dyv = dyb[(size_t)tpos * a.ldy];
    };
    if (owner && len > 0) fetch(len - 1);
    for (int s = len - 1; s >= 0; --s) {
        if (owner) {
            const int tpos = dir == 0 ? s : (len - 1 - s);
            const float dh = dyv + dh_rec;
            const float tc = ttsc_tanhf(ct);
            const float d_o = dh * tc * og * (1.f - og);
            const float dc = dc_next + dh * og * (1.f - tc * tc);
            const float d_i = dc * gg * ig * (1.f - ig);
            const float d_g = dc * ig * (1.f - gg * gg);
            const float d_f = dc * cp * fg * (1.f - fg);
            dc_next = dc * fg;
            dg[u] = d_i;
            dg[H + u] = d_f;
            dg[2 * H + u] = d_g;
            dg[3 * H + u] = d_o;
            float* p = dgb + (size_t)tpos * gstride;
            p[0] = d_i;
            p[H] = d_f;
            p[2 * H] = d_g;
            p[3 * H] = d_o;
        }
        if (s == 0) break;
        __syncthreads();
        if (owner) fetch(s - 1);   // in flight during the chain
        float x = 0.f;
        const float4* d4 = reinterpret_cast<const float4*>(dg + ks * H);
#pragma unroll
        for (int kb = 0; kb < H / 4; ++kb) {
            const float4 dv = d4[kb];
            x = fmaf(w[4 * kb], dv.x, x);
            x = fmaf(w[4 * kb + 1], dv.y, x);
            x = fmaf(w[4 * kb + 2], dv.z, x);
            x = fmaf(w[4 * kb + 3], dv.w, x);
        }
        part[ks][u] = x;
        __syncthreads();
        if (owner) dh_rec = ((part[0][u] + part[1][u]) + part[2][u]) + part[3][u];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Split recurrences (same scheme as gru.hip): when B * ndir is small the kernels above leave most CUs idle while each
// workgroup is bound by ONE CU's L2 load path (W_hh: 1 MB for H = 256, 4 MB for H = 512, per step).  Here G workgroups share
// one (utterance, direction): member m owns H/G hidden units and streams only their gate rows, its 512 threads are (unit,
// k-slice) pairs whose partial sums meet in LDS, and the members exchange h_t (forward, through y) / the gate gradients
// (backward, through dgates) once per step.  Results differ from the single-workgroup kernels only in summation order.
// All G * B * ndir workgroups must be co-resident: the host takes this path only when they fit the CUs.
struct LstmSplitArgs {
    LstmArgs f;
    LstmBwdArgs bw;
    unsigned* cnt;        // [B * ndir] monotonic counters (zeroed by the host before the launch)
    unsigned* abort_word;
    int G, HU, KS;        // members, hidden units per member, k-slices (HU * KS = 512 threads)
    int p0;               // lstm_seq_split_res_kernel: first (utterance, direction) pair of this launch (pair = b * ndir + dir)
};

__global__ __launch_bounds__(512) void lstm_seq_split_kernel(LstmSplitArgs s) {
    extern __shared__ __attribute__((aligned(16))) float sm[];   // h[H] | part[KS][4][HU]
    __shared__ int ok_s;
    const LstmArgs& a = s.f;
    const int H = a.H, H4 = 4 * H, HU = s.HU, KS = s.KS;
    const int tid = threadIdx.x, u = tid % HU, ks = tid / HU;
    const int m = blockIdx.x, b = blockIdx.y, dir = blockIdx.z;
    const int j = m * HU + u;
    const int KL = H / KS;
    float* hs = sm;
    float* part = sm + H;
    unsigned* cnt = s.cnt + (b * a.ndir + dir);
    const bool owner = ks == 0;
    const int len = a.lengths ? a.lengths[b] : a.T;
    const float* whh = a.whh + (size_t)dir * H * H4;
    float* yb = a.y + (size_t)b * a.T * a.ldy + a.yoff + dir * H;
    float c = (owner && a.c_0) ? a.c_0[((size_t)dir * a.B + b) * H + j] : 0.f;
    float hlast = (owner && a.h_0) ? a.h_0[((size_t)dir * a.B + b) * H + j] : 0.f;
    if (owner)
        for (int t = len; t < a.T; ++t) yb[(size_t)t * a.ldy + j] = 0.f;   // pad_packed_sequence: zeros beyond the length
    for (int st = 0; st < len; ++st) {
        const int tpos = dir == 0 ? st : (len - 1 - st);
        const int tprev = dir == 0 ? tpos - 1 : tpos + 1;
        float xv[4] = {0.f, 0.f, 0.f, 0.f};
        if (owner) {   // in flight while the other members finish the previous step
            const float* xr = a.xg + ((size_t)b * a.T + tpos) * ((size_t)a.ndir * H4) + (size_t)dir * H4 + j;
#pragma unroll
            for (int g = 0; g < 4; ++g) xv[g] = xr[g * H];
        }
        if (st > 0) {
            if (!g_wait(cnt, (unsigned)st * (unsigned)s.G, s.abort_word, &ok_s)) return;
            for (int i = tid; i < H; i += 512) hs[i] = g_ld(yb + (size_t)tprev * a.ldy + i);
        } else {
            for (int i = tid; i < H; i += 512) hs[i] = a.h_0 ? a.h_0[((size_t)dir * a.B + b) * H + i] : 0.f;
        }
        __syncthreads();
        float acc[1][4] = {{0.f, 0.f, 0.f, 0.f}};
        lstm_chain<1, 4, 2>(acc, whh + (size_t)(ks * KL / 4) * H4 * 4, H4, H, j, hs + ks * KL, H, KL);
#pragma unroll
        for (int g = 0; g < 4; ++g) part[(ks * 4 + g) * HU + u] = acc[0][g];
        __syncthreads();
        if (owner) {
            float gs[4] = {xv[0], xv[1], xv[2], xv[3]};
            for (int q = 0; q < KS; ++q)
#pragma unroll
                for (int g = 0; g < 4; ++g) gs[g] += part[(q * 4 + g) * HU + u];
            const float ig = ttsc_sigmoidf(gs[0]);
            const float fg = ttsc_sigmoidf(gs[1]);
            const float gg = ttsc_tanhf(gs[2]);
            const float og = ttsc_sigmoidf(gs[3]);
            c = fmaf(fg, c, ig * gg);
            hlast = og * ttsc_tanhf(c);
            g_st(yb + (size_t)tpos * a.ldy + j, hlast);
            if (a.gates_out) {
                float* gp = a.gates_out + ((size_t)b * a.T + tpos) * ((size_t)a.ndir * H4) + (size_t)dir * H4 + j;
                gp[0] = ig;
                gp[H] = fg;
                gp[2 * H] = gg;
                gp[3 * H] = og;
                a.c_out[((size_t)b * a.T + tpos) * ((size_t)a.ndir * H) + (size_t)dir * H + j] = c;
            }
        }
        g_publish(cnt);
    }
    if (owner) {
        if (a.h_n) a.h_n[((size_t)dir * a.B + b) * H + j] = hlast;
        if (a.c_n) a.c_n[((size_t)dir * a.B + b) * H + j] = c;
    }
}

// The same split with the member's rows of W_hh RESIDENT IN REGISTERS (4 * KL <= 128 weights per thread: H = 256 over 4
// members) and the state handed over as 8-byte {value, step tag} granules that the consumers poll directly (one agent-scope store
// per value, no counter, no producer-side drain; ring of two slots per sequence, see wavernn_tile.hip for the argument why two
// suffice).  Summation order = lstm_seq_split_kernel's (k-ordered chain per k-slice, slices added in order), so results are
// bit-identical to it; what changes is the step time: no 256 KB weight stream and one round trip instead of three per step.
typedef unsigned long long lstm_u64;
typedef float lstm_f32x2 __attribute__((ext_vector_type(2)));

// Up to N granules per thread in ONE round trip: every load (src[off[r]] for the r set in `mask`) is issued before the first tag is looked at; a pass is repeated only while
// some granule is missing.  Polling granule by granule pays the L2 round trip once per granule even when all of them have arrived (the backward recurrence
// hands 4H values to 512 threads: two dependent round trips per step; the batched forward kernel NB * H).  Values land in out[dst[r]].
template <int N>
__device__ __forceinline__ bool lstm_poll_n(const lstm_u64* src, const int (&off)[N], const int (&dst)[N], unsigned mask, unsigned tag, unsigned* abort_word, float* out) {
    lstm_u64 g[N];
    unsigned spins = 0;
    for (;;) {
#pragma unroll
        for (int r = 0; r < N; ++r)
            if (mask >> r & 1u) g[r] = __hip_atomic_load(src + off[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool all = true;
#pragma unroll
        for (int r = 0; r < N; ++r)
            if (mask >> r & 1u) all = all && ((unsigned)(g[r] >> 32) == tag);
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
        if (mask >> r & 1u) out[dst[r]] = __uint_as_float((unsigned)g[r]);
    return true;
}

template <int KL>
__global__ __launch_bounds__(512) void lstm_seq_split_res_kernel(LstmSplitArgs s, lstm_u64* ring) {
    extern __shared__ __attribute__((aligned(16))) float sm[];   // h[H] | part[KS][4][HU] | act[4][HU]
    const LstmArgs& a = s.f;
    const int H = a.H, H4 = 4 * H, HU = s.HU, KS = s.KS;
    const int tid = threadIdx.x, u = tid % HU, ks = tid / HU;
    const int m = blockIdx.x, pair = s.p0 + blockIdx.y, b = pair / a.ndir, dir = pair % a.ndir;
    const int j = m * HU + u;
    float* hs = sm;
    float* part = sm + H;
    float* act = part + KS * 4 * HU;   // [4][HU] activated gates of this step
    lstm_u64* rg = ring + (size_t)pair * 2 * H;
    const bool owner = ks == 0;
    const int len = a.lengths ? a.lengths[b] : a.T;
    float* yb = a.y + (size_t)b * a.T * a.ldy + a.yoff + dir * H;
    // gate pairs (i, f) and (g, o) side by side: one v_pk_fma_f32 advances two of the four k-ordered chains of a thread (each lane of the packed
    // instruction is the fused multiply-add the scalar form issues: same bits, half the vector instructions of the step's 128)
    lstm_f32x2 w[2][KL];
    {
        // packed [H/4][4H][4]: row g*H + j, k-block kb holds k = 4*kb .. 4*kb+3
        const float4* w4 = reinterpret_cast<const float4*>(a.whh + (size_t)dir * H * H4) + j;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int kb = 0; kb < KL / 4; ++kb) {
                const float4 v = w4[(size_t)(ks * (KL / 4) + kb) * H4 + g * H];
                w[g >> 1][4 * kb][g & 1] = v.x;
                w[g >> 1][4 * kb + 1][g & 1] = v.y;
                w[g >> 1][4 * kb + 2][g & 1] = v.z;
                w[g >> 1][4 * kb + 3][g & 1] = v.w;
            }
    }
    float c = (owner && a.c_0) ? a.c_0[((size_t)dir * a.B + b) * H + j] : 0.f;
    float hlast = (owner && a.h_0) ? a.h_0[((size_t)dir * a.B + b) * H + j] : 0.f;
    if (owner)
        for (int t = len; t < a.T; ++t) yb[(size_t)t * a.ldy + j] = 0.f;   // pad_packed_sequence: zeros beyond the length
    for (int st = 0; st < len; ++st) {
        const int tpos = dir == 0 ? st : (len - 1 - st);
        // gate g of the member's units is finished by k-slice row g (ks = 0..3: input-projection value + the KS partial sums in slice order + its
        // activation) — four rows working side by side instead of row 0 walking all four gates and five transcendental functions in series; row 0 then
        // only forms c and h.  Per value the same operations in the same order as before: same bits.
        float xv = 0.f;
        if (ks < 4)    // in flight while the other members finish the previous step
            xv = a.xg[((size_t)b * a.T + tpos) * ((size_t)a.ndir * H4) + (size_t)dir * H4 + (size_t)ks * H + j];
        bool fail = false;
        if (st > 0) {
            const lstm_u64* src = rg + (size_t)((st - 1) & 1) * H;
            for (int i = tid; i < H; i += 512) {
                lstm_u64 gq;
                unsigned spins = 0;
                for (;;) {
                    gq = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((unsigned)(gq >> 32) == (unsigned)st) break;
                    if (++spins > GS_SPIN_LIMIT || ((spins & 63u) == 0u && __hip_atomic_load(s.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                        __hip_atomic_store(s.abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(s.abort_word + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sticky copy
                        fail = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                hs[i] = __uint_as_float((unsigned)gq);
            }
        } else {
            for (int i = tid; i < H; i += 512) hs[i] = a.h_0 ? a.h_0[((size_t)dir * a.B + b) * H + i] : 0.f;
        }
        if (__syncthreads_or(fail)) return;
        lstm_f32x2 acc[2] = {{0.f, 0.f}, {0.f, 0.f}};
        {
            const float4* h4 = reinterpret_cast<const float4*>(hs + ks * KL);
#pragma unroll
            for (int kb = 0; kb < KL / 4; ++kb) {
                const float4 hv = h4[kb];
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    lstm_f32x2 x = acc[gp];
                    x = __builtin_elementwise_fma(w[gp][4 * kb], (lstm_f32x2){hv.x, hv.x}, x);
                    x = __builtin_elementwise_fma(w[gp][4 * kb + 1], (lstm_f32x2){hv.y, hv.y}, x);
                    x = __builtin_elementwise_fma(w[gp][4 * kb + 2], (lstm_f32x2){hv.z, hv.z}, x);
                    x = __builtin_elementwise_fma(w[gp][4 * kb + 3], (lstm_f32x2){hv.w, hv.w}, x);
                    acc[gp] = x;
                }
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) part[(ks * 4 + g) * HU + u] = acc[g >> 1][g & 1];
        __syncthreads();
        if (ks < 4) {
            float gsum = xv;
            for (int q = 0; q < KS; ++q) gsum += part[(q * 4 + ks) * HU + u];
            act[ks * HU + u] = ks == 2 ? ttsc_tanhf(gsum) : ttsc_sigmoidf(gsum);
        }
        __syncthreads();
        if (owner) {
            const float ig = act[u], fg = act[HU + u], gg = act[2 * HU + u], og = act[3 * HU + u];
            c = fmaf(fg, c, ig * gg);
            hlast = og * ttsc_tanhf(c);
            __hip_atomic_store(rg + (size_t)(st & 1) * H + j, ((lstm_u64)(unsigned)(st + 1) << 32) | (lstm_u64)__float_as_uint(hlast), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            yb[(size_t)tpos * a.ldy + j] = hlast;
            if (a.gates_out) {
                float* gp = a.gates_out + ((size_t)b * a.T + tpos) * ((size_t)a.ndir * H4) + (size_t)dir * H4 + j;
                gp[0] = ig;
                gp[H] = fg;
                gp[2 * H] = gg;
                gp[3 * H] = og;
                a.c_out[((size_t)b * a.T + tpos) * ((size_t)a.ndir * H) + (size_t)dir * H + j] = c;
            }
        }
    }
    if (owner) {
        if (a.h_n) a.h_n[((size_t)dir * a.B + b) * H + j] = hlast;
        if (a.c_n) a.c_n[((size_t)dir * a.B + b) * H + j] = c;
    }
}

// Several utterances per member group.  A launch of lstm_seq_split_res_kernel holds cus / G (utterance, direction) pairs; a padded batch of 64 sentences
// (128 pairs at H = 256) therefore took two consecutive launches, each paying the full hand-off latency per step for 128 FMAs per thread.  Here the G
// members of a group keep the same 128 weights per thread and step NB utterances of one direction together: the exchange latency is paid once per step
// for NB sequences, the FMAs become v_pk_fma_f32 over {utterance 2p, utterance 2p + 1} with the weight broadcast by op_sel (two chains per
// instruction), and the k-slice threads 0 .. NB-1 each own one utterance's gate reduction / activations / publication, so those run side by side too.
// Per utterance the arithmetic is the NB = 1 kernel's — same k-ordered fma chain per slice, slices added in order — so results are bit-identical to it
// and to the utterance run alone.  Ragged groups: a finished (or absent) utterance is neither polled nor published; its lane of the packed chain runs
// on stale h and is discarded.
typedef float lstm_f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void lstm_pkfma_lo(lstm_f2& acc, const lstm_f2& w, const lstm_f2& h) {   // acc += w.x * h   (both halves)
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(w), "v"(h));
}
__device__ __forceinline__ void lstm_pkfma_hi(lstm_f2& acc, const lstm_f2& w, const lstm_f2& h) {   // acc += w.y * h
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(w), "v"(h));
}

template <int KL, int NB>
__global__ __launch_bounds__(512) void lstm_seq_split_res_nb_kernel(LstmSplitArgs s, lstm_u64* ring) {
    static_assert(NB == 2 || NB == 4 || NB == 8, "utterances per member group");
    extern __shared__ __attribute__((aligned(16))) float sm[];   // h[H][NB] (utterance-interleaved) | part[NB][KS][4][HU] | act[NB][4][HU]
    const LstmArgs& a = s.f;
    const int H = a.H, H4 = 4 * H, HU = s.HU, KS = s.KS;
    const int tid = threadIdx.x, u = tid % HU, ks = tid / HU;
    const int m = blockIdx.x, grp = s.p0 + blockIdx.y, dir = grp % a.ndir, b0 = (grp / a.ndir) * NB;
    const int j = m * HU + u;
    float* hs = sm;
    float* part = sm + NB * H;
    float* act = part + (size_t)NB * KS * 4 * HU;   // [NB][4][HU] activated gates of this step
    const int NG = KS / NB < 4 ? KS / NB : 4, GPR = 4 / NG;          // gate sets, gates per row
    const int aq = ks % NB, gset = ks / NB;
    const bool actor = gset < NG && b0 + aq < a.B;
    int len[NB], maxlen = 0;
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        len[q] = b0 + q < a.B ? (a.lengths ? a.lengths[b0 + q] : a.T) : 0;
        maxlen = len[q] > maxlen ? len[q] : maxlen;
    }
    // k-slice thread row `ks` owns utterance b0 + ks (KS >= 8 >= NB)
    const int ob = b0 + ks;
    const bool owner = ks < NB && ob < a.B;
    int olen = 0, alen = 0;
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        olen = (q == ks) ? len[q] : olen;
        alen = (q == aq) ? len[q] : alen;
    }
    lstm_u64* org = ring + (size_t)(ob * a.ndir + dir) * 2 * H;
    float* yb = a.y + (size_t)ob * a.T * a.ldy + a.yoff + dir * H;
    lstm_f2 w[4][KL / 2];
    {
        const float4* w4 = reinterpret_cast<const float4*>(a.whh + (size_t)dir * H * H4) + j;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int kb = 0; kb < KL / 4; ++kb) {
                const float4 v = w4[(size_t)(ks * (KL / 4) + kb) * H4 + g * H];
                w[g][2 * kb] = lstm_f2{v.x, v.y};
                w[g][2 * kb + 1] = lstm_f2{v.z, v.w};
            }
    }
    float c = (owner && a.c_0) ? a.c_0[((size_t)dir * a.B + ob) * H + j] : 0.f;
    float hlast = (owner && a.h_0) ? a.h_0[((size_t)dir * a.B + ob) * H + j] : 0.f;
    if (owner)
        for (int t = olen; t < a.T; ++t) yb[(size_t)t * a.ldy + j] = 0.f;
    // the (up to four) granules this thread fetches per step: ring offset without the parity term, LDS slot, length of the granule's utterance
    int pb[4], pd[4], pl[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int e = tid + r * 512;
        pb[r] = pd[r] = pl[r] = 0;
        if (e < NB * H) {
            const int q = e / H, i = e - q * H;
#pragma unroll
            for (int r2 = 0; r2 < NB; ++r2) pl[r] = (r2 == q) ? len[r2] : pl[r];
            pb[r] = (int)((size_t)((b0 + q) * a.ndir + dir) * 2 * H + i);
            pd[r] = i * NB + q;
        }
    }
    for (int st = 0; st < maxlen; ++st) {
        const bool mine = owner && st < olen;
        const int tpos = dir == 0 ? st : (olen - 1 - st);
        // (utterance, gate) items of the step's finish — input-projection value + KS partial sums in slice order + activation — spread over the k-slice
        // rows: row ks takes utterance ks % NB and the gates gset, gset + NG, ... (gset = ks / NB, NG = min(4, KS / NB) gate sets), so that with NB = 2 every
        // gate of both utterances has a row of its own and with NB = 4 every row takes two gates, instead of NB rows walking four gates and five
        // transcendental functions each while the others wait.  The owner rows then only form c and h.  Per value the same operations in the same order.
        const bool acting = actor && st < alen;
        float xv[4] = {0.f, 0.f, 0.f, 0.f};
        if (acting) {
            const int atpos = dir == 0 ? st : (alen - 1 - st);
            const float* xr = a.xg + ((size_t)(b0 + aq) * a.T + atpos) * ((size_t)a.ndir * H4) + (size_t)dir * H4 + j;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < GPR) xv[i] = xr[(gset + i * NG) * H];
        }
        bool fail = false;
        if (st == 0) {
            for (int e = tid; e < NB * H; e += 512) {
                const int q = e / H, i = e - q * H;
                hs[i * NB + q] = (a.h_0 && b0 + q < a.B) ? a.h_0[((size_t)dir * a.B + b0 + q) * H + i] : 0.f;
            }
        } else {
            // the thread's granules (one per 512 elements of the NB * H staged values; those of utterances that have ended are skipped and keep their
            // last value, which nothing reads) in one round trip; the first four are described by pb / pd / pl, set up before the loop
            if (NB * H <= 512) {   // one granule per thread (two utterances at H = 256, the padded batch's usual launch): the plain poll
                if (tid < NB * H && st < pl[0]) {
                    const lstm_u64* src = ring + pb[0] + ((st - 1) & 1) * H;
                    lstm_u64 gq;
                    unsigned spins = 0;
                    for (;;) {
                        gq = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if ((unsigned)(gq >> 32) == (unsigned)st) break;
                        if (++spins > GS_SPIN_LIMIT || ((spins & 63u) == 0u && __hip_atomic_load(s.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                            __hip_atomic_store(s.abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(s.abort_word + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sticky copy
                            fail = true;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    hs[pd[0]] = __uint_as_float((unsigned)gq);
                }
            } else {
                const int par = ((st - 1) & 1) * H;
                int off[4];
                unsigned mask = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    off[r] = pb[r] + par;
                    mask |= st < pl[r] ? 1u << r : 0u;
                }
                if (mask) fail = !lstm_poll_n<4>(ring, off, pd, mask, (unsigned)st, s.abort_word, hs) || fail;
            }
            for (int e0 = tid + 4 * 512; e0 < NB * H; e0 += 4 * 512) {   // (H = 512 with 8 utterances per group: a second round)
                int off[4], dst[4];
                unsigned mask = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int e = e0 + r * 512;
                    off[r] = dst[r] = 0;
                    if (e < NB * H) {
                        const int q = e / H, i = e - q * H;
                        int lq = 0;
#pragma unroll
                        for (int r2 = 0; r2 < NB; ++r2) lq = (r2 == q) ? len[r2] : lq;
                        if (st < lq) {
                            off[r] = (int)(((size_t)((b0 + q) * a.ndir + dir) * 2 + (size_t)((st - 1) & 1)) * H + i);
                            dst[r] = i * NB + q;
                            mask |= 1u << r;
                        }
                    }
                }
                if (mask) fail = !lstm_poll_n<4>(ring, off, dst, mask, (unsigned)st, s.abort_word, hs) || fail;
            }
        }
        if (__syncthreads_or(fail)) return;
        lstm_f2 acc[NB / 2][4];
#pragma unroll
        for (int p = 0; p < NB / 2; ++p)
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[p][g] = lstm_f2{0.f, 0.f};
        {
            const lstm_f2* h2 = reinterpret_cast<const lstm_f2*>(hs + (size_t)ks * KL * NB);   // [k][NB / 2] pairs
#pragma unroll
            for (int kb = 0; kb < KL / 2; ++kb) {
                lstm_f2 ha[NB / 2], hb[NB / 2];
#pragma unroll
                for (int p = 0; p < NB / 2; ++p) {
                    ha[p] = h2[(2 * kb) * (NB / 2) + p];
                    hb[p] = h2[(2 * kb + 1) * (NB / 2) + p];
                }
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int p = 0; p < NB / 2; ++p) {
                        lstm_pkfma_lo(acc[p][g], w[g][kb], ha[p]);
                        lstm_pkfma_hi(acc[p][g], w[g][kb], hb[p]);
                    }
            }
        }
#pragma unroll
        for (int p = 0; p < NB / 2; ++p)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                part[(((2 * p) * KS + ks) * 4 + g) * HU + u] = acc[p][g].x;
                part[(((2 * p + 1) * KS + ks) * 4 + g) * HU + u] = acc[p][g].y;
            }
        __syncthreads();
        if (acting) {
            const float* pq = part + (size_t)aq * KS * 4 * HU;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < GPR) {
                    const int g = gset + i * NG;
                    float gsum = xv[i];
                    for (int q = 0; q < KS; ++q) gsum += pq[(q * 4 + g) * HU + u];
                    act[(aq * 4 + g) * HU + u] = g == 2 ? ttsc_tanhf(gsum) : ttsc_sigmoidf(gsum);
                }
        }
        __syncthreads();
        if (mine) {
            const float* aw = act + (size_t)ks * 4 * HU + u;
            const float ig = aw[0], fg = aw[HU], gg = aw[2 * HU], og = aw[3 * HU];
            c = fmaf(fg, c, ig * gg);
            hlast = og * ttsc_tanhf(c);
            __hip_atomic_store(org + (size_t)(st & 1) * H + j, ((lstm_u64)(unsigned)(st + 1) << 32) | (lstm_u64)__float_as_uint(hlast), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            yb[(size_t)tpos * a.ldy + j] = hlast;
            if (a.gates_out) {
                float* gp = a.gates_out + ((size_t)ob * a.T + tpos) * ((size_t)a.ndir * H4) + (size_t)dir * H4 + j;
                gp[0] = ig;
                gp[H] = fg;
                gp[2 * H] = gg;
                gp[3 * H] = og;
                a.c_out[((size_t)ob * a.T + tpos) * ((size_t)a.ndir * H) + (size_t)dir * H + j] = c;
            }
        }
    }
    if (owner) {
        if (a.h_n) a.h_n[((size_t)dir * a.B + ob) * H + j] = hlast;
        if (a.c_n) a.c_n[((size_t)dir * a.B + ob) * H + j] = c;
    }
}

__global__ __launch_bounds__(512) void lstm_bwd_split_kernel(LstmSplitArgs s) {
    extern __shared__ __attribute__((aligned(16))) float sm[];   // dg[4H] | part[KS][HU]
    __shared__ int ok_s;
    const LstmBwdArgs& a = s.bw;
    const int H = a.H, H4 = 4 * H, HU = s.HU, KS = s.KS;
    const int tid = threadIdx.x, u = tid % HU, ks = tid / HU;
    const int m = blockIdx.x, b = blockIdx.y, dir = blockIdx.z;
    const int j = m * HU + u;
    const int KL = H4 / KS;
    float* dg = sm;
    float* part = sm + H4;
    unsigned* cnt = s.cnt + (b * a.ndir + dir);
    const bool owner = ks == 0;
    const int len = a.lengths ? a.lengths[b] : a.T;
    const float* whhT = a.whhT + (size_t)dir * H * H4;
    const size_t gstride = (size_t)a.ndir * H4, cstride = (size_t)a.ndir * H;
    const float* gb = a.gates + (size_t)b * a.T * gstride + (size_t)dir * H4 + j;
    const float* cb = a.cst + (size_t)b * a.T * cstride + (size_t)dir * H + j;
    const float* dyb = a.dy + (size_t)b * a.T * a.ldy + a.yoff + dir * H + j;
    float* dgrow = a.dgates + (size_t)b * a.T * gstride + (size_t)dir * H4;
    if (owner)
        for (int t = len; t < a.T; ++t) {
            float* p = dgrow + (size_t)t * gstride + j;
            p[0] = 0.f;
            p[H] = 0.f;
            p[2 * H] = 0.f;
            p[3 * H] = 0.f;
        }
    float dh_rec = 0.f, dc_next = 0.f;
    unsigned step = 0;
    for (int st = len - 1; st >= 0; --st) {
        const int tpos = dir == 0 ? st : (len - 1 - st);
        const int tprev = dir == 0 ? tpos - 1 : tpos + 1;
        if (owner) {
            const float* g = gb + (size_t)tpos * gstride;
            const float ig = g[0], fg = g[H], gg = g[2 * H], og = g[3 * H];
            const float ct = cb[(size_t)tpos * cstride];
            const float cp = st > 0 ? cb[(size_t)tprev * cstride] : 0.f;
            const float dh = dyb[(size_t)tpos * a.ldy] + dh_rec;
            const float tc = ttsc_tanhf(ct);
            const float d_o = dh * tc * og * (1.f - og);
            const float dc = dc_next + dh * og * (1.f - tc * tc);
            const float d_i = dc * gg * ig * (1.f - ig);
            const float d_g = dc * ig * (1.f - gg * gg);
            const float d_f = dc * cp * fg * (1.f - fg);
            dc_next = dc * fg;
            float* p = dgrow + (size_t)tpos * gstride + j;
            g_st(p, d_i);
            g_st(p + H, d_f);
            g_st(p + 2 * H, d_g);
            g_st(p + 3 * H, d_o);
        }
        g_publish(cnt);
        ++step;
        if (st == 0) break;
        if (!g_wait(cnt, step * (unsigned)s.G, s.abort_word, &ok_s)) return;
        for (int i = tid; i < H4; i += 512) dg[i] = g_ld(dgrow + (size_t)tpos * gstride + i);
        __syncthreads();
        float acc[1][1] = {{0.f}};
        lstm_chain<1, 1, 4>(acc, whhT + (size_t)(ks * KL / 4) * H * 4, H, 0, j, dg + ks * KL, H4, KL);
        part[ks * HU + u] = acc[0][0];
        __syncthreads();
        if (owner) {
            float v = 0.f;
            for (int q = 0; q < KS; ++q) v += part[q * HU + u];
            dh_rec = v;
        }
        __syncthreads();
    }
}

// Backward-through-time with the member's slice of W_hh^T RESIDENT IN REGISTERS (KL = 4H / KS = 128 gate rows per thread: H = 256 over 4 members) —
// the counterpart of lstm_seq_split_res_kernel.  lstm_bwd_split_kernel above streams 256 KB of W_hh^T per member and step from L2, hands the gate
// gradients over through the output tensor plus a counter (two round trips) and loads the saved gates of a step only when it gets there: 2.9 us per
// step against the forward's 0.85.  Here the weights are loaded once, the gate gradients travel as 8-byte {value, step tag} granules in a
// two-slot ring that the consumers poll directly, and the owner threads fetch the saved gates / cell states / dy of step t-1 before they wait for
// the exchange of step t.  Chain order = the streaming kernel's (k ascending per slice, slices added in order): results are bit-identical to it.
template <int KL>
__global__ __launch_bounds__(512) void lstm_bwd_split_res_kernel(LstmSplitArgs s, lstm_u64* ring) {
    extern __shared__ __attribute__((aligned(16))) float sm[];   // dg[4H] | part[KS][HU]
    const LstmBwdArgs& a = s.bw;
    const int H = a.H, H4 = 4 * H, HU = s.HU, KS = s.KS;
    const int tid = threadIdx.x, u = tid % HU, ks = tid / HU;
    const int m = blockIdx.x, b = blockIdx.y, dir = blockIdx.z;
    const int j = m * HU + u;
    float* dg = sm;
    float* part = sm + H4;
    lstm_u64* rg = ring + (size_t)(b * a.ndir + dir) * 2 * H4;
    const bool owner = ks == 0;
    const int len = a.lengths ? a.lengths[b] : a.T;
    const size_t gstride = (size_t)a.ndir * H4, cstride = (size_t)a.ndir * H;
    const float* gb = a.gates + (size_t)b * a.T * gstride + (size_t)dir * H4 + j;
    const float* cb = a.cst + (size_t)b * a.T * cstride + (size_t)dir * H + j;
    const float* dyb = a.dy + (size_t)b * a.T * a.ldy + a.yoff + dir * H + j;
    float* dgrow = a.dgates + (size_t)b * a.T * gstride + (size_t)dir * H4;
    float w[KL];
    {
        // transposed pack [4H/4][H][4]: k-block kb of this slice holds gate rows ks*KL + 4*kb .. + 3 of column j
        const float4* w4 = reinterpret_cast<const float4*>(a.whhT + (size_t)dir * H * H4 + (size_t)(ks * (KL / 4)) * H * 4) + j;
#pragma unroll
        for (int kb = 0; kb < KL / 4; ++kb) {
            const float4 v = w4[(size_t)kb * H];
            w[4 * kb] = v.x;
            w[4 * kb + 1] = v.y;
            w[4 * kb + 2] = v.z;
            w[4 * kb + 3] = v.w;
        }
    }
    if (owner)
        for (int t = len; t < a.T; ++t) {
            float* p = dgrow + (size_t)t * gstride + j;
            p[0] = 0.f;
            p[H] = 0.f;
            p[2 * H] = 0.f;
            p[3 * H] = 0.f;
        }
    float dh_rec = 0.f, dc_next = 0.f;
    float pg[4] = {0.f, 0.f, 0.f, 0.f}, pct = 0.f, pcp = 0.f, pdy = 0.f;   // saved state of the step about to be differentiated
    auto fetch = [&](int st) __attribute__((always_inline)) {
        const int tpos = dir == 0 ? st : (len - 1 - st);
        const int tprev = dir == 0 ? tpos - 1 : tpos + 1;
        const float* g = gb + (size_t)tpos * gstride;
        pg[0] = g[0];
        pg[1] = g[H];
        pg[2] = g[2 * H];
        pg[3] = g[3 * H];
        pct = cb[(size_t)tpos * cstride];
        pcp = st > 0 ? cb[(size_t)tprev * cstride] : 0.f;
        pdy = dyb[(size_t)tpos * a.ldy];
    };
    if (owner && len > 0) fetch(len - 1);
    for (int st = len - 1; st >= 0; --st) {
        const int tpos = dir == 0 ? st : (len - 1 - st);
        const unsigned tag = (unsigned)(len - st);
        lstm_u64* slot = rg + (size_t)(st & 1) * H4;
        if (owner) {
            const float ig = pg[0], fg = pg[1], gg = pg[2], og = pg[3];
            const float dh = pdy + dh_rec;
            const float tc = ttsc_tanhf(pct);
            const float d_o = dh * tc * og * (1.f - og);
            const float dc = dc_next + dh * og * (1.f - tc * tc);
            const float d_i = dc * gg * ig * (1.f - ig);
            const float d_g = dc * ig * (1.f - gg * gg);
            const float d_f = dc * pcp * fg * (1.f - fg);
            dc_next = dc * fg;
            if (st > 0) {   // hand-off first: the other members are waiting for these four values
                const lstm_u64 tg = (lstm_u64)tag << 32;
                __hip_atomic_store(slot + j, tg | (lstm_u64)__float_as_uint(d_i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(slot + H + j, tg | (lstm_u64)__float_as_uint(d_f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(slot + 2 * H + j, tg | (lstm_u64)__float_as_uint(d_g), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(slot + 3 * H + j, tg | (lstm_u64)__float_as_uint(d_o), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            float* p = dgrow + (size_t)tpos * gstride + j;   // (read by the weight-gradient GEMMs after this kernel)
            p[0] = d_i;
            p[H] = d_f;
            p[2 * H] = d_g;
            p[3 * H] = d_o;
        }
        if (st == 0) break;
        if (owner) fetch(st - 1);   // in flight while the exchange completes
        bool fail = false;
        for (int i0 = tid; i0 < H4; i0 += 4 * 512) {   // (H = 256: both granules of a thread in one round trip)
            int off[4];
            unsigned mask = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                off[r] = i0 + r * 512;
                mask |= off[r] < H4 ? 1u << r : 0u;
            }
            fail = !lstm_poll_n<4>(slot, off, off, mask, tag, s.abort_word, dg) || fail;
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
            float v = 0.f;
            for (int q = 0; q < KS; ++q) v += part[q * HU + u];
            dh_rec = v;
        }
    }
}

// [ndir][4H][H] (torch weight_hh layout, device)  ->  forward pack [ndir][H/4][4H][4]  or  transposed pack [ndir][4H/4][H][4]
__global__ void lstm_pack_kernel(const float* __restrict__ w, float* __restrict__ out, int ndir, int H, int transpose) {
    const long per = (long)4 * H * H;
    const long total = per * ndir;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int d = (int)(i / per);
        const long e = i - (long)d * per;
        const int q = (int)(e & 3);
        const long t = e >> 2;
        float v;
        if (!transpose) {   // out[d][k/4][r][k%4] = w[d][r][k]
            const int r = (int)(t % (4 * H)), k4 = (int)(t / (4 * H));
            v = w[(size_t)d * per + (size_t)r * H + 4 * k4 + q];
        } else {            // out[d][r/4][k][r%4] = w[d][r][k]
            const int k = (int)(t % H), r4 = (int)(t / H);
            v = w[(size_t)d * per + (size_t)(4 * r4 + q) * H + k];
        }
        out[i] = v;
    }
}

}  // namespace ttsc

using namespace ttsc;

// Recurrent weights: host torch layout weight_hh [4H, H] per direction -> device [ndir][H/4][4H][4]
extern "C" int ttsc_lstm_pack_whh(const float* whh_host, int32_t ndir, int32_t H, float** whh_dev_out) {
    TTSC_REQUIRE(whh_host && whh_dev_out, "ttsc_lstm_pack_whh: null argument");
    TTSC_REQUIRE(ndir >= 1 && ndir <= 2 && H >= 4 && H <= 512 && H % 4 == 0, "ttsc_lstm_pack_whh: need ndir in {1,2}, H %% 4 == 0, H <= 512 (got %d, %d)", ndir, H);
    const size_t per = (size_t)4 * H * H;
    std::vector<float> t(per * ndir);
    for (int d = 0; d < ndir; ++d)
        for (int r = 0; r < 4 * H; ++r)
            for (int k = 0; k < H; ++k) t[d * per + ((size_t)(k >> 2) * 4 * H + r) * 4 + (k & 3)] = whh_host[d * per + (size_t)r * H + k];
    float* dptr = nullptr;
    TTSC_HIP_CHECK(hipMalloc((void**)&dptr, t.size() * sizeof(float)));
    TTSC_HIP_CHECK(hipMemcpy(dptr, t.data(), t.size() * sizeof(float), hipMemcpyHostToDevice));
    *whh_dev_out = dptr;
    return TTSC_OK;
}

extern "C" void ttsc_device_free(void* p) {
    if (p) (void)hipFree(p);
}


// members per (utterance, direction) for the split kernels: power of two <= 4, >= 32 units per member, all workgroups resident
static int lstm_split_members(int B, int ndir, int H) {
    const int cus = device_cus();   // of the CURRENT device
    int gmax = 4;
    if (const char* ev = getenv("TTSC_LSTM_SPLIT")) gmax = atoi(ev);
    int G = 1;
    while (G * 2 <= gmax && (long)G * 2 * B * ndir <= cus && H % (G * 2) == 0 && H / (G * 2) >= 32 && 512 % (H / (G * 2)) == 0) {
        const int HU = H / (G * 2), KS = 512 / HU;
        if (H % KS != 0 || (H / KS) % 8 != 0 || (4 * H / KS) % 16 != 0) break;
        G *= 2;
    }
    return G;
}

// Counters / abort words / granule ring of the split recurrences: one area per (device, stream) — common.hpp HandoffArea
// counters, this launch's abort word and (zero_ring) the granule ring restart at zero — ONE small launch.  Two hipMemsetAsync calls were THREE
// fill kernels per recurrence launch (the runtime splits the 32 772-byte counter block into an aligned part and a 4-byte tail): at B = 1 that is 14 us
// of a 150 us layer, eight layers per sentence (profiles/r06_e2e_single_sentence_launches.txt)
__global__ __launch_bounds__(256) void lstm_rearm_kernel(unsigned* __restrict__ words, unsigned nwords, lstm_u64* __restrict__ ring, size_t n64) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nwords; i += stride) words[i] = 0u;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n64; i += stride) ring[i] = 0ull;
}

static HandoffArea* lstm_area(hipStream_t s, size_t ring_bytes, bool zero_ring = false) {
    HandoffArea* ar = handoff_area("lstm", s, 8192, ring_bytes);
    if (!ar) return nullptr;
    static const bool by_memset = getenv("TTSC_LSTM_REARM") && !strcmp(getenv("TTSC_LSTM_REARM"), "memset");   // (measurement switch: round 5's fills)
    if (by_memset) {
        if (ar->rearm(s) != hipSuccess) return nullptr;
        if (zero_ring && ring_bytes && hipMemsetAsync(ar->buf, 0, ring_bytes, s) != hipSuccess) return nullptr;
        return ar;
    }
    const size_t n64 = zero_ring ? ring_bytes / sizeof(lstm_u64) : 0;
    const size_t work = std::max<size_t>(ar->nwords + 1, n64);
    const unsigned blocks = (unsigned)std::min<size_t>(std::max<size_t>((work + 2047) / 2048, 1), 2048);
    hipLaunchKernelGGL(lstm_rearm_kernel, dim3(blocks), dim3(256), 0, s, ar->words, (unsigned)(ar->nwords + 1), reinterpret_cast<lstm_u64*>(ar->buf), n64);
    if (hipGetLastError() != hipSuccess) return nullptr;
    return ar;
}

// 0 = every hand-off of the split LSTM launches on this device since the last call completed, 1 = a bounded spin timed out.
// Synchronises the device.
extern "C" int32_t ttsc_lstm_split_status(void) { return handoff_status("lstm"); }

// utterances per member group of the register-resident split recurrence: 0 = automatic (see lstm_forward_impl)
static std::atomic<int> g_lstm_group_size{getenv("TTSC_LSTM_NB") ? atoi(getenv("TTSC_LSTM_NB")) : 0};

extern "C" int32_t ttsc_lstm_set_group_size(int32_t n) {
    if (n < 0 || n > 8 || (n & (n - 1)) != 0) return -1;
    return g_lstm_group_size.exchange(n, std::memory_order_relaxed);
}



static int lstm_forward_impl(const float* xg_dev, const float* whh_packed_dev, float* y_dev, const int32_t* lengths_dev,
                             int32_t B, int32_t T, int32_t H, int32_t ndir, int64_t ldy, int32_t yoff, const float* h0_dev,
                             const float* c0_dev, float* hn_dev, float* cn_dev, float* gates_dev, float* c_dev, void* stream);

extern "C" int ttsc_lstm_seq_forward(const float* xg_dev, const float* whh_packed_dev, float* y_dev, const int32_t* lengths_dev,
                                     int32_t B, int32_t T, int32_t H, int32_t ndir, int64_t ldy, int32_t yoff, const float* h0_dev,
                                     const float* c0_dev, float* hn_dev, float* cn_dev, void* stream) {
    return lstm_forward_impl(xg_dev, whh_packed_dev, y_dev, lengths_dev, B, T, H, ndir, ldy, yoff, h0_dev, c0_dev, hn_dev, cn_dev,
                             nullptr, nullptr, stream);
}

extern "C" int ttsc_lstm_seq_forward_train(const float* xg_dev, const float* whh_packed_dev, float* y_dev, const int32_t* lengths_dev,
                                           int32_t B, int32_t T, int32_t H, int32_t ndir, int64_t ldy, int32_t yoff, float* gates_dev,
                                           float* c_dev, void* stream) {
    TTSC_REQUIRE(gates_dev && c_dev, "ttsc_lstm_seq_forward_train: null argument");
    return lstm_forward_impl(xg_dev, whh_packed_dev, y_dev, lengths_dev, B, T, H, ndir, ldy, yoff, nullptr, nullptr, nullptr, nullptr,
                             gates_dev, c_dev, stream);
}

extern "C" int ttsc_lstm_pack_whh_device(const float* whh_dev, int32_t ndir, int32_t H, int32_t transpose, float* out_dev, void* stream) {
    TTSC_REQUIRE(whh_dev && out_dev, "ttsc_lstm_pack_whh_device: null argument");
    TTSC_REQUIRE(ndir >= 1 && ndir <= 2 && H >= 4 && H <= 512 && H % 4 == 0, "ttsc_lstm_pack_whh_device: need ndir in {1,2}, H %% 4 == 0, H <= 512 (got %d, %d)", ndir, H);
    const long total = (long)4 * H * H * ndir;
    hipLaunchKernelGGL(lstm_pack_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 2048)), dim3(256), 0, (hipStream_t)stream, whh_dev, out_dev,
                       ndir, H, transpose);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("lstm_pack_kernel launch failed: %s", hipGetErrorString(e));
        return TTSC_EHIP;
    }
    return TTSC_OK;
}

extern "C" int ttsc_lstm_seq_backward(const float* dy_dev, const float* gates_dev, const float* c_dev, const float* whhT_packed_dev,
                                      float* dgates_dev, const int32_t* lengths_dev, int32_t B, int32_t T, int32_t H, int32_t ndir,
                                      int64_t ldy, int32_t yoff, void* stream) {
    TTSC_REQUIRE(dy_dev && gates_dev && c_dev && whhT_packed_dev && dgates_dev, "ttsc_lstm_seq_backward: null argument");
    TTSC_REQUIRE(B > 0 && T > 0 && ndir >= 1 && ndir <= 2 && H >= 4 && H <= 512 && H % 4 == 0, "ttsc_lstm_seq_backward: bad shape B=%d T=%d H=%d ndir=%d", B, T, H, ndir);
    TTSC_REQUIRE(ldy >= (int64_t)yoff + (int64_t)ndir * H, "ttsc_lstm_seq_backward: ldy too small");
    LstmBwdArgs a{dy_dev, gates_dev, c_dev, whhT_packed_dev, dgates_dev, lengths_dev, B, T, H, ndir, (int)ldy, yoff};
    const int G = lstm_split_members(B, ndir, H);
    if (G > 1) {
        TTSC_REQUIRE(B * ndir <= 8192, "ttsc_lstm_seq_backward: too many sequences for the split kernel");
        LstmSplitArgs sa{};
        sa.bw = a;
        sa.G = G;
        sa.HU = H / G;
        sa.KS = 512 / sa.HU;
        const size_t lds = ((size_t)4 * H + (size_t)sa.KS * sa.HU) * sizeof(float);
        static const bool bwd_resident = !(getenv("TTSC_LSTM_BWD_RESIDENT") && atoi(getenv("TTSC_LSTM_BWD_RESIDENT")) == 0);
        if (bwd_resident && 4 * H / sa.KS == 128) {   // H = 256 over 4 members: 128 rows of W_hh^T per thread stay in registers, granule hand-off
            const size_t ring_bytes = (size_t)B * ndir * 2 * 4 * H * sizeof(lstm_u64);
            HandoffArea* ar2 = lstm_area((hipStream_t)stream, ring_bytes, true);
            TTSC_REQUIRE(ar2, "ttsc_lstm_seq_backward: cannot allocate the hand-off ring");
            sa.cnt = ar2->words;
            sa.abort_word = ar2->abort_word();
            lstm_u64* ring = reinterpret_cast<lstm_u64*>(ar2->buf);
            hipLaunchKernelGGL(lstm_bwd_split_res_kernel<128>, dim3((unsigned)G, (unsigned)B, (unsigned)ndir), dim3(512), lds, (hipStream_t)stream, sa, ring);
        } else {
            HandoffArea* ar = lstm_area((hipStream_t)stream, 0);
            TTSC_REQUIRE(ar, "ttsc_lstm_seq_backward: cannot allocate the hand-off counters");
            sa.cnt = ar->words;
            sa.abort_word = ar->abort_word();
            hipLaunchKernelGGL(lstm_bwd_split_kernel, dim3((unsigned)G, (unsigned)B, (unsigned)ndir), dim3(512), lds, (hipStream_t)stream, sa);
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) {
            set_error("lstm_bwd_split_kernel launch failed: %s", hipGetErrorString(e));
            return TTSC_EHIP;
        }
        return TTSC_OK;
    }
    static const bool bwd_res1 = !(getenv("TTSC_LSTM_BWD_RESIDENT") && atoi(getenv("TTSC_LSTM_BWD_RESIDENT")) == 0);
    if (bwd_res1 && (H == 64 || H == 128)) {   // the slice's column of W_hh^T in registers, four k-slices per unit
        if (H == 64)
            hipLaunchKernelGGL(lstm_bwd_resident_kernel<64>, dim3((unsigned)B, (unsigned)ndir), dim3(256), 0, (hipStream_t)stream, a);
        else
            hipLaunchKernelGGL(lstm_bwd_resident_kernel<128>, dim3((unsigned)B, (unsigned)ndir), dim3(512), 0, (hipStream_t)stream, a);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) {
            set_error("lstm_bwd_resident_kernel launch failed: %s", hipGetErrorString(e));
            return TTSC_EHIP;
        }
        return TTSC_OK;
    }
    const int threads = (int)round_up(H, 64);
    hipLaunchKernelGGL(lstm_bwd_kernel, dim3((unsigned)B, (unsigned)ndir), dim3(threads), (size_t)4 * H * sizeof(float), (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("lstm_bwd_kernel launch failed: %s", hipGetErrorString(e));
        return TTSC_EHIP;
    }
    return TTSC_OK;
}

static int lstm_forward_impl(const float* xg_dev, const float* whh_packed_dev, float* y_dev, const int32_t* lengths_dev,
                             int32_t B, int32_t T, int32_t H, int32_t ndir, int64_t ldy, int32_t yoff, const float* h0_dev,
                             const float* c0_dev, float* hn_dev, float* cn_dev, float* gates_dev, float* c_dev, void* stream) {
    TTSC_REQUIRE(xg_dev && whh_packed_dev && y_dev, "ttsc_lstm_seq_forward: null argument");
    TTSC_REQUIRE(B > 0 && T > 0 && ndir >= 1 && ndir <= 2 && H >= 4 && H <= 512 && H % 4 == 0, "ttsc_lstm_seq_forward: bad shape B=%d T=%d H=%d ndir=%d", B, T, H, ndir);
    TTSC_REQUIRE(ldy >= (int64_t)yoff + (int64_t)ndir * H, "ttsc_lstm_seq_forward: ldy too small");
    LstmArgs a{xg_dev, whh_packed_dev, y_dev, lengths_dev, hn_dev, cn_dev, h0_dev, c0_dev, B, T, H, ndir, (int)ldy, yoff, gates_dev, c_dev};
    // The split changes the summation order with G, so a result is bit-reproducible only among launches that pick the same
    // G: always the case up to 64 (utterance, direction) pairs per launch (G = 4), e.g. a padded batch of <= 32 sentences
    // against the same sentences run alone (tests/test_lstm_gpu.py, tests/test_api_gpu.py); larger batches (G = 2 / 1)
    // agree to ~1e-6 relative.  TTSC_LSTM_SPLIT_INFER=0 keeps inference on the single-workgroup kernel.
    static const bool split_infer = !(getenv("TTSC_LSTM_SPLIT_INFER") && atoi(getenv("TTSC_LSTM_SPLIT_INFER")) == 0);
    if (H == 64 || H == 128) {   // W_hh resident in registers, one thread per gate row (same kernel for every batch size)
        dim3 grid((unsigned)B, (unsigned)ndir);
        if (H == 64)
            hipLaunchKernelGGL(lstm_seq_resident_kernel<64>, grid, dim3(256), 0, (hipStream_t)stream, a);
        else
            hipLaunchKernelGGL(lstm_seq_resident_kernel<128>, grid, dim3(512), 0, (hipStream_t)stream, a);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) {
            set_error("lstm_seq_resident_kernel launch failed: %s", hipGetErrorString(e));
            return TTSC_EHIP;
        }
        return TTSC_OK;
    }
    static const bool resident = !(getenv("TTSC_LSTM_RESIDENT") && atoi(getenv("TTSC_LSTM_RESIDENT")) == 0);
    if ((gates_dev || split_infer) && resident && (H == 256 || H == 512)) {
        // H = 256 / 512: G = 4 / 16 members per (utterance, direction), each thread holding 128 weights of W_hh in registers
        // (lstm_seq_split_res_kernel: 512 threads = HU units x KS k-slices of 32).  All members of a launch must be resident, so a
        // launch takes cus / G pairs; up to three consecutive launches still beat the streaming kernels (H = 256: 3 us per step
        // and layer per launch against 9.8), and every batch size up to 3 * cus / (2 G) sentences then sums in the same order as
        // a sentence run alone.
        const int cus = device_cus();
        if (cus >= 4) {
            const int Gm = H == 256 ? 4 : 16;
            const int pairs = B * ndir, cap = cus / Gm;
            // utterances per member group (lstm_seq_split_res_nb_kernel: per utterance the same arithmetic as NB = 1).  Default: the smallest of
            // 1 / 2 / 4 that fits the batch into ONE launch (shortest step), else 4 and up to three launches.  ttsc_lstm_set_group_size(n) asks
            // for n per group whenever the batch has that many: fewer CUs held for a somewhat longer step — what a caller wants who runs the
            // recurrence beside a kernel that fills the chip (Cubegan.inference_pipelined).
            const int pref = g_lstm_group_size.load(std::memory_order_relaxed);
            int NB = 1;
            if (pref > 0) {
                while (NB * 2 <= pref && NB * 2 <= B) NB *= 2;
                while (NB < 8 && ((B + NB - 1) / NB) * ndir > 3 * cap) NB *= 2;
            } else {
                while (NB < 4 && ((B + NB - 1) / NB) * ndir > cap) NB *= 2;
            }
            const int groups = ((B + NB - 1) / NB) * ndir;
            if (cap >= 1 && groups <= 3 * cap && pairs <= 16384) {
                // granule ring [pairs][2 slots][H]: sized for this launch, grown on demand, one per (device, stream)
                HandoffArea* ar = lstm_area((hipStream_t)stream, (size_t)pairs * 2 * H * sizeof(lstm_u64), true);
                if (!ar) {
                    set_error("ttsc_lstm_seq_forward: cannot allocate the hand-off counters / ring");
                    return TTSC_ENOMEM;
                }
                lstm_u64* ring = reinterpret_cast<lstm_u64*>(ar->buf);
                LstmSplitArgs sa{};
                sa.f = a;
                sa.cnt = ar->words;
                sa.abort_word = ar->abort_word();
                sa.G = Gm;
                sa.HU = H / Gm;
                sa.KS = 512 / sa.HU;
                const size_t lds = ((size_t)NB * ((size_t)H + (size_t)sa.KS * 4 * sa.HU) + (size_t)NB * 4 * sa.HU) * sizeof(float);
                for (int p0 = 0; p0 < groups; p0 += cap) {
                    sa.p0 = p0;
                    const int n = groups - p0 < cap ? groups - p0 : cap;
                    const dim3 grid((unsigned)Gm, (unsigned)n, 1u);
                    if (NB == 1)
                        hipLaunchKernelGGL(lstm_seq_split_res_kernel<32>, grid, dim3(512), lds, (hipStream_t)stream, sa, ring);
                    else if (NB == 2)
                        hipLaunchKernelGGL((lstm_seq_split_res_nb_kernel<32, 2>), grid, dim3(512), lds, (hipStream_t)stream, sa, ring);
                    else if (NB == 4)
                        hipLaunchKernelGGL((lstm_seq_split_res_nb_kernel<32, 4>), grid, dim3(512), lds, (hipStream_t)stream, sa, ring);
                    else
                        hipLaunchKernelGGL((lstm_seq_split_res_nb_kernel<32, 8>), grid, dim3(512), lds, (hipStream_t)stream, sa, ring);
                }
                hipError_t e = hipGetLastError();
                if (e != hipSuccess) {
                    set_error("lstm_seq_split_res_kernel launch failed: %s", hipGetErrorString(e));
                    return TTSC_EHIP;
                }
                return TTSC_OK;
            }
        }
    }
    const int G = (gates_dev || split_infer) ? lstm_split_members(B, ndir, H) : 1;
    if (G > 1) {
        TTSC_REQUIRE(B * ndir <= 8192, "ttsc_lstm_seq_forward: too many sequences for the split kernel");
        HandoffArea* ar = lstm_area((hipStream_t)stream, 0);
        TTSC_REQUIRE(ar, "ttsc_lstm_seq_forward: cannot allocate the hand-off counters");
        LstmSplitArgs sa{};
        sa.f = a;
        sa.cnt = ar->words;
        sa.abort_word = ar->abort_word();
        sa.G = G;
        sa.HU = H / G;
        sa.KS = 512 / sa.HU;
        const size_t lds = ((size_t)H + (size_t)sa.KS * 4 * sa.HU) * sizeof(float);
        hipLaunchKernelGGL(lstm_seq_split_kernel, dim3((unsigned)G, (unsigned)B, (unsigned)ndir), dim3(512), lds, (hipStream_t)stream, sa);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) {
            set_error("lstm_seq_split_kernel launch failed: %s", hipGetErrorString(e));
            return TTSC_EHIP;
        }
        return TTSC_OK;
    }
    const int threads = (int)round_up(H, 64);
    const int bt = B * ndir > 512 ? 2 : 1;
    dim3 grid((unsigned)ceil_div(B, bt), (unsigned)ndir);
    const size_t lds = (size_t)2 * bt * H * sizeof(float);
    if (bt == 1)
        hipLaunchKernelGGL(lstm_seq_kernel<1>, grid, dim3(threads), lds, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(lstm_seq_kernel<2>, grid, dim3(threads), lds, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("lstm_seq_kernel launch failed: %s", hipGetErrorString(e));
        return TTSC_EHIP;
    }
    return TTSC_OK;
}
