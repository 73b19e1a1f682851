// Small training-side kernels of the generator — gfx950.
//
// The reference wraps every generator convolution in torch.nn.utils.weight_norm [EXTERNAL hifigan/models.py] and lets
// torch autograd differentiate  w = g * v / ||v||  and the bias add as a dozen elementwise / reduction launches per layer.
// At the crop sizes of `Cubegan.training_step` (cube/networks/cubegan.py:116-134: 50 frames) those ~5 us launches add up
// to a third of the step, so each becomes one HBM-bound kernel here:
//   ttsc_weight_norm_forward    w[r,:] = v[r,:] * (g[r] / ||v[r,:]||),  norm[r] = ||v[r,:]||          (one workgroup per row)
//   ttsc_weight_norm_backward   dg[r] = <dw,v> / n,  dv = (g/n) * dw - v * (g * <dw,v> / n^3)
//   ttsc_bias_grad              db[c] = sum_{b,t} dy[b,c,t]   (fixed-order two-level sum, last workgroup of a channel finishes)
#include <algorithm>
#include <cmath>
#include <cstdint>

#include "common.hpp"

namespace ttsc {

__device__ __forceinline__ float block_sum(float v, float* red) {
    // fixed-order tree: wave shuffle, then the waves through LDS
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float s = 0.f;
    const int nw = blockDim.x >> 6;
    for (int w = 0; w < nw; ++w) s += red[w];
    return s;
}

__global__ __launch_bounds__(256) void wn_fwd_kernel(const float* __restrict__ v, const float* __restrict__ g, float* __restrict__ w,
                                                     float* __restrict__ norm, int C) {
    __shared__ float red[4];
    const int r = blockIdx.x;
    const float* vr = v + (size_t)r * C;
    float s = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) s = fmaf(vr[c], vr[c], s);
    const float n = sqrtf(block_sum(s, red));
    const float k = g[r] / n;
    for (int c = threadIdx.x; c < C; c += 256) w[(size_t)r * C + c] = vr[c] * k;
    if (threadIdx.x == 0) norm[r] = n;
}

__global__ __launch_bounds__(256) void wn_bwd_kernel(const float* __restrict__ dw, const float* __restrict__ v, const float* __restrict__ g,
                                                     const float* __restrict__ norm, float* __restrict__ dv, float* __restrict__ dg, int C) {
    __shared__ float red[4];
    const int r = blockIdx.x;
    const float* vr = v + (size_t)r * C;
    const float* dr = dw + (size_t)r * C;
    float s = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) s = fmaf(dr[c], vr[c], s);
    s = block_sum(s, red);
    const float n = norm[r], gr = g[r];
    const float k1 = gr / n, k2 = gr * s / (n * n * n);
    for (int c = threadIdx.x; c < C; c += 256) dv[(size_t)r * C + c] = k1 * dr[c] - k2 * vr[c];
    if (threadIdx.x == 0) dg[r] = s / n;
}

// db[c] = sum over (b, t): grid (C, S); block s sums its contiguous share of the B*L positions, writes part[c][s], and the
// LAST block of channel c to finish (ticket counter) adds the S partials in index order -> deterministic, one launch.
__global__ __launch_bounds__(256) void bias_grad_kernel(const float* __restrict__ dy, float* __restrict__ db, float* __restrict__ part,
                                                        unsigned* __restrict__ ticket, int B, int C, int L, int S) {
    __shared__ float red[4];
    __shared__ bool last;
    const int c = blockIdx.x, s = blockIdx.y;
    // block s owns the time slice [t0, t1) of every batch item: rows of contiguous floats, no index division, four
    // independent accumulators so the loads of a row overlap
    const int per = (L + S - 1) / S;
    const int t0 = s * per, t1 = min(t0 + per, L);
    // flattened (batch item, position of the slice) index space, four independent loads per thread and trip
    const int w = t1 - t0;
    const int n = B * w;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    auto at = [&](int i) {
        const int b = i / w, t = t0 + (i - b * w);
        return dy[((size_t)b * C + c) * L + t];
    };
    int i = threadIdx.x;
    for (; i + 768 < n; i += 1024) {
        const float v0 = at(i), v1 = at(i + 256), v2 = at(i + 512), v3 = at(i + 768);
        a0 += v0;
        a1 += v1;
        a2 += v2;
        a3 += v3;
    }
    for (; i < n; i += 256) a0 += at(i);
    float acc = (a0 + a1) + (a2 + a3);
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) {
        __hip_atomic_store(&part[(size_t)c * S + s], acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        const unsigned t = atomicAdd(&ticket[c], 1u);
        last = (t == (unsigned)S - 1);
    }
    __syncthreads();
    if (last && threadIdx.x == 0) {
        __threadfence();
        float tot = 0.f;
        for (int k = 0; k < S; ++k) tot += __hip_atomic_load(&part[(size_t)c * S + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        db[c] = tot;
        ticket[c] = 0;   // ready for the next call on this stream
    }
}

// ---- AdamW over a flat arena ---------------------------------------------------------------------------------------------
// cube/networks/cubegan.py:275-311 builds three torch.optim.AdamW(betas=(0.8, 0.99)) over ~900 parameter tensors.  Here the
// parameters of a group, their gradients (the gradient-exchange bucket itself, ttscube_amd/distributed.py) and both moment
// estimates are four flat fp32 arenas, so one optimizer step is ONE streaming kernel (16-byte accesses, 5 reads + 3 writes per
// element) instead of a multi-tensor launch chain.  Same update rule and operation order as torch's single-tensor AdamW.
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct AdamArgs {
    float* p;
    const float* g;
    float* m;
    float* v;
    long n;
    float decay;      // 1 - lr * weight_decay
    float om_b1;      // 1 - beta1
    float b2, om_b2;  // beta2, 1 - beta2
    float inv_bc2s;   // 1 / sqrt(1 - beta2^t)
    float eps;
    float step_size;  // lr / (1 - beta1^t)
    const unsigned* guard;   // device word; non-zero = the gradients are not to be trusted: leave everything as it is (nullptr = no guard)
};

__device__ __forceinline__ void adamw_elem(float& p, float g, float& m, float& v, const AdamArgs& a) {
    p *= a.decay;
    m = m + (g - m) * a.om_b1;                 // exp_avg.lerp_(grad, 1 - beta1)
    v = v * a.b2 + (g * g) * a.om_b2;          // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    const float denom = sqrtf(v) * a.inv_bc2s + a.eps;
    p = p - a.step_size * (m / denom);
}

__global__ __launch_bounds__(256) void adamw_flat_kernel(AdamArgs a) {
    if (a.guard && __hip_atomic_load(a.guard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;   // (uniform: every workgroup reads the same word)
    const long n4 = a.n >> 2;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        f32x4 p = reinterpret_cast<f32x4*>(a.p)[i], m = reinterpret_cast<f32x4*>(a.m)[i], v = reinterpret_cast<f32x4*>(a.v)[i];
        const f32x4 g = reinterpret_cast<const f32x4*>(a.g)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float pe = p[e], me = m[e], ve = v[e];
            adamw_elem(pe, g[e], me, ve, a);
            p[e] = pe, m[e] = me, v[e] = ve;
        }
        reinterpret_cast<f32x4*>(a.p)[i] = p;
        reinterpret_cast<f32x4*>(a.m)[i] = m;
        reinterpret_cast<f32x4*>(a.v)[i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {   // tail
        const long i = (n4 << 2) + threadIdx.x;
        adamw_elem(a.p[i], a.g[i], a.m[i], a.v[i], a);
    }
}

// ---- GAN loss terms over lists of tensors ----------------------------------------------------------------------------------
// hifigan.models.{feature_loss, generator_loss, discriminator_loss} [EXTERNAL; call sites cube/networks/cubegan.py:144-149,
// 160-167] walk Python lists of ~50 discriminator outputs / feature maps: a mean, a subtraction, an abs or a square and an add
// per tensor, forward and backward.  One launch evaluates a whole list AND its gradient:
//   kind 0  sum_k w_k * mean|a_k - b_k|          (feature matching, w = 2)       d/db = w sign(b - a) / n_k,  d/da = -d/db
//   kind 1  sum_k w_k * mean (t - a_k)^2         (least squares against target t) d/da = 2 w (a - t) / n_k
// The segment table {a, b, ga, gb, n, w} travels BY VALUE in the kernel arguments (<= 64 segments = 3 KiB: no host
// synchronisation, no staging copy).  Partial sums per workgroup are added in a fixed order by the last workgroup (ticket), so
// the value is deterministic.
struct LossSeg {
    const float* a;
    const float* b;
    float* ga;
    float* gb;
    long n;
    float w;
    float target;
    float slope;   // kind 0: both operands pass through leaky_relu(., slope) first (1 = as they are)
};

constexpr int GAN_LOSS_MAX_SEG = 64;
struct LossTable {
    LossSeg seg[GAN_LOSS_MAX_SEG];
};

__global__ __launch_bounds__(256) void gan_loss_kernel(const LossTable tab, int nseg, int kind, int blocks_per_seg,
                                                       float* __restrict__ partial, unsigned* __restrict__ ticket, float* __restrict__ out) {
    __shared__ float red[4];
    __shared__ bool last;
    const int si = blockIdx.x / blocks_per_seg, bi = blockIdx.x - si * blocks_per_seg;
    const LossSeg sg = tab.seg[si];
    const float scale = sg.w / (float)sg.n;
    float acc = 0.f;
    for (long i = (long)bi * 256 + threadIdx.x; i < sg.n; i += (long)blocks_per_seg * 256) {
        const float a = sg.a[i];
        if (kind == 0) {
            const float b = sg.b[i];
            const float sa = a > 0.f ? 1.f : sg.slope, sb = b > 0.f ? 1.f : sg.slope;   // (slope 1: x * 1 is exact, the plain difference)
            const float d = b * sb - a * sa;
            acc += fabsf(d);
            const float gsign = d > 0.f ? scale : (d < 0.f ? -scale : 0.f);
            if (sg.gb) sg.gb[i] = gsign * sb;
            if (sg.ga) sg.ga[i] = -gsign * sa;
        } else {
            const float d = a - sg.target;
            acc += d * d;
            if (sg.ga) sg.ga[i] = 2.f * scale * d;
        }
    }
    const float s = block_sum(acc, red);
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = s * scale;
        __threadfence();
        last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (last) {   // fixed-order final sum by one workgroup
        float t = 0.f;
        for (int i = threadIdx.x; i < (int)gridDim.x; i += 256) t += __hip_atomic_load(partial + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float tot = block_sum(t, red);
        if (threadIdx.x == 0) {
            *out = tot;
            *ticket = 0u;   // re-armed for the next call
        }
    }
}

// ---- embedding rows (text stacks of the mel decoder, cube/networks/modules.py:869-872) -----------------------------------------
// forward: out[i, :] = table[idx[i], :];  backward: gtable[v, :] = sum_{i: idx[i] == v} gout[i, :], one workgroup per table row walking
// the index list in order (deterministic; the tables have a few dozen rows and a batch a few hundred indices).
__global__ __launch_bounds__(256) void rows_gather_kernel(const float* __restrict__ table, const int* __restrict__ idx, float* __restrict__ out,
                                                          long n, int C, int V) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n * C; e += (long)gridDim.x * blockDim.x) {
        const long i = e / C;
        const int c = (int)(e - i * C);
        const int v = idx[i];
        out[e] = (v >= 0 && v < V) ? table[(size_t)v * C + c] : 0.f;
    }
}

__global__ __launch_bounds__(256) void rows_scatter_add_kernel(const float* __restrict__ gout, const int* __restrict__ idx, float* __restrict__ gtable,
                                                               long n, int C, int skip_row) {
    const int v = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float acc = 0.f;
        if (v != skip_row)
            for (long i = 0; i < n; ++i)
                if (idx[i] == v) acc += gout[(size_t)i * C + c];
        gtable[(size_t)v * C + c] = acc;
    }
}


// ---- polyphase de-interleave of a strided convolution's operands (hifigan/disc_hip.py::HipStridedConv) ------------------------------------
// A Conv1d with stride s runs as a stride-1 convolution over  xr[n, (g, r, ci), m P + w] = x[n, (g, ci), ((m s + r) - pad) P + w]  (zero outside the
// sequence; P = 1, or MPD's period: rows of P samples) with the taps  wp[co, (r, ci), j] = w[co, ci, s j + r]  (zero beyond K).  torch builds both
// with pad + view + permute + reshape: two copies and a fill forward, as many again in the backward pass, per operand — ~800 of the ~1500 element-wise
// launches of a Cubegan step.  Each direction is ONE gather here (the maps are one-to-one onto the unpadded tensors, so the backward is a gather too).
struct DeintArgs {
    const float* src;
    float* dst;
    int N, C, G, s, P, pad, M;   // x [N, C, L P]  <->  xr [N, s C, M P]
    long LP;                     // L * P
};
// One grid column (blockIdx.x) per INPUT row (n, c) of x, 2 048 consecutive positions of it per workgroup, lanes on consecutive positions: x is read
// (forward) / written (backward) in whole lines, the s phase rows of xr each get / give the contiguous run that belongs to those positions; index
// arithmetic in 32 bits with the stride a compile-time constant (the first version walked a flat 64-bit element index with six divisions per element and
// was bound by them: 164 us for the 16 M elements of MPD's second layer against ~30 us of memory time).
//   forward walks the VIRTUAL positions v in [0, M s P) of the zero-padded row: row = v / P, w = v % P, (m, r) = divmod(row, s), x position (row - pad) P + w —
//   a bijection onto xr's s rows, padding included; backward walks the real positions t in [0, L P) and writes zero where the window never reaches (m >= M).
template <int S>
__global__ __launch_bounds__(256) void deinterleave_x_kernel(DeintArgs a, int backward) {
    const int s = S ? S : a.s;
    const int Cg = a.C / a.G;
    const int y = blockIdx.x;
    const int n = y / a.C, c = y - n * a.C;
    const int g = c / Cg, ci = c - g * Cg;
    const unsigned MP = (unsigned)a.M * (unsigned)a.P;
    const unsigned P = (unsigned)a.P;
    const size_t xoff = ((size_t)n * a.C + c) * (size_t)a.LP;                                           // row (n, c) of x
    const size_t roff = ((size_t)n * s * a.C + (size_t)g * s * Cg + ci) * (size_t)MP;                   // row (n, g, r = 0, ci) of xr
    const size_t rstep = (size_t)Cg * MP;                                                                // r -> r + 1
    const unsigned base = blockIdx.y * 2048u + threadIdx.x;
    if (!backward) {
        const unsigned span = MP * (unsigned)s;
        const float* __restrict__ x = a.src + xoff;
        float* __restrict__ xr = a.dst + roff;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned v = base + 256u * j;
            if (v >= span) break;
            const unsigned row = P == 1u ? v : v / P;
            const unsigned w = v - row * P;
            const unsigned m = row / (unsigned)s, r = row - m * (unsigned)s;
            const long xpos = ((long)row - a.pad) * (long)P + w;
            xr[r * rstep + m * P + w] = (row >= (unsigned)a.pad && xpos < a.LP) ? x[xpos] : 0.f;
        }
    } else {
        const float* __restrict__ xr = a.src + roff;
        float* __restrict__ x = a.dst + xoff;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned t = base + 256u * j;
            if ((long)t >= a.LP) break;
            const unsigned row0 = P == 1u ? t : t / P;
            const unsigned w = t - row0 * P;
            const unsigned row = row0 + (unsigned)a.pad;
            const unsigned m = row / (unsigned)s, r = row - m * (unsigned)s;
            x[t] = m < (unsigned)a.M ? xr[r * rstep + m * P + w] : 0.f;
        }
    }
}
// w [Cout, Cg, K]  <->  wp [Cout, s Cg, J]
__global__ __launch_bounds__(256) void deinterleave_w_kernel(const float* __restrict__ src, float* __restrict__ dst, int Cout, int Cg, int K, int s, int J,
                                                            int backward) {
    if (!backward) {
        const long total = (long)Cout * s * Cg * J;
        for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
            const int j = (int)(e % J);
            long t = e / J;
            const int ci = (int)(t % Cg);
            t /= Cg;
            const int r = (int)(t % s);
            const int co = (int)(t / s);
            const int k = s * j + r;
            dst[e] = k < K ? src[((size_t)co * Cg + ci) * K + k] : 0.f;
        }
    } else {
        const long total = (long)Cout * Cg * K;
        for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
            const int k = (int)(e % K);
            long t = e / K;
            const int ci = (int)(t % Cg);
            const int co = (int)(t / Cg);
            dst[e] = src[(((size_t)co * s + (k % s)) * Cg + ci) * J + k / s];
        }
    }
}

// ---- spectral normalisation of a weight (torch.nn.utils.spectral_norm: the first scale discriminator, [EXTERNAL hifigan/models.py]
//      DiscriminatorS(use_spectral_norm=True), instantiated by cube/networks/cubegan.py:40-41) — the small pieces around the two mat-vecs of the power iteration (those run on the MFMA GEMM):
// out[r] = sum_c W[r, c] x[c]: one workgroup per row (per-thread strided chains, fixed LDS tree)
__global__ __launch_bounds__(256) void matvec_rows_kernel(const float* __restrict__ W, const float* __restrict__ x, float* __restrict__ out, long Cc) {
    __shared__ float red[256];
    const float* w = W + (size_t)blockIdx.x * Cc;
    float s = 0.f;
    for (long c = threadIdx.x; c < Cc; c += 256) s = fmaf(w[c], x[c], s);
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}

// ws[part][c] = sum_{r in part} W[r, c] u[r] (rows ascending), then out[c] = sum_part ws[part][c] (parts ascending): W^T u in a fixed order
__global__ __launch_bounds__(256) void matvec_cols_partial_kernel(const float* __restrict__ W, const float* __restrict__ u, float* __restrict__ ws, int R, long Cc,
                                                                  int rows_per_part) {
    const long c = (long)blockIdx.x * 256 + threadIdx.x;
    if (c >= Cc) return;
    const int r0 = blockIdx.y * rows_per_part, r1 = r0 + rows_per_part < R ? r0 + rows_per_part : R;
    float s = 0.f;
    for (int r = r0; r < r1; ++r) s = fmaf(W[(size_t)r * Cc + c], u[r], s);
    ws[(size_t)blockIdx.y * Cc + c] = s;
}

__global__ __launch_bounds__(256) void matvec_cols_final_kernel(const float* __restrict__ ws, float* __restrict__ out, int parts, long Cc) {
    const long c = (long)blockIdx.x * 256 + threadIdx.x;
    if (c >= Cc) return;
    float s = 0.f;
    for (int p = 0; p < parts; ++p) s += ws[(size_t)p * Cc + c];
    out[c] = s;
}

// gtable[v, :] = sum of gout[i, :] over the CONTIGUOUS run of i with idx[i] == v (idx non-decreasing), i ascending: the adjoint of a row gather
// whose index list is sorted (phoneme rows -> frame rows) in O(n C) — rows_scatter_add_kernel walks the whole index list per table row
__global__ __launch_bounds__(256) void rows_segment_sum_kernel(const float* __restrict__ gout, const int* __restrict__ idx, float* __restrict__ gtable, long n,
                                                               int C) {
    const int v = blockIdx.x;
    long lo = 0, hi = n;                    // first i with idx[i] >= v
    while (lo < hi) {
        const long mid = (lo + hi) >> 1;
        if (idx[mid] < v) lo = mid + 1; else hi = mid;
    }
    const long beg = lo;
    hi = n;                                 // first i with idx[i] > v
    while (lo < hi) {
        const long mid = (lo + hi) >> 1;
        if (idx[mid] <= v) lo = mid + 1; else hi = mid;
    }
    const long end = lo;
    for (int c = threadIdx.x; c < C; c += 256) {
        float acc = 0.f;
        for (long i = beg; i < end; ++i) acc += gout[(size_t)i * C + c];
        gtable[(size_t)v * C + c] = acc;
    }
}

// x[n] -> out = x / max(||x||, eps), norm_out[0] = ||x||: one workgroup, squares summed per thread over a fixed stride, fixed LDS tree
__global__ __launch_bounds__(256) void l2_normalize_kernel(const float* __restrict__ x, int n, float eps, float* __restrict__ out, float* __restrict__ norm_out) {
    __shared__ float red[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s = fmaf(x[i], x[i], s);
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    const float nrm = sqrtf(red[0]);
    if (threadIdx.x == 0 && norm_out) norm_out[0] = nrm;
    if (out) {
        const float d = fmaxf(nrm, eps);
        for (int i = threadIdx.x; i < n; i += 256) out[i] = x[i] / d;
    }
}

// sum_i a[i] b[i] in a fixed order: `parts` workgroups over contiguous ranges (per-thread strided chains, LDS tree), then one workgroup over the parts
__global__ __launch_bounds__(256) void dot_partial_kernel(const float* __restrict__ a, const float* __restrict__ b, long n, long per, float* __restrict__ ws) {
    __shared__ float red[256];
    const long beg = (long)blockIdx.x * per, end = beg + per < n ? beg + per : n;
    float s = 0.f;
    for (long i = beg + threadIdx.x; i < end; i += 256) s = fmaf(a[i], b[i], s);
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) ws[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void dot_final_kernel(const float* __restrict__ ws, int parts, float* __restrict__ out) {
    __shared__ float red[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < parts; i += 256) s += ws[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}

// out = w / sigma (sigma on the device)
__global__ __launch_bounds__(256) void div_scalar_kernel(const float* __restrict__ w, const float* __restrict__ sigma, float* __restrict__ out, long n) {
    const float s = sigma[0];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = w[i] / s;
}

// gradient of wn = w / sigma, sigma = u^T W v (u, v constants): dW[r, c] = dWn[r, c] / sigma - (sum(dWn . W) / sigma^2) u[r] v[c]
__global__ __launch_bounds__(256) void spectral_bwd_kernel(const float* __restrict__ dwn, const float* __restrict__ u, const float* __restrict__ v,
                                                           const float* __restrict__ sigma, const float* __restrict__ dotp, float* __restrict__ dw, int R,
                                                           long Cc) {
    const float s = sigma[0];
    const float k = dotp[0] / (s * s);
    const long n = (long)R * Cc;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long r = i / Cc, c = i - r * Cc;
        dw[i] = dwn[i] / s - k * (u[r] * v[c]);
    }
}

}  // namespace ttsc

using namespace ttsc;

static int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s launch failed: %s", what, hipGetErrorString(e));
        return TTSC_EHIP;
    }
    return TTSC_OK;
}

extern "C" int ttsc_weight_norm_forward(const float* v_dev, const float* g_dev, float* w_dev, float* norm_dev, int32_t rows, int64_t cols,
                                        void* stream) {
    TTSC_REQUIRE(v_dev && g_dev && w_dev && norm_dev && rows > 0 && cols > 0 && cols < (1ll << 31), "ttsc_weight_norm_forward: bad argument");
    hipLaunchKernelGGL(wn_fwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, v_dev, g_dev, w_dev, norm_dev, (int)cols);
    return check_launch("wn_fwd_kernel");
}

extern "C" int ttsc_weight_norm_backward(const float* dw_dev, const float* v_dev, const float* g_dev, const float* norm_dev, float* dv_dev,
                                         float* dg_dev, int32_t rows, int64_t cols, void* stream) {
    TTSC_REQUIRE(dw_dev && v_dev && g_dev && norm_dev && dv_dev && dg_dev && rows > 0 && cols > 0 && cols < (1ll << 31),
                 "ttsc_weight_norm_backward: bad argument");
    hipLaunchKernelGGL(wn_bwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, dw_dev, v_dev, g_dev, norm_dev, dv_dev, dg_dev, (int)cols);
    return check_launch("wn_bwd_kernel");
}

static int bias_grad_splits(int32_t B, int32_t C, int64_t L) {
    long s = (L + 1023) / 1024;                  // >= 1024 positions of every batch item per block
    const long cap = (1024 + C - 1) / C;         // ~1024 blocks in all
    if (s > cap) s = cap;
    if (s < 1) s = 1;
    return (int)s;
}

static int matvec_parts(int R) { return R >= 64 ? (R / 32 > 32 ? 32 : R / 32) : 1; }

extern "C" size_t ttsc_matvec_workspace_bytes(int32_t rows, int64_t cols) { return rows > 0 && cols > 0 ? (size_t)matvec_parts(rows) * cols * sizeof(float) : 0; }

// W [rows, cols] row-major.  transpose = 0: out[rows] = W x (x [cols]);  transpose = 1: out[cols] = W^T x (x [rows]; ws_dev >=
// ttsc_matvec_workspace_bytes).  One pass over W, fixed summation order.
extern "C" int ttsc_matvec(const float* w_dev, int32_t rows, int64_t cols, const float* x_dev, int32_t transpose, float* out_dev, void* ws_dev,
                           size_t ws_bytes, void* stream) {
    TTSC_REQUIRE(w_dev && x_dev && out_dev && rows > 0 && cols > 0, "ttsc_matvec: bad argument");
    hipStream_t s = (hipStream_t)stream;
    if (!transpose) {
        hipLaunchKernelGGL(matvec_rows_kernel, dim3((unsigned)rows), dim3(256), 0, s, w_dev, x_dev, out_dev, (long)cols);
        return check_launch("matvec_rows_kernel");
    }
    const int parts = matvec_parts(rows);
    TTSC_REQUIRE(ws_dev && ws_bytes >= (size_t)parts * cols * sizeof(float), "ttsc_matvec: workspace too small");
    const int rpp = (rows + parts - 1) / parts;
    const unsigned gx = (unsigned)((cols + 255) / 256);
    hipLaunchKernelGGL(matvec_cols_partial_kernel, dim3(gx, (unsigned)parts), dim3(256), 0, s, w_dev, x_dev, (float*)ws_dev, rows, (long)cols, rpp);
    hipLaunchKernelGGL(matvec_cols_final_kernel, dim3(gx), dim3(256), 0, s, (const float*)ws_dev, out_dev, parts, (long)cols);
    return check_launch("matvec_cols kernels");
}

// the adjoint of ttsc_rows_gather for a NON-DECREASING index list: gtable [V, C] = per-row sums of gout [n, C] (rows without an index get zeros)
extern "C" int ttsc_rows_segment_sum(const float* gout_dev, const int32_t* idx_sorted_dev, float* gtable_dev, int64_t n, int32_t C, int32_t V, void* stream) {
    TTSC_REQUIRE(gout_dev && idx_sorted_dev && gtable_dev && n > 0 && C > 0 && V > 0, "ttsc_rows_segment_sum: bad argument");
    hipLaunchKernelGGL(rows_segment_sum_kernel, dim3((unsigned)V), dim3(256), 0, (hipStream_t)stream, gout_dev, idx_sorted_dev, gtable_dev, (long)n, C);
    return check_launch("rows_segment_sum_kernel");
}

extern "C" int ttsc_l2_normalize(const float* x_dev, int32_t n, float eps, float* out_dev, float* norm_dev, void* stream) {
    TTSC_REQUIRE(x_dev && n > 0 && (out_dev || norm_dev), "ttsc_l2_normalize: bad argument");
    hipLaunchKernelGGL(l2_normalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, x_dev, n, eps, out_dev, norm_dev);
    return check_launch("l2_normalize_kernel");
}

static int dot_parts(int64_t n) { return (int)(n >= (1 << 16) ? 256 : (n >= 4096 ? 16 : 1)); }

extern "C" size_t ttsc_dot_workspace_bytes(int64_t n) { return n > 0 ? (size_t)dot_parts(n) * sizeof(float) : 0; }

// out_dev[0] = sum_i a[i] b[i], summed in a fixed order (deterministic)
extern "C" int ttsc_dot(const float* a_dev, const float* b_dev, int64_t n, float* out_dev, void* ws_dev, size_t ws_bytes, void* stream) {
    TTSC_REQUIRE(a_dev && b_dev && out_dev && ws_dev && n > 0, "ttsc_dot: bad argument");
    const int parts = dot_parts(n);
    TTSC_REQUIRE(ws_bytes >= (size_t)parts * sizeof(float), "ttsc_dot: workspace too small");
    const long per = (long)((n + parts - 1) / parts);
    hipLaunchKernelGGL(dot_partial_kernel, dim3((unsigned)parts), dim3(256), 0, (hipStream_t)stream, a_dev, b_dev, (long)n, per, (float*)ws_dev);
    hipLaunchKernelGGL(dot_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)ws_dev, parts, out_dev);
    return check_launch("dot kernels");
}

extern "C" int ttsc_div_scalar(const float* w_dev, const float* sigma_dev, float* out_dev, int64_t n, void* stream) {
    TTSC_REQUIRE(w_dev && sigma_dev && out_dev && n > 0, "ttsc_div_scalar: bad argument");
    hipLaunchKernelGGL(div_scalar_kernel, dim3((unsigned)std::min<long>((n + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream, w_dev, sigma_dev, out_dev,
                       (long)n);
    return check_launch("div_scalar_kernel");
}

extern "C" int ttsc_spectral_norm_backward(const float* dwn_dev, const float* u_dev, const float* v_dev, const float* sigma_dev, const float* dot_dev,
                                           float* dw_dev, int32_t rows, int64_t cols, void* stream) {
    TTSC_REQUIRE(dwn_dev && u_dev && v_dev && sigma_dev && dot_dev && dw_dev && rows > 0 && cols > 0, "ttsc_spectral_norm_backward: bad argument");
    const long n = (long)rows * cols;
    hipLaunchKernelGGL(spectral_bwd_kernel, dim3((unsigned)std::min<long>((n + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream, dwn_dev, u_dev, v_dev,
                       sigma_dev, dot_dev, dw_dev, rows, (long)cols);
    return check_launch("spectral_bwd_kernel");
}

extern "C" size_t ttsc_bias_grad_workspace_bytes(int32_t B, int32_t C, int64_t L) {
    if (B <= 0 || C <= 0 || L <= 0) return 0;
    return (size_t)C * bias_grad_splits(B, C, L) * sizeof(float) + (size_t)C * sizeof(unsigned);
}

extern "C" int ttsc_bias_grad(const float* dy_dev, float* db_dev, int32_t B, int32_t C, int64_t L, void* ws_dev, size_t ws_bytes,
                              int32_t ws_is_fresh, void* stream) {
    TTSC_REQUIRE(dy_dev && db_dev && ws_dev && B > 0 && C > 0 && L > 0 && L < (1ll << 31), "ttsc_bias_grad: bad argument");
    TTSC_REQUIRE(ws_bytes >= ttsc_bias_grad_workspace_bytes(B, C, L), "ttsc_bias_grad: workspace too small");
    const int S = bias_grad_splits(B, C, L);
    float* part = (float*)ws_dev;
    unsigned* ticket = (unsigned*)(part + (size_t)C * S);
    hipStream_t s = (hipStream_t)stream;
    if (ws_is_fresh) TTSC_HIP_CHECK(hipMemsetAsync(ticket, 0, (size_t)C * sizeof(unsigned), s));   // tickets must start at zero
    hipLaunchKernelGGL(bias_grad_kernel, dim3(C, S), dim3(256), 0, s, dy_dev, db_dev, part, ticket, B, C, (int)L, S);
    return check_launch("bias_grad_kernel");
}

extern "C" int ttsc_adamw_step(float* p_dev, const float* g_dev, float* m_dev, float* v_dev, int64_t n, float lr, float beta1, float beta2,
                               float eps, float weight_decay, int64_t step, void* stream) {
    return ttsc_adamw_step_guarded(p_dev, g_dev, m_dev, v_dev, n, lr, beta1, beta2, eps, weight_decay, step, nullptr, stream);
}

extern "C" int ttsc_adamw_step_guarded(float* p_dev, const float* g_dev, float* m_dev, float* v_dev, int64_t n, float lr, float beta1, float beta2,
                                       float eps, float weight_decay, int64_t step, const uint32_t* guard_dev, void* stream) {
    TTSC_REQUIRE(p_dev && g_dev && m_dev && v_dev && n > 0 && step >= 1, "ttsc_adamw_step: bad argument");
    TTSC_REQUIRE((((uintptr_t)p_dev | (uintptr_t)g_dev | (uintptr_t)m_dev | (uintptr_t)v_dev) & 15) == 0, "ttsc_adamw_step: arenas must be 16-byte aligned");
    AdamArgs a;
    a.p = p_dev;
    a.g = g_dev;
    a.m = m_dev;
    a.v = v_dev;
    a.n = n;
    a.decay = 1.f - lr * weight_decay;
    a.om_b1 = 1.f - beta1;
    a.b2 = beta2;
    a.om_b2 = 1.f - beta2;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    a.inv_bc2s = (float)(1.0 / sqrt(bc2));
    a.eps = eps;
    a.step_size = (float)((double)lr / bc1);
    a.guard = guard_dev;
    const long n4 = n >> 2;
    const unsigned blocks = (unsigned)std::min<long>(std::max<long>((n4 + 255) / 256, 1), 2048);
    hipLaunchKernelGGL(adamw_flat_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("adamw_flat_kernel");
}

extern "C" size_t ttsc_gan_loss_workspace_bytes(int32_t nseg) {
    if (nseg <= 0) return 0;
    return 64 + (size_t)nseg * 64 * sizeof(float);   // ticket + partial sums
}

extern "C" int ttsc_gan_loss(int32_t kind, int32_t nseg, const void* const* a_dev, const void* const* b_dev, void* const* ga_dev,
                             void* const* gb_dev, const int64_t* numel, const float* weight, float target, float* out_dev, void* ws_dev,
                             size_t ws_bytes, void* stream) {
    return ttsc_gan_loss_lrelu(kind, nseg, a_dev, b_dev, ga_dev, gb_dev, numel, weight, nullptr, target, out_dev, ws_dev, ws_bytes, stream);
}

extern "C" int ttsc_gan_loss_lrelu(int32_t kind, int32_t nseg, const void* const* a_dev, const void* const* b_dev, void* const* ga_dev,
                                   void* const* gb_dev, const int64_t* numel, const float* weight, const float* slope, float target, float* out_dev,
                                   void* ws_dev, size_t ws_bytes, void* stream) {
    TTSC_REQUIRE(kind == 0 || kind == 1, "ttsc_gan_loss: kind must be 0 (L1 between pairs) or 1 (squared distance to a target)");
    TTSC_REQUIRE(nseg > 0 && nseg <= GAN_LOSS_MAX_SEG && a_dev && numel && weight && out_dev && ws_dev, "ttsc_gan_loss: bad argument (1..%d tensors per call)", GAN_LOSS_MAX_SEG);
    TTSC_REQUIRE(kind == 1 || b_dev, "ttsc_gan_loss: kind 0 needs the second operand list");
    TTSC_REQUIRE(ws_bytes >= ttsc_gan_loss_workspace_bytes(nseg), "ttsc_gan_loss: workspace too small");
    LossTable tab;
    memset(&tab, 0, sizeof(tab));
    long nmax = 0;
    for (int i = 0; i < nseg; ++i) {
        TTSC_REQUIRE(a_dev[i] && numel[i] > 0, "ttsc_gan_loss: empty segment %d", i);
        tab.seg[i] = LossSeg{(const float*)a_dev[i], kind == 0 ? (const float*)b_dev[i] : nullptr, ga_dev ? (float*)ga_dev[i] : nullptr,
                             (kind == 0 && gb_dev) ? (float*)gb_dev[i] : nullptr, (long)numel[i], weight[i], target, slope ? slope[i] : 1.f};
        TTSC_REQUIRE(tab.seg[i].slope >= 0.f && tab.seg[i].slope <= 1.f && (kind == 0 || tab.seg[i].slope == 1.f), "ttsc_gan_loss: slope of segment %d (in [0, 1]; kind 0 only)", i);
        TTSC_REQUIRE(kind == 1 || tab.seg[i].b, "ttsc_gan_loss: null operand in segment %d", i);
        nmax = std::max<long>(nmax, numel[i]);
    }
    const int bps = (int)std::min<long>(std::max<long>(nmax / 8192, 1), 64);   // workgroups per segment
    hipStream_t s = (hipStream_t)stream;
    unsigned* ticket = (unsigned*)ws_dev;                    // (re-armed by the kernel; zeroed here so that a fresh workspace works too)
    float* partial = (float*)((char*)ws_dev + 64);
    TTSC_HIP_CHECK(hipMemsetAsync(ticket, 0, 64, s));
    hipLaunchKernelGGL(gan_loss_kernel, dim3((unsigned)(nseg * bps)), dim3(256), 0, s, tab, nseg, kind, bps, partial, ticket, out_dev);
    return check_launch("gan_loss_kernel");
}

extern "C" int ttsc_rows_gather(const float* table_dev, const int32_t* idx_dev, float* out_dev, int64_t n, int32_t C, int32_t V, void* stream) {
    TTSC_REQUIRE(table_dev && idx_dev && out_dev && n > 0 && C > 0 && V > 0, "ttsc_rows_gather: bad argument");
    const unsigned blocks = (unsigned)std::min<long>((n * C + 255) / 256, 2048);
    hipLaunchKernelGGL(rows_gather_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, table_dev, idx_dev, out_dev, (long)n, C, V);
    return check_launch("rows_gather_kernel");
}

extern "C" int ttsc_rows_scatter_add(const float* gout_dev, const int32_t* idx_dev, float* gtable_dev, int64_t n, int32_t C, int32_t V,
                                     int32_t skip_row, void* stream) {
    TTSC_REQUIRE(gout_dev && idx_dev && gtable_dev && n > 0 && C > 0 && V > 0, "ttsc_rows_scatter_add: bad argument");
    hipLaunchKernelGGL(rows_scatter_add_kernel, dim3((unsigned)V), dim3(256), 0, (hipStream_t)stream, gout_dev, idx_dev, gtable_dev, (long)n, C, skip_row);
    return check_launch("rows_scatter_add_kernel");
}


extern "C" int ttsc_deinterleave_x(const float* src_dev, float* dst_dev, int32_t N, int32_t C, int64_t L, int32_t groups, int32_t stride, int32_t period,
                                   int32_t pad, int32_t M, int32_t backward, void* stream) {
    TTSC_REQUIRE(src_dev && dst_dev && N > 0 && C > 0 && L > 0 && groups > 0 && C % groups == 0 && stride > 0 && period > 0 && pad >= 0 && M > 0,
                 "ttsc_deinterleave_x: bad argument");
    DeintArgs a{src_dev, dst_dev, N, C, groups, stride, period, pad, M, (long)L * period};
    const long span = backward ? a.LP : (long)M * period * stride;       // positions walked per row of x
    TTSC_REQUIRE(a.LP < (1L << 30) && (long)M * period * stride < (1L << 30) && (long)N * C < (1L << 31), "ttsc_deinterleave_x: sequence too long for 32-bit positions");
    const dim3 grid((unsigned)((long)N * C), (unsigned)((span + 2047) / 2048));
    hipStream_t st = (hipStream_t)stream;
    switch (stride) {
        case 2: hipLaunchKernelGGL(deinterleave_x_kernel<2>, grid, dim3(256), 0, st, a, backward); break;
        case 3: hipLaunchKernelGGL(deinterleave_x_kernel<3>, grid, dim3(256), 0, st, a, backward); break;
        case 4: hipLaunchKernelGGL(deinterleave_x_kernel<4>, grid, dim3(256), 0, st, a, backward); break;
        default: hipLaunchKernelGGL(deinterleave_x_kernel<0>, grid, dim3(256), 0, st, a, backward); break;
    }
    return check_launch("deinterleave_x_kernel");
}

extern "C" int ttsc_deinterleave_w(const float* src_dev, float* dst_dev, int32_t Cout, int32_t Cg, int32_t K, int32_t stride, int32_t backward, void* stream) {
    TTSC_REQUIRE(src_dev && dst_dev && Cout > 0 && Cg > 0 && K > 0 && stride > 0, "ttsc_deinterleave_w: bad argument");
    const int J = (K + stride - 1) / stride;
    const long total = backward ? (long)Cout * Cg * K : (long)Cout * stride * Cg * J;
    hipLaunchKernelGGL(deinterleave_w_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream, src_dev, dst_dev,
                       Cout, Cg, K, stride, J, backward);
    return check_launch("deinterleave_w_kernel");
}
