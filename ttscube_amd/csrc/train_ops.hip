// Small training-side kernels of the generator — gfx950.
//
// The reference wraps every generator convolution in torch.nn.utils.weight_norm [EXTERNAL hifigan/models.py] and lets
// torch autograd differentiate  w = g * v / ||v||  and the bias add as a dozen elementwise / reduction launches per layer.
// At the crop sizes of `Cubegan.training_step` (cube/networks/cubegan.py:116-134: 50 frames) those ~5 us launches add up
// to a third of the step, so each becomes one HBM-bound kernel here:
//   ttsc_weight_norm_forward    w[r,:] = v[r,:] * (g[r] / ||v[r,:]||),  norm[r] = ||v[r,:]||          (one workgroup per row)
//   ttsc_weight_norm_backward   dg[r] = <dw,v> / n,  dv = (g/n) * dw - v * (g * <dw,v> / n^3)
//   ttsc_bias_grad              db[c] = sum_{b,t} dy[b,c,t]   (fixed-order two-level sum, last workgroup of a channel finishes)
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
