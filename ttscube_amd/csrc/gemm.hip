// Y[M,N] = act(X[M,K] . W[N,K]^T + bias[N])  — fp32 MFMA (v_mfma_f32_32x32x2_f32) "NT" GEMM for gfx950.
//
// This is torch.nn.Linear's contract (weight [out,in], row-major activations).  On the hot path it carries the
// HOISTED input projections of every LSTM/BiLSTM (all time steps of a sequence in one GEMM instead of one mat-vec
// per step: cube/networks/modules.py:873-905, textcoder.py:55-92) and the Linear heads (_dur_output,
// _pitch_output, _cond_output, _mel_output, PreNet).
//
// 128x128x16 workgroup tile, 4 waves as 2x2, each wave 2x2 MFMA tiles of 32x32 (4 independent accumulators keep the
// fp32 matrix pipe issuing back to back).  Both operands are K-contiguous, so they are staged the same way:
// 16-byte global loads -> LDS rows padded to 17 floats (bank-conflict-free column reads for the MFMA fragments).
#include "common.hpp"

namespace ttsc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GM = 128, GN = 128, GK = 16, GLD = GK + 1;

struct GemmArgs {
    const float* X;  // [M, ldx]
    const float* W;  // [N, K]
    const float* bias;
    float* Y;        // [M, ldy]
    int M, N, K, ldx, ldy;
    int act, accumulate;
};

__device__ __forceinline__ float gemm_act(float v, int act) {
    if (act == TTSC_ACT_TANH) return tanhf(v);
    if (act == TTSC_ACT_RELU) return fmaxf(v, 0.f);
    if (act == TTSC_ACT_SIGMOID) return 1.f / (1.f + expf(-v));
    return v;
}

template <bool VEC>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmArgs a) {
    __shared__ float As[GM * GLD];
    __shared__ float Bs[GN * GLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * GM, n0 = blockIdx.x * GN;
    const int half = lane >> 5, l31 = lane & 31;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int k0 = 0; k0 < a.K; k0 += GK) {
        __syncthreads();
        if (VEC) {
            // 128 rows x 4 float4 per operand = 512 float4 -> 2 per thread per operand
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int idx = tid + it * 256;
                const int row = idx >> 2, k4 = (idx & 3) * 4;
                float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
                if (m0 + row < a.M && k0 + k4 < a.K) va = *reinterpret_cast<const float4*>(a.X + (size_t)(m0 + row) * a.ldx + k0 + k4);
                if (n0 + row < a.N && k0 + k4 < a.K) vb = *reinterpret_cast<const float4*>(a.W + (size_t)(n0 + row) * a.K + k0 + k4);
                float* da = As + row * GLD + k4;
                float* db = Bs + row * GLD + k4;
                da[0] = va.x; da[1] = va.y; da[2] = va.z; da[3] = va.w;
                db[0] = vb.x; db[1] = vb.y; db[2] = vb.z; db[3] = vb.w;
            }
        } else {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int idx = tid + it * 256;
                const int row = idx >> 4, k = idx & 15;
                float va = 0.f, vb = 0.f;
                if (m0 + row < a.M && k0 + k < a.K) va = a.X[(size_t)(m0 + row) * a.ldx + k0 + k];
                if (n0 + row < a.N && k0 + k < a.K) vb = a.W[(size_t)(n0 + row) * a.K + k0 + k];
                As[row * GLD + k] = va;
                Bs[row * GLD + k] = vb;
            }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < GK / 2; ++kk) {
            float af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = As[(wm * 64 + i * 32 + l31) * GLD + kk * 2 + half];
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = Bs[(wn * 64 + j * 32 + l31) * GLD + kk * 2 + half];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    }
    // C/D layout: col (n) = lane & 31, row (m) = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + l31;
            if (n >= a.N) continue;
            const float bv = a.bias ? a.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (m < a.M) {
                    float* y = a.Y + (size_t)m * a.ldy + n;
                    float v = acc[i][j][r] + bv;
                    if (a.accumulate) v += *y;
                    *y = gemm_act(v, a.act);
                }
            }
        }
}

}  // namespace ttsc

using namespace ttsc;

// x_dev [M, ldx] (first K columns used), w_dev [N, K], bias_dev [N] or NULL -> y_dev [M, ldy] (first N columns written).
// ldx is only a row stride: 0 < ldx < K reads OVERLAPPING rows (row m = x_dev[m*ldx .. m*ldx+K)), which is how the
// STFT frames a signal without materialising the frames (io_utils/melspec.py: ldx = hop, K = n_fft).
extern "C" int ttsc_linear_forward(const float* x_dev, const float* w_dev, const float* bias_dev, float* y_dev, int64_t M,
                                   int32_t N, int32_t K, int64_t ldx, int64_t ldy, int32_t act, int32_t accumulate, void* stream) {
    TTSC_REQUIRE(x_dev && w_dev && y_dev, "ttsc_linear_forward: null argument");
    TTSC_REQUIRE(M > 0 && N > 0 && K > 0 && ldx > 0 && ldy >= N, "ttsc_linear_forward: bad shape M=%lld N=%d K=%d ldx=%lld ldy=%lld",
                 (long long)M, N, K, (long long)ldx, (long long)ldy);
    TTSC_REQUIRE(M < (1ll << 31) && ldx < (1ll << 31) && ldy < (1ll << 31), "ttsc_linear_forward: dimension too large");
    GemmArgs a{x_dev, w_dev, bias_dev, y_dev, (int)M, N, K, (int)ldx, (int)ldy, act, accumulate};
    dim3 grid((unsigned)ceil_div(N, GN), (unsigned)ceil_div(M, GM));
    const bool vec = (K % 4 == 0) && (ldx % 4 == 0) && (((uintptr_t)x_dev & 15) == 0) && (((uintptr_t)w_dev & 15) == 0);
    if (vec)
        hipLaunchKernelGGL(gemm_nt_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(gemm_nt_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("gemm_nt_kernel launch failed: %s", hipGetErrorString(e));
        return TTSC_EHIP;
    }
    return TTSC_OK;
}
