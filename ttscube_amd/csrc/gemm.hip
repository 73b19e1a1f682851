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
#include "conv_internal.hpp"
#include <map>
#include <mutex>

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


// ---------------------------------------------------------------------------------------------------------------------------
// The same NT contract on the f16 matrix pipe with fp32-class accuracy: every fp32 operand value v is split on its way into LDS into
// hi = rn16(v), lo = rn16(v - hi) and the product is summed as lo_x.hi_w + hi_x.lo_w + hi_x.hi_w in fp32 accumulators (three
// v_mfma_f32_32x32x16_f16 per tile and 16 k: the ceiling is a third of the dense f16 peak = 833 TFLOP/s against 157 for the fp32 MFMA above).
// The dropped lo.lo term and the two roundings leave ~2^-22 relative per product — the split the vocoder convolutions use (conv1d.hip), here
// without pre-scales: operands must lie inside the fp16 range (|v| <= 65504; anything beyond sets the launch's status word, reported by
// ttsc_gemm_split_status) and values below 2^-14 keep only an absolute 2^-25.  Callers: the hoisted input projections of the text-side
// BiLSTMs (activations of tanh / LSTM layers, embeddings, mel frames).  `lengths` / `period`: rows are [utterance][period] and rows at or
// beyond an utterance's length are padding nobody reads — a 128-row tile that lies wholly in padding returns at once (its outputs stay
// unwritten).  128 x 128 x 32 tile, 4 waves as 2 x 2, each 2 x 2 MFMA tiles; the next k-slice's global loads are in flight during the MFMAs.
typedef _Float16 gemm_half8 __attribute__((ext_vector_type(8)));
constexpr int SK = 32, SLD = SK + 8;   // k-slice, LDS row pitch in halfs (80 bytes: 16-byte aligned fragments)

struct GemmSplitArgs {
    const float* X;
    const float* W;
    const float* bias;
    float* Y;
    int M, N, K, ldx, ldy;
    int act, accumulate;
    const int* lengths;
    int period;
    unsigned* status;
};

__global__ __launch_bounds__(256) void gemm_nt_f16x3_kernel(GemmSplitArgs a) {
    __shared__ __attribute__((aligned(16))) _Float16 Ah[GM * SLD];
    __shared__ __attribute__((aligned(16))) _Float16 Al[GM * SLD];
    __shared__ __attribute__((aligned(16))) _Float16 Bh[GN * SLD];
    __shared__ __attribute__((aligned(16))) _Float16 Bl[GN * SLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * GM, n0 = blockIdx.x * GN;
    const int half = lane >> 5, l31 = lane & 31;
    if (a.lengths) {
        int live = 0;
        if (tid < GM && m0 + tid < a.M) {
            const int b = (m0 + tid) / a.period;
            live = (m0 + tid) - b * a.period < a.lengths[b];
        }
        if (!__syncthreads_or(live)) return;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float4 ra[4], rb[4];
    unsigned bad = 0;
    // 128 rows x 8 float4 per operand and k-slice -> 4 per thread and operand; 8 consecutive lanes read one row's 128 bytes
    auto gload = [&](int k0) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = tid + it * 256;
            const int row = idx >> 3, k4 = (idx & 7) * 4;
            ra[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            rb[it] = ra[it];
            if (m0 + row < a.M && k0 + k4 < a.K) ra[it] = *reinterpret_cast<const float4*>(a.X + (size_t)(m0 + row) * a.ldx + k0 + k4);
            if (n0 + row < a.N && k0 + k4 < a.K) rb[it] = *reinterpret_cast<const float4*>(a.W + (size_t)(n0 + row) * a.K + k0 + k4);
        }
    };
    auto put = [&](const float4& v, _Float16* hi_plane, _Float16* lo_plane, int off) {
        unsigned h0, l0, h1, l1;
        split2_f16(v.x, v.y, h0, l0);
        split2_f16(v.z, v.w, h1, l1);
        // a hi half with an all-ones exponent: the value was beyond the fp16 range (or not finite)
        bad |= (unsigned)(((h0 & 0x7c00u) == 0x7c00u) | ((h0 & 0x7c000000u) == 0x7c000000u) | ((h1 & 0x7c00u) == 0x7c00u) | ((h1 & 0x7c000000u) == 0x7c000000u));
        *reinterpret_cast<uint2*>(hi_plane + off) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(lo_plane + off) = make_uint2(l0, l1);
    };
    gload(0);
    for (int k0 = 0; k0 < a.K; k0 += SK) {
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = tid + it * 256;
            const int off = (idx >> 3) * SLD + (idx & 7) * 4;
            put(ra[it], Ah, Al, off);
            put(rb[it], Bh, Bl, off);
        }
        __syncthreads();
        if (k0 + SK < a.K) gload(k0 + SK);
#pragma unroll
        for (int ks = 0; ks < SK / 16; ++ks) {
            gemm_half8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int off = (wm * 64 + i * 32 + l31) * SLD + ks * 16 + half * 8;
                ah[i] = *reinterpret_cast<const gemm_half8*>(Ah + off);
                al[i] = *reinterpret_cast<const gemm_half8*>(Al + off);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int off = (wn * 64 + j * 32 + l31) * SLD + ks * 16 + half * 8;
                bh[j] = *reinterpret_cast<const gemm_half8*>(Bh + off);
                bl[j] = *reinterpret_cast<const gemm_half8*>(Bl + off);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
    if (bad) atomicOr(a.status, 1u);
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


// ---------------------------------------------------------------------------------------------------------------------------
// General fp32-MFMA GEMM for the BACKWARD passes of the Linears and of the hoisted recurrent projections (training, row a9):
//     C[M,N] (+)= opA(A) . opB(B)         opA(A) = A [M,K] or A^T with A stored [K,M];   opB(B) = B [K,N] or B^T with B stored [N,K]
//   dx  = dG . W        (NN)      cube/networks/modules.py:505-563 (WaveRNN._train_forward differentiated by autograd in the reference)
//   dW  = dG^T . x      (TN: the contraction runs over all B x T rows — split over K across workgroups, partial tiles through a
//                        workspace, added in a fixed order by gemm_splitk_reduce_kernel: deterministic)
//   dW_hh = dG^T . h_prev   (TN with the B operand's rows shifted by one time step inside every sequence: row r reads B row r + shift
//                        when 0 <= r % period + shift < period, zero otherwise — h_prev is never materialised)
// Same 128 x 128 x 16 tile and fragment reads as gemm_nt_kernel; an operand whose global rows run along K is staged with 16-byte
// loads along K, one whose rows run along M / N with 16-byte loads along M / N and transposed on its way into LDS.
struct GemmGArgs {
    const float* A;
    const float* B;
    float* C;        // [M, ldc] or the split-K workspace [splits][M][N]
    int M, N, K, lda, ldb, ldc;
    int accumulate;
    int kchunk;      // K range of one split (multiple of GK); gridDim.z splits
    int splitk;      // 1: write partial tiles to C as [z][M][N]
    int b_shift, period;
};

template <bool AT, bool BT>   // AT: A stored [K, M];  BT: B stored [N, K]
__global__ __launch_bounds__(256) void gemm_general_kernel(GemmGArgs a) {
    __shared__ float As[GM * GLD];
    __shared__ float Bs[GN * GLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * GM, n0 = blockIdx.x * GN;
    const int half = lane >> 5, l31 = lane & 31;
    const int kbeg = blockIdx.z * a.kchunk;
    const int kend = kbeg + a.kchunk < a.K ? kbeg + a.kchunk : a.K;
    const bool veca = (a.lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.A) & 15) == 0);
    const bool vecb = (a.ldb % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.B) & 15) == 0);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // stage one 128 x 16 operand tile into LDS as [row (m or n)][k]: `kmajor` = the global rows run along K (transposed operand)
    auto stage = [&](const float* __restrict__ src, int ld, int rows, int r0, int k0, bool kmajor, bool vec, float* dst, bool shifted) {
        if (!kmajor) {
            // global [rows, K]: thread -> (row, 4 consecutive k)
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int idx = tid + it * 256;
                const int row = idx >> 2, k4 = (idx & 3) * 4;
                float v[4] = {0.f, 0.f, 0.f, 0.f};
                if (r0 + row < rows) {
                    const float* p = src + (size_t)(r0 + row) * ld + k0 + k4;
                    if (vec && k0 + k4 + 3 < kend) {
                        const float4 t = *reinterpret_cast<const float4*>(p);
                        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (k0 + k4 + e < kend) v[e] = p[e];
                    }
                }
                float* d = dst + row * GLD + k4;
                d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
            }
        } else {
            // global [K, rows]: thread -> (k, 4 consecutive rows); 16 k x 32 row-quads = 512 items
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int idx = tid + it * 256;
                const int k = idx >> 5, r4 = (idx & 31) * 4;
                float v[4] = {0.f, 0.f, 0.f, 0.f};
                int kr = k0 + k;
                bool ok = kr < kend;
                if (shifted && ok) {
                    const int t = kr % a.period + a.b_shift;
                    ok = t >= 0 && t < a.period;
                    kr += a.b_shift;
                }
                if (ok) {
                    const float* p = src + (size_t)kr * ld + r0 + r4;
                    if (vec && r0 + r4 + 3 < rows) {
                        const float4 t = *reinterpret_cast<const float4*>(p);
                        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (r0 + r4 + e < rows) v[e] = p[e];
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) dst[(r4 + e) * GLD + k] = v[e];
            }
        }
    };

    for (int k0 = kbeg; k0 < kend; k0 += GK) {
        __syncthreads();
        stage(a.A, a.lda, a.M, m0, k0, AT, veca, As, false);
        stage(a.B, a.ldb, a.N, n0, k0, !BT, vecb, Bs, a.b_shift != 0);
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
    float* Cb = a.splitk ? a.C + (size_t)blockIdx.z * a.M * a.N : a.C;
    const int ldc = a.splitk ? a.N : a.ldc;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + l31;
            if (n >= a.N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (m < a.M) {
                    float* y = Cb + (size_t)m * ldc + n;
                    float v = acc[i][j][r];
                    if (a.accumulate && !a.splitk) v += *y;
                    *y = v;
                }
            }
        }
}

// C[m, n] (+)= sum_z ws[z][m][n], z ascending (fixed order)
__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(const float* __restrict__ ws, float* __restrict__ C, int M, int N, int ldc, int splits,
                                                                 int accumulate) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)M * N) return;
    const int m = (int)(i / N), n = (int)(i % N);
    float s = accumulate ? C[(size_t)m * ldc + n] : 0.f;
    for (int z = 0; z < splits; ++z) s += ws[(size_t)z * M * N + i];
    C[(size_t)m * ldc + n] = s;
}

// out[c] (+)= sum_r x[r, c]: bias gradients of the Linears / recurrent projections.  Two fixed-order stages: partial sums of row
// ranges into ws [parts][C], then the parts in ascending order.
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, int R, int Cn, int ld, int rows_per_part, float* __restrict__ ws) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
    const int r0 = blockIdx.y * rows_per_part;
    const int r1 = r0 + rows_per_part < R ? r0 + rows_per_part : R;
    float s = 0.f;
    if (c < Cn)
        for (int r = r0 + q; r < r1; r += 4) s += x[(size_t)r * ld + c];
    red[q][threadIdx.x & 63] = s;
    __syncthreads();
    if (q == 0 && c < Cn) ws[(size_t)blockIdx.y * Cn + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ ws, int parts, int Cn, float* __restrict__ out, int accumulate) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= Cn) return;
    float s = accumulate ? out[c] : 0.f;
    for (int p = 0; p < parts; ++p) s += ws[(size_t)p * Cn + c];
    out[c] = s;
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

// status word of the split-precision GEMM, one per device (an operand beyond the fp16 range sets it; sticky until read)
static std::mutex g_split_mu;
static std::map<int, unsigned*> g_split_words;

static unsigned* split_status_word() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_split_mu);
    auto it = g_split_words.find(dev);
    if (it != g_split_words.end()) return it->second;
    unsigned* w = nullptr;
    if (hipMalloc(&w, sizeof(unsigned)) != hipSuccess) return nullptr;
    if (hipMemset(w, 0, sizeof(unsigned)) != hipSuccess) return nullptr;
    g_split_words[dev] = w;
    return w;
}

namespace ttsc {
unsigned* gemm_split_word_if_any() {     // (util.cpp: ttsc_split_status_collect) the current device's word if a split GEMM ever ran on it
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_split_mu);
    auto it = g_split_words.find(dev);
    return it == g_split_words.end() ? nullptr : it->second;
}
}  // namespace ttsc

extern "C" int32_t ttsc_gemm_split_status(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    unsigned* w = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_split_mu);
        auto it = g_split_words.find(dev);
        if (it == g_split_words.end()) return 0;
        w = it->second;
    }
    unsigned v = 0;
    if (hipMemcpy(&v, w, sizeof(unsigned), hipMemcpyDeviceToHost) != hipSuccess) return -1;   // synchronises
    if (v && hipMemset(w, 0, sizeof(unsigned)) != hipSuccess) return -1;
    return v ? 1 : 0;
}

extern "C" int32_t ttsc_linear_split_supported(int64_t M, int32_t N, int32_t K, int64_t ldx) {
    return M > 0 && N > 0 && K >= 32 && K % 4 == 0 && ldx % 4 == 0 && ldx >= K;
}

// ttsc_linear_forward's contract on the split-precision kernel (gemm_nt_f16x3_kernel).  lengths_dev [M / period] or NULL: row tiles wholly at or beyond
// their utterances' lengths are skipped (outputs unwritten).  TTSC_EINVAL when the shape / alignment needs the exact kernel.
extern "C" int ttsc_linear_forward_split(const float* x_dev, const float* w_dev, const float* bias_dev, float* y_dev, int64_t M, int32_t N, int32_t K,
                                         int64_t ldx, int64_t ldy, int32_t act, int32_t accumulate, const int32_t* lengths_dev, int32_t period,
                                         void* stream) {
    TTSC_REQUIRE(x_dev && w_dev && y_dev, "ttsc_linear_forward_split: null argument");
    TTSC_REQUIRE(M > 0 && N > 0 && K > 0 && ldx > 0 && ldy >= N && M < (1ll << 31) && ldx < (1ll << 31) && ldy < (1ll << 31),
                 "ttsc_linear_forward_split: bad shape M=%lld N=%d K=%d ldx=%lld ldy=%lld", (long long)M, N, K, (long long)ldx, (long long)ldy);
    TTSC_REQUIRE(!lengths_dev || (period > 0 && M % period == 0), "ttsc_linear_forward_split: rows must be [utterance][period] when lengths are given");
    if (!ttsc_linear_split_supported(M, N, K, ldx) || (((uintptr_t)x_dev | (uintptr_t)w_dev) & 15) != 0) {
        set_error("ttsc_linear_forward_split: shape / alignment not supported (K %% 4, ldx %% 4, 16-byte aligned operands, ldx >= K)");
        return TTSC_EINVAL;
    }
    unsigned* st = split_status_word();
    TTSC_REQUIRE(st, "ttsc_linear_forward_split: cannot allocate the status word");
    GemmSplitArgs a{x_dev, w_dev, bias_dev, y_dev, (int)M, N, K, (int)ldx, (int)ldy, act, accumulate, lengths_dev, period, st};
    dim3 grid((unsigned)ceil_div(N, GN), (unsigned)ceil_div(M, GM));
    hipLaunchKernelGGL(gemm_nt_f16x3_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("gemm_nt_f16x3_kernel launch failed: %s", hipGetErrorString(e));
        return TTSC_EHIP;
    }
    return TTSC_OK;
}

static int gemm_splits(int64_t M, int64_t N, int64_t K) {
    const int64_t tiles = ceil_div(M, GM) * ceil_div(N, GN);
    // (round 6: the text side's weight gradients at b = 16 are 1024 x 256 tiles over K = B x T = 3 840 rows — 16 workgroups walking the whole
    // contraction took 1.0-1.7 ms each, 8 ms of a 79 ms Cubegan step, profiles/r06_train_timeline_b16_start.txt; the threshold was K >= 4096)
    if (tiles >= 256 || K < 1024) return 1;
    int64_t s = ceil_div(768, tiles);                       // ~3 workgroups per CU
    const int64_t smax = ceil_div(K, 256);                  // at least 256 rows of K per split
    s = s < smax ? s : smax;
    return (int)(s < 1 ? 1 : (s > 512 ? 512 : s));
}

extern "C" size_t ttsc_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const int s = gemm_splits(M, N, K);
    return s > 1 ? (size_t)s * M * N * sizeof(float) : 0;
}

extern "C" int ttsc_gemm(int32_t transA, int32_t transB, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb,
                         float* C, int64_t ldc, int32_t accumulate, int64_t b_row_shift, int64_t b_period, void* ws_dev, size_t ws_bytes, void* stream) {
    TTSC_REQUIRE(A && B && C, "ttsc_gemm: null argument");
    TTSC_REQUIRE(M > 0 && N > 0 && K > 0 && M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31), "ttsc_gemm: bad shape M=%lld N=%lld K=%lld",
                 (long long)M, (long long)N, (long long)K);
    TTSC_REQUIRE(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N && lda < (1ll << 31) && ldb < (1ll << 31) && ldc < (1ll << 31),
                 "ttsc_gemm: leading dimensions too small (lda=%lld ldb=%lld ldc=%lld)", (long long)lda, (long long)ldb, (long long)ldc);
    TTSC_REQUIRE(b_row_shift == 0 || (!transB && b_period > 0 && b_row_shift > -b_period && b_row_shift < b_period),
                 "ttsc_gemm: the row shift applies to an untransposed B ([K, N]) with a positive period");
    const int splits = gemm_splits(M, N, K);
    TTSC_REQUIRE(splits == 1 || (ws_dev && ws_bytes >= (size_t)splits * M * N * sizeof(float)), "ttsc_gemm: workspace too small (need %zu bytes)",
                 (size_t)splits * M * N * sizeof(float));
    GemmGArgs a;
    a.A = A; a.B = B;
    a.M = (int)M; a.N = (int)N; a.K = (int)K;
    a.lda = (int)lda; a.ldb = (int)ldb; a.ldc = (int)ldc;
    a.accumulate = accumulate;
    a.splitk = splits > 1;
    a.C = a.splitk ? (float*)ws_dev : C;
    a.kchunk = (int)round_up(ceil_div(K, splits), GK);
    a.b_shift = (int)b_row_shift;
    a.period = b_period > 0 ? (int)b_period : 1;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)ceil_div(N, GN), (unsigned)ceil_div(M, GM), (unsigned)ceil_div(K, a.kchunk));
    const int nz = (int)grid.z;
    if (transA && transB) hipLaunchKernelGGL((gemm_general_kernel<true, true>), grid, dim3(256), 0, s, a);
    else if (transA) hipLaunchKernelGGL((gemm_general_kernel<true, false>), grid, dim3(256), 0, s, a);
    else if (transB) hipLaunchKernelGGL((gemm_general_kernel<false, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((gemm_general_kernel<false, false>), grid, dim3(256), 0, s, a);
    if (a.splitk)
        hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)ceil_div(M * N, 256)), dim3(256), 0, s, (const float*)ws_dev, C, (int)M, (int)N,
                           (int)ldc, nz, accumulate);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("gemm_general_kernel launch failed: %s", hipGetErrorString(e));
        return TTSC_EHIP;
    }
    return TTSC_OK;
}

static int colsum_parts(int64_t R) { return (int)(R >= 1024 ? (ceil_div(R, 256) > 256 ? 256 : ceil_div(R, 256)) : 1); }

extern "C" size_t ttsc_colsum_workspace_bytes(int64_t R, int64_t Cn) { return R > 0 && Cn > 0 ? (size_t)colsum_parts(R) * Cn * sizeof(float) : 0; }

extern "C" int ttsc_colsum(const float* x_dev, int64_t R, int64_t Cn, int64_t ld, float* out_dev, int32_t accumulate, void* ws_dev, size_t ws_bytes,
                           void* stream) {
    TTSC_REQUIRE(x_dev && out_dev && ws_dev, "ttsc_colsum: null argument");
    TTSC_REQUIRE(R > 0 && Cn > 0 && ld >= Cn && R < (1ll << 31) && ld < (1ll << 31), "ttsc_colsum: bad shape R=%lld C=%lld ld=%lld", (long long)R,
                 (long long)Cn, (long long)ld);
    const int parts = colsum_parts(R);
    TTSC_REQUIRE(ws_bytes >= (size_t)parts * Cn * sizeof(float), "ttsc_colsum: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int rpp = (int)ceil_div(R, parts);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)ceil_div(Cn, 64), (unsigned)parts), dim3(256), 0, s, x_dev, (int)R, (int)Cn, (int)ld, rpp,
                       (float*)ws_dev);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)ceil_div(Cn, 256)), dim3(256), 0, s, (const float*)ws_dev, parts, (int)Cn, out_dev, accumulate);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("colsum kernels launch failed: %s", hipGetErrorString(e));
        return TTSC_EHIP;
    }
    return TTSC_OK;
}
