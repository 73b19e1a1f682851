// Device-side duration -> alignment for the mel decoders.
//
// The reference turns the duration head into a frame->phone map on the HOST: `torch.argmax(...).detach().cpu().numpy()`
// followed by nested Python loops (cube/networks/modules.py:946-953 Languasito2._get_cond_selection, textcoder.py:160-166)
// and gathers rows through a numpy index array (modules.py:1043-1053 _expand_i, textcoder.py:291-302 _expand).  Here the
// argmax, the exclusive scan over phones, the scatter of phone indices and the row gather are kernels; the host only reads
// the B frame counts it needs to size the output tensor.
#include "common.hpp"

namespace ttsc {

// one workgroup per utterance: durs[b,p] = argmax_d logits[b,p,d] (first maximum, like torch.argmax) for p < len[b], else 0;
// f2p[b, start_p .. start_p + durs[b,p]) = p;  flen[b] = min(sum_p durs[b,p], Fcap)
__global__ __launch_bounds__(256) void align_durations_kernel(const float* __restrict__ logits, const int* __restrict__ len, int N, int D,
                                                              int* __restrict__ durs, int* __restrict__ f2p, int* __restrict__ flen,
                                                              int Fcap) {
    __shared__ int part[256];
    __shared__ int run;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = len ? (len[b] < N ? len[b] : N) : N;
    if (tid == 0) run = 0;
    __syncthreads();
    for (int p0 = 0; p0 < n; p0 += 256) {
        const int p = p0 + tid;
        int d = 0;
        if (p < n) {
            const float* row = logits + ((size_t)b * N + p) * D;
            float best = row[0];
            for (int k = 1; k < D; ++k) {
                const float v = row[k];
                if (v > best) {   // strict: keeps the FIRST maximum; NaN never wins (torch would return a NaN's index: not produced here)
                    best = v;
                    d = k;
                }
            }
        }
        part[tid] = d;
        __syncthreads();
        // Hillis-Steele inclusive scan over the 256 phones of this pass
        for (int off = 1; off < 256; off <<= 1) {
            const int v = tid >= off ? part[tid - off] : 0;
            __syncthreads();
            part[tid] += v;
            __syncthreads();
        }
        const int start = run + part[tid] - d;
        if (p < n) {
            durs[(size_t)b * N + p] = d;
            for (int k = 0; k < d; ++k)
                if (start + k < Fcap) f2p[(size_t)b * Fcap + start + k] = p;
        }
        __syncthreads();
        if (tid == 255) run += part[255];
        __syncthreads();
    }
    for (int p = n + tid; p < N; p += 256) durs[(size_t)b * N + p] = 0;
    if (tid == 0) flen[b] = run < Fcap ? run : Fcap;
}

// out[b, f, :] = x[b, row(b, f), :],  row = f2p[b, f*stride] for f < flen[b]/stride; beyond that the padding rule of the
// reference's gathers: stride 1 -> the utterance's last aligned row (modules.py:1049-1051), stride > 1 -> row N-1
// (textcoder.py:298-300); an utterance without frames reads row 0.
__global__ __launch_bounds__(256) void expand_rows_kernel(const float* __restrict__ x, const int* __restrict__ f2p, const int* __restrict__ flen,
                                                          int N, int C, int Fcap, int stride, int F, float* __restrict__ out) {
    const int f = blockIdx.x, b = blockIdx.y;
    const int nf = flen[b] / stride;
    int row;
    if (f < nf)
        row = f2p[(size_t)b * Fcap + (size_t)f * stride];
    else if (nf == 0)
        row = 0;
    else
        row = stride == 1 ? f2p[(size_t)b * Fcap + (size_t)(nf - 1)] : N - 1;
    const float* src = x + ((size_t)b * N + row) * C;
    float* dst = out + ((size_t)b * F + f) * C;
    if ((C & 3) == 0) {
        for (int i = threadIdx.x; i < (C >> 2); i += blockDim.x) reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[i];
    } else {
        for (int i = threadIdx.x; i < C; i += blockDim.x) dst[i] = src[i];
    }
}

// Input of Languasito2's conditioning recurrence in ONE launch (modules.py:962-994 of the reference: vuv = round(p_vuv), pitch = p * max_pitch * vuv,
// g = cat[expand(g_phoneme), pitch / max_pitch]): out[b, f, :C] = the gathered phoneme row (expand_rows_kernel's rule at stride 1), out[b, f, C] = pitch *
// (1 / max_pitch) — the product form torch's division by a host scalar takes on the device —, out[b, f, C + 1 .. Cp) = 0 (the split GEMM's K % 4 padding);
// pitch[b, f] is written beside it (X['y_pitch']).  The reference's graph is nine elementwise launches here, ~9 us each on the critical path of a sentence.
__global__ __launch_bounds__(256) void cond_input_kernel(const float* __restrict__ x, const int* __restrict__ f2p, const int* __restrict__ flen,
                                                         const float* __restrict__ op, float max_pitch, int N, int C, int Cp, int Fcap, int F,
                                                         float* __restrict__ pitch, float* __restrict__ out) {
    const int f = blockIdx.x, b = blockIdx.y;
    const int nf = flen[b];
    const int row = f < nf ? f2p[(size_t)b * Fcap + f] : (nf == 0 ? 0 : f2p[(size_t)b * Fcap + (nf - 1)]);
    const float* src = x + ((size_t)b * N + row) * C;
    float* dst = out + ((size_t)b * F + f) * Cp;
    if ((C & 3) == 0) {
        for (int i = threadIdx.x; i < (C >> 2); i += blockDim.x) reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[i];
    } else {
        for (int i = threadIdx.x; i < C; i += blockDim.x) dst[i] = src[i];
    }
    if (threadIdx.x == 0) {
        const float* o = op + ((size_t)b * F + f) * 2;
        const float vuv = rintf(o[1]);                    // torch.round: half to even
        const float p = (o[0] * max_pitch) * vuv;
        pitch[(size_t)b * F + f] = p;
        const float inv = 1.0f / max_pitch;
        dst[C] = p * inv;
        for (int i = C + 1; i < Cp; ++i) dst[i] = 0.f;
    }
}

}  // namespace ttsc

using namespace ttsc;

extern "C" int ttsc_cond_input(const float* g_dev, const int32_t* f2p_dev, const int32_t* flen_dev, const float* pitch_out_dev, float max_pitch, int32_t B,
                               int32_t N, int32_t C, int32_t Cp, int32_t Fcap, int32_t F, float* pitch_dev, float* out_dev, void* stream) {
    TTSC_REQUIRE(g_dev && f2p_dev && flen_dev && pitch_out_dev && pitch_dev && out_dev, "ttsc_cond_input: null argument");
    TTSC_REQUIRE(B > 0 && N > 0 && C > 0 && Cp > C && Fcap > 0 && F > 0 && max_pitch > 0.f, "ttsc_cond_input: bad sizes (B=%d N=%d C=%d Cp=%d F=%d)", B, N, C, Cp, F);
    hipLaunchKernelGGL(cond_input_kernel, dim3((unsigned)F, (unsigned)B), dim3(C >= 1024 ? 256 : 64), 0, (hipStream_t)stream, g_dev, f2p_dev, flen_dev,
                       pitch_out_dev, max_pitch, N, C, Cp, Fcap, F, pitch_dev, out_dev);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("cond_input_kernel launch failed: %s", hipGetErrorString(e));
        return TTSC_EHIP;
    }
    return TTSC_OK;
}

extern "C" int ttsc_align_durations(const float* logits_dev, const int32_t* len_dev, int32_t B, int32_t N, int32_t D, int32_t* durs_dev,
                                    int32_t* f2p_dev, int32_t* flen_dev, int32_t Fcap, void* stream) {
    TTSC_REQUIRE(logits_dev && durs_dev && f2p_dev && flen_dev, "ttsc_align_durations: null argument");
    TTSC_REQUIRE(B > 0 && N > 0 && D > 0 && Fcap > 0, "ttsc_align_durations: bad sizes (B=%d N=%d D=%d Fcap=%d)", B, N, D, Fcap);
    hipLaunchKernelGGL(align_durations_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, logits_dev, len_dev, N, D, durs_dev,
                       f2p_dev, flen_dev, Fcap);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("align_durations_kernel launch failed: %s", hipGetErrorString(e));
        return TTSC_EHIP;
    }
    return TTSC_OK;
}

extern "C" int ttsc_expand_rows(const float* x_dev, const int32_t* f2p_dev, const int32_t* flen_dev, int32_t B, int32_t N, int32_t C,
                                int32_t Fcap, int32_t stride, int32_t F, float* out_dev, void* stream) {
    TTSC_REQUIRE(x_dev && f2p_dev && flen_dev && out_dev, "ttsc_expand_rows: null argument");
    TTSC_REQUIRE(B > 0 && N > 0 && C > 0 && Fcap > 0 && stride > 0 && F > 0, "ttsc_expand_rows: bad sizes");
    const int threads = C >= 1024 ? 256 : (C >= 256 ? 64 : 64);
    hipLaunchKernelGGL(expand_rows_kernel, dim3((unsigned)F, (unsigned)B), dim3(threads), 0, (hipStream_t)stream, x_dev, f2p_dev, flen_dev, N,
                       C, Fcap, stride, F, out_dev);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("expand_rows_kernel launch failed: %s", hipGetErrorString(e));
        return TTSC_EHIP;
    }
    return TTSC_OK;
}
