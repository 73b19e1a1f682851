// STFT-magnitude / mel-spectrogram plumbing around the MFMA GEMM (gemm.hip), for
//   * hifigan.meldataset.mel_spectrogram [EXTERNAL; called at cube/networks/cubegan.py:137-138,247-248]: the 45 x mel-L1 loss
//     of the GAN step (forward AND backward, the generated waveform carries the gradient), and
//   * MelVocoder.melspectrogram (cube/io_utils/vocoder.py:54-98): feature extraction, STFT 1024 / hop 240 -> 80 mel ->
//     log10(max(1e-5, .))  (SURVEY.md §8 row f4).
// The DFT is a GEMM: frames [M, n_fft] (rows of the padded signal at a constant stride `hop`: no gather, the GEMM reads the
// overlapping rows in place) x basis [2*NB, n_fft] (window folded in: rows 0..NB-1 = hann*cos, NB..2NB-1 = -hann*sin), then
// the mel projection is a second GEMM.  What remains are the element-wise kernels below.
#include "common.hpp"

namespace ttsc {

// mag[m, k] = sqrt(re^2 + im^2 + eps)  for reim [M, 2*NB] (re | im) -> mag [M, ldm] (columns >= NB zeroed: GEMM padding)
__global__ __launch_bounds__(256) void stft_mag_kernel(const float* __restrict__ reim, long M, int NB, int ldm, float eps, float* __restrict__ mag) {
    const long total = M * ldm;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long m = i / ldm;
        const int k = (int)(i - m * ldm);
        float v = 0.f;
        if (k < NB) {
            const float re = reim[m * 2 * NB + k], im = reim[m * 2 * NB + NB + k];
            v = sqrtf(re * re + im * im + eps);
        }
        mag[i] = v;
    }
}

// d(re|im)[m, k] = dmag[m, k] * (re|im)[m, k] / mag[m, k]
__global__ __launch_bounds__(256) void stft_mag_bwd_kernel(const float* __restrict__ dmag, const float* __restrict__ reim,
                                                           const float* __restrict__ mag, long M, int NB, int ldm, float* __restrict__ dreim) {
    const long total = M * NB;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long m = i / NB;
        const int k = (int)(i - m * NB);
        const float mg = mag[m * ldm + k];
        const float g = mg > 0.f ? dmag[m * ldm + k] / mg : 0.f;
        dreim[m * 2 * NB + k] = g * reim[m * 2 * NB + k];
        dreim[m * 2 * NB + NB + k] = g * reim[m * 2 * NB + NB + k];
    }
}

// y = scale * log(max(x, minv));   dx = dy * scale / x where x > minv, else 0
__global__ __launch_bounds__(256) void log_clamp_kernel(const float* __restrict__ x, long n, float minv, float scale, float* __restrict__ y) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = scale * logf(fmaxf(x[i], minv));
}
__global__ __launch_bounds__(256) void log_clamp_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, long n, float minv,
                                                            float scale, float* __restrict__ dx) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        dx[i] = x[i] > minv ? dy[i] * scale / x[i] : 0.f;
}

// backward of the framing: y[b, t] = sum over frames f with f*hop <= t < f*hop + n_fft of frames[b, f, t - f*hop]   (gather
// form: one thread per output sample, at most ceil(n_fft / hop) terms, fixed summation order -> deterministic)
__global__ __launch_bounds__(256) void overlap_add_kernel(const float* __restrict__ frames, int F, int n_fft, int hop, long Lp,
                                                          float* __restrict__ y) {
    const int b = blockIdx.y;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < Lp; t += (long)gridDim.x * blockDim.x) {
        long f_hi = t / hop;
        if (f_hi > F - 1) f_hi = F - 1;
        const long f_lo = t < n_fft ? 0 : (t - n_fft) / hop + 1;   // smallest f with f*hop + n_fft > t
        float s = 0.f;
        for (long f = f_lo; f <= f_hi; ++f) s += frames[((size_t)b * F + f) * n_fft + (t - f * hop)];
        y[(size_t)b * Lp + t] = s;
    }
}

static inline int grid_for(long n) {
    long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace ttsc

using namespace ttsc;

#define TTSC_LAUNCH_CHECK(name)                                              \
    do {                                                                     \
        hipError_t _e = hipGetLastError();                                   \
        if (_e != hipSuccess) {                                              \
            set_error(name " launch failed: %s", hipGetErrorString(_e));     \
            return TTSC_EHIP;                                                \
        }                                                                    \
    } while (0)

extern "C" int ttsc_stft_mag(const float* reim_dev, int64_t M, int32_t NB, int32_t ldm, float eps, float* mag_dev, void* stream) {
    TTSC_REQUIRE(reim_dev && mag_dev && M > 0 && NB > 0 && ldm >= NB, "ttsc_stft_mag: bad argument");
    hipLaunchKernelGGL(stft_mag_kernel, dim3(grid_for(M * ldm)), dim3(256), 0, (hipStream_t)stream, reim_dev, (long)M, NB, ldm, eps, mag_dev);
    TTSC_LAUNCH_CHECK("stft_mag_kernel");
    return TTSC_OK;
}

extern "C" int ttsc_stft_mag_backward(const float* dmag_dev, const float* reim_dev, const float* mag_dev, int64_t M, int32_t NB, int32_t ldm,
                                      float* dreim_dev, void* stream) {
    TTSC_REQUIRE(dmag_dev && reim_dev && mag_dev && dreim_dev && M > 0 && NB > 0 && ldm >= NB, "ttsc_stft_mag_backward: bad argument");
    hipLaunchKernelGGL(stft_mag_bwd_kernel, dim3(grid_for(M * NB)), dim3(256), 0, (hipStream_t)stream, dmag_dev, reim_dev, mag_dev, (long)M, NB,
                       ldm, dreim_dev);
    TTSC_LAUNCH_CHECK("stft_mag_bwd_kernel");
    return TTSC_OK;
}

extern "C" int ttsc_log_clamp(const float* x_dev, int64_t n, float minv, float scale, float* y_dev, void* stream) {
    TTSC_REQUIRE(x_dev && y_dev && n > 0 && minv > 0.f, "ttsc_log_clamp: bad argument");
    hipLaunchKernelGGL(log_clamp_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x_dev, (long)n, minv, scale, y_dev);
    TTSC_LAUNCH_CHECK("log_clamp_kernel");
    return TTSC_OK;
}

extern "C" int ttsc_log_clamp_backward(const float* dy_dev, const float* x_dev, int64_t n, float minv, float scale, float* dx_dev, void* stream) {
    TTSC_REQUIRE(dy_dev && x_dev && dx_dev && n > 0 && minv > 0.f, "ttsc_log_clamp_backward: bad argument");
    hipLaunchKernelGGL(log_clamp_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dy_dev, x_dev, (long)n, minv, scale, dx_dev);
    TTSC_LAUNCH_CHECK("log_clamp_bwd_kernel");
    return TTSC_OK;
}

extern "C" int ttsc_overlap_add(const float* frames_dev, int32_t B, int32_t F, int32_t n_fft, int32_t hop, int64_t Lp, float* y_dev, void* stream) {
    TTSC_REQUIRE(frames_dev && y_dev && B > 0 && F > 0 && n_fft > 0 && hop > 0 && Lp >= (int64_t)(F - 1) * hop + n_fft, "ttsc_overlap_add: bad argument");
    hipLaunchKernelGGL(overlap_add_kernel, dim3(grid_for(Lp), (unsigned)B), dim3(256), 0, (hipStream_t)stream, frames_dev, F, n_fft, hop, (long)Lp,
                       y_dev);
    TTSC_LAUNCH_CHECK("overlap_add_kernel");
    return TTSC_OK;
}
