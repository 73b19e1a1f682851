// Measurement helper (bench.py, tools/): what the f16 matrix pipe of THIS device sustains, by operand data.
//
// MI355X clocks to its power budget (MI355X_MICROARCH.md, "DVFS give-back"): a register-only loop of v_mfma_f32_32x32x16_f16 reaches ~0.95 of the
// nominal 2.5 PFLOP/s on all-zero operands and ~0.66 - 0.68 on random fp16 operands (round-4 measurement, tools/probes/mfma_power_probe.hip:
// 2363 / 1655 / 1706 TFLOP/s).  Every MFMA-bound kernel of the generator runs on real data, so `roofline.peak` (nominal, as the contract asks)
// overstates what is reachable by that factor; bench.py reports both.
#include "common.hpp"

namespace ttsc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// 4 independent accumulators per wave, the three products of the split-precision scheme per step (lo x hi, hi x lo, hi x hi)
__global__ __launch_bounds__(512) void mfma_sustained_kernel(const half8* ops, int iters, float* sink) {
    const int lane = threadIdx.x & 63;
    const half8 a0 = ops[lane], a1 = ops[64 + lane], b0 = ops[128 + lane], b1 = ops[192 + lane];   // a0 / b0: "hi", a1 / b1: "lo"
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[i], 0, 0, 0);
        }
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) t += acc[i][0] + acc[i][7];
    if (t == 123.456f) sink[0] = t;   // keeps the accumulators alive
}

}  // namespace ttsc

using namespace ttsc;

// mode 0: all-zero operands; 1: random fp16 operands of unit scale; 2: the split-precision mix (hi operands unit scale, lo operands 2^-11 scale).
// Runs the loop for ~ms_target milliseconds on every CU (8 waves per CU) on `stream`, returns executed dense f16 MFMA TFLOP/s in *tflops_out.
extern "C" int ttsc_probe_mfma_tflops(int32_t mode, double ms_target, double* tflops_out, void* stream) {
    TTSC_REQUIRE(tflops_out && mode >= 0 && mode <= 2 && ms_target > 0 && ms_target <= 2000, "ttsc_probe_mfma_tflops: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const int cus = device_cus();
    std::vector<_Float16> h(256 * 8);
    uint64_t st = 0x9E3779B97F4A7C15ull;
    for (size_t i = 0; i < h.size(); ++i) {
        st = st * 6364136223846793005ull + 1442695040888963407ull;
        const float u = (float)((st >> 40) & 0xffffff) / 8388608.f - 1.f;   // [-1, 1)
        const bool lo = (i / (64 * 8)) & 1;
        h[i] = (_Float16)(mode == 0 ? 0.f : (mode == 2 && lo ? u * 4.8828125e-4f : u));
    }
    half8* d = nullptr;
    float* sink = nullptr;
    TTSC_HIP_CHECK(hipMalloc((void**)&d, h.size() * sizeof(_Float16)));
    if (hipMalloc((void**)&sink, sizeof(float)) != hipSuccess) {
        (void)hipFree(d);
        set_error("ttsc_probe_mfma_tflops: allocation failed");
        return TTSC_ENOMEM;
    }
    int rc = TTSC_OK;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipMemcpy(d, h.data(), h.size() * sizeof(_Float16), hipMemcpyHostToDevice) != hipSuccess || hipEventCreate(&e0) != hipSuccess ||
        hipEventCreate(&e1) != hipSuccess)
        rc = TTSC_EHIP;
    if (!rc) {
        // 12 MFMAs of 32 nominal cycles per iteration and wave, two waves per SIMD: ~0.32 us per iteration at 2.4 GHz
        const int iters = (int)(ms_target * 1e3 / 0.32);
        hipLaunchKernelGGL(mfma_sustained_kernel, dim3(cus), dim3(512), 0, s, d, iters / 10 + 1, sink);   // warm-up: clocks settle
        (void)hipEventRecord(e0, s);
        hipLaunchKernelGGL(mfma_sustained_kernel, dim3(cus), dim3(512), 0, s, d, iters, sink);
        (void)hipEventRecord(e1, s);
        float ms = 0.f;
        if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || !(ms > 0.f)) {
            set_error("ttsc_probe_mfma_tflops: timing failed");
            rc = TTSC_EHIP;
        } else {
            *tflops_out = (double)cus * 8 * iters * 12 * (2.0 * 32 * 32 * 16) / ms / 1e9;
        }
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(d);
    (void)hipFree(sink);
    return rc;
}
