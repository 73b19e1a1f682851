// What the f16 matrix pipe SUSTAINS on this chip, by operand data: a register-only loop of v_mfma_f32_32x32x16_f16 (no memory, no LDS,
// 4 independent accumulators per wave, 8 waves per CU, every CU busy for ~0.2 s) with (a) all-zero operands, (b) random fp16 operands of
// unit scale (what the generator's hi halves look like), (c) the split-precision operand mix: products lo x hi, hi x lo, hi x hi with
// lo = 2^-11-scale random values.  MI355X clocks to its power budget (MI355X_MICROARCH.md, "DVFS give-back"), so the nominal 2.5 PF only
// holds for (a); the ratio (b or c) / nominal is the ceiling any MFMA-bound kernel has on real data.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/mfma_power tools/probes/mfma_power_probe.hip && /tmp/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(512) void mfma_loop(const half8* ops, int iters, float* sink) {
    const int lane = threadIdx.x & 63;
    half8 a0 = ops[lane], a1 = ops[64 + lane], b0 = ops[128 + lane], b1 = ops[192 + lane];   // a0/b0: "hi", a1/b1: "lo"
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
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
    for (int i = 0; i < 4; ++i) t += acc[i][0] + acc[i][7];
    if (t == 123.456f) sink[0] = t;
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    half8* d;
    float* sink;
    hipMalloc(&d, 256 * sizeof(half8));
    hipMalloc(&sink, 4);
    const char* names[3] = {"zeros", "random fp16 (all four operands unit scale)", "split mix (hi unit scale, lo 2^-11 scale)"};
    for (int mode = 0; mode < 3; ++mode) {
        std::vector<_Float16> h(256 * 8);
        srand(1);
        for (int i = 0; i < 256 * 8; ++i) {
            const float u = (float)rand() / RAND_MAX * 2.f - 1.f;
            const bool lo = (i / (64 * 8)) & 1;   // blocks 1 and 3 are the "lo" operands
            h[i] = (_Float16)(mode == 0 ? 0.f : (mode == 2 && lo ? u * 4.8828125e-4f : u));
        }
        hipMemcpy(d, h.data(), h.size() * sizeof(_Float16), hipMemcpyHostToDevice);
        const int iters = 40000;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        mfma_loop<<<cus, 512>>>(d, 2000, sink);   // warm-up
        hipDeviceSynchronize();
        hipEventRecord(e0);
        mfma_loop<<<cus, 512>>>(d, iters, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)cus * 8 * iters * 12 * (2.0 * 32 * 32 * 16);
        printf("%-52s %8.1f ms  %7.1f TFLOP/s  = %.3f of 2500\n", names[mode], ms, flops / ms / 1e9, flops / ms / 1e9 / 2500.0);
    }
    return 0;
}
