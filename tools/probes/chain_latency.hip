// Issue interval of DEPENDENT instruction chains on one wave (gfx950): v_mfma_f32_4x4x1 (1 and 2 interleaved chains),
// v_fma_f32 (1 and 4 interleaved chains).  hipcc --offload-arch=gfx950 -O2 -o tools/probes/chain.bin tools/probes/chain_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int N = 4096;

__global__ void k(float* out, long long* t, float a, float b) {
    f32x4 c0 = {0, 0, 0, 0}, c1 = {1, 1, 1, 1};
    long long w0, w1, s0, s1;
    // 1 MFMA chain
    s0 = clock64(); w0 = wall_clock64();
#pragma unroll 16
    for (int i = 0; i < N; ++i) c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
    s1 = clock64(); w1 = wall_clock64();
    t[0] = s1 - s0; t[1] = w1 - w0;
    // 2 MFMA chains
    s0 = clock64(); w0 = wall_clock64();
#pragma unroll 16
    for (int i = 0; i < N; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(b, a, c1, 0, 0, 0);
    }
    s1 = clock64(); w1 = wall_clock64();
    t[2] = s1 - s0; t[3] = w1 - w0;
    // 1 fma chain
    float x0 = a, x1 = b, x2 = a + 1, x3 = b + 1;
    s0 = clock64(); w0 = wall_clock64();
#pragma unroll 16
    for (int i = 0; i < N; ++i) x0 = __builtin_fmaf(x0, a, b);
    s1 = clock64(); w1 = wall_clock64();
    t[4] = s1 - s0; t[5] = w1 - w0;
    // 4 fma chains
    s0 = clock64(); w0 = wall_clock64();
#pragma unroll 16
    for (int i = 0; i < N; ++i) {
        x0 = __builtin_fmaf(x0, a, b);
        x1 = __builtin_fmaf(x1, a, b);
        x2 = __builtin_fmaf(x2, a, b);
        x3 = __builtin_fmaf(x3, a, b);
    }
    s1 = clock64(); w1 = wall_clock64();
    t[6] = s1 - s0; t[7] = w1 - w0;
    out[threadIdx.x] = c0[0] + c1[1] + x0 + x1 + x2 + x3;
}

int main() {
    float* o; long long* t;
    hipMalloc(&o, 256); hipMalloc(&t, 64);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, t, 0.5f, 0.25f);
    long long h[8];
    hipMemcpy(h, t, 64, hipMemcpyDeviceToHost);
    const char* nm[4] = {"mfma 4x4x1, 1 chain ", "mfma 4x4x1, 2 chains", "v_fma_f32, 1 chain  ", "v_fma_f32, 4 chains "};
    const int per[4] = {1, 2, 1, 4};
    for (int i = 0; i < 4; ++i)
        printf("%s: %.2f s_memtime ticks / iteration, %.2f ns / iteration (%d instr)\n", nm[i], (double)h[2 * i] / N, (double)h[2 * i + 1] * 10.0 / N, per[i]);
    return 0;
}
