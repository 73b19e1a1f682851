// Where the power budget goes besides the MFMAs: the register-only loop of mfma_power_probe.hip (split-precision operand mix on random data)
// with (a) nothing else, (b) R conflict-free ds_read_b128 per 12 MFMAs feeding the B operands (R = 6, 10, 12), (c) V plain VALU instructions
// per MFMA (V = 2, 4).  Sustained TFLOP/s per variant: the chip clocks to its power budget, so every extra picojoule per MFMA shows up as rate.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/mfma_energy tools/probes/mfma_energy_probe.hip && /tmp/mfma_energy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

template <int R, int V>
__global__ __launch_bounds__(512) void loop_kernel(const half8* ops, int iters, float* sink) {
    __shared__ half8 lds[64 * 24];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 64 * 24; i += 512) lds[i] = ops[i & 255];
    __syncthreads();
    half8 a0 = ops[lane], a1 = ops[64 + lane];
    half8 b[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) b[i] = ops[128 + ((i & 1) ? 64 : 0) + lane];   // even: hi, odd: lo
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float v0 = lane * 1e-3f, v1 = 1.0001f;
    const half8* lp = lds + ((wv * 3) % 12) * 64 + lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            const int i = q & 3, term = q >> 2;
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(term == 0 ? a1 : a0, b[(2 * i + (term == 1 ? 1 : 0)) % 12], acc[i], 0, 0, 0);
            if (q < R) b[q] = lp[(q % 12) * 64 + ((it & 1) ? 64 * 12 : 0)];
#pragma unroll
            for (int v = 0; v < V; ++v) v0 = __builtin_fmaf(v0, v1, 1e-7f);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float t = v0;
#pragma unroll
    for (int i = 0; i < 4; ++i) t += acc[i][0] + acc[i][7];
    if (t == 123.456f) sink[0] = t;
}

template <int R, int V>
static void run(const char* name, const half8* d, float* sink, int cus) {
    const int iters = 30000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    loop_kernel<R, V><<<cus, 512>>>(d, 3000, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    loop_kernel<R, V><<<cus, 512>>>(d, iters, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)cus * 8 * iters * 12 * (2.0 * 32 * 32 * 16);
    printf("%-60s %8.1f ms  %7.1f TFLOP/s  = %.3f of 2500\n", name, ms, flops / ms / 1e9, flops / ms / 1e9 / 2500.0);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    std::vector<_Float16> h(256 * 8);
    srand(1);
    for (int i = 0; i < 256 * 8; ++i) {
        const float u = (float)rand() / RAND_MAX * 2.f - 1.f;
        const bool lo = (i / (64 * 8)) & 1;
        h[i] = (_Float16)(lo ? u * 4.8828125e-4f : u);
    }
    half8* d;
    float* sink;
    hipMalloc(&d, 256 * sizeof(half8));
    hipMalloc(&sink, 4);
    hipMemcpy(d, h.data(), h.size() * sizeof(_Float16), hipMemcpyHostToDevice);
    run<0, 0>("MFMA only (split mix, random data)", d, sink, cus);
    run<6, 0>("+ 6 ds_read_b128 per 12 MFMAs", d, sink, cus);
    run<10, 0>("+ 10 ds_read_b128 per 12 MFMAs (32-channel chain)", d, sink, cus);
    run<12, 0>("+ 12 ds_read_b128 per 12 MFMAs", d, sink, cus);
    run<0, 2>("+ 2 VALU per MFMA", d, sink, cus);
    run<0, 4>("+ 4 VALU per MFMA", d, sink, cus);
    run<10, 2>("+ 10 ds_read_b128 per 12 MFMAs + 2 VALU per MFMA", d, sink, cus);
    return 0;
}
