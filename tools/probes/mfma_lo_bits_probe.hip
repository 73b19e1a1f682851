// Round 5 probe (VERDICT r4 #1e): does the sustained rate of the f16 matrix pipe under the chip's power budget depend on how many mantissa
// bits of the LO halves of the split-precision operands are live?  Same register-only loop as mfma_power_probe.hip (12 MFMAs per iteration:
// lo_w x hi_x, hi_w x lo_x, hi_w x hi_x on 4 accumulators), operands ROTATING over four register sets per product so that consecutive MFMAs
// see different bit patterns (as the kernels' fragments do), and the lo operands rounded to n explicit mantissa bits (10 = as they are).
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/lobits tools/probes/mfma_lo_bits_probe.hip && /tmp/lobits
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// ops: [set 0..3][a_hi, a_lo, b_hi, b_lo][64 lanes]
template <int ROT>   // 2: A and B operands change with every MFMA (four register sets); 1: only B changes; 0: neither
__global__ __launch_bounds__(512) void mfma_loop(const half8* ops, int iters, float* sink) {
    const int lane = threadIdx.x & 63;
    half8 ah[4], al[4], bh[4], bl[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        ah[s] = ops[(s * 4 + 0) * 64 + lane];
        al[s] = ops[(s * 4 + 1) * 64 + lane];
        bh[s] = ops[(s * 4 + 2) * 64 + lane];
        bl[s] = ops[(s * 4 + 3) * 64 + lane];
    }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ROT == 2 ? i : 0], bh[ROT >= 1 ? i : 0], acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ROT == 2 ? i : 0], bl[ROT >= 1 ? i : 0], acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ROT == 2 ? i : 0], bh[ROT >= 1 ? i : 0], acc[i], 0, 0, 0);
    }
    float t = 0.f;
    for (int i = 0; i < 4; ++i) t += acc[i][0] + acc[i][7];
    if (t == 123.456f) sink[0] = t;
}

static _Float16 round_bits(_Float16 v, int nbits) {   // round the 10 explicit mantissa bits to nbits (nearest, ties away)
    if (nbits >= 10) return v;
    unsigned short u;
    memcpy(&u, &v, 2);
    const int drop = 10 - nbits;
    u = (unsigned short)((u + (1u << (drop - 1))) & ~((1u << drop) - 1));
    memcpy(&v, &u, 2);
    return v;
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    half8* d;
    float* sink;
    hipMalloc(&d, 16 * 64 * sizeof(half8));
    hipMalloc(&sink, 4);
    struct Mode { const char* name; int kind; int lo_bits_w, lo_bits_x, hi_bits; int rot = 2; };
    // kind 0 zeros, 1 split mix (values v ~ U(-1,1) * scale split into hi + lo)
    const Mode modes[] = {
        {"zeros", 0, 10, 10, 10},
        {"split mix, lo as is (10 bits)", 1, 10, 10, 10},
        {"split mix, lo 8 bits", 1, 8, 8, 10},
        {"split mix, lo 6 bits", 1, 6, 6, 10},
        {"split mix, lo 5 bits", 1, 5, 5, 10},
        {"split mix, lo 4 bits", 1, 4, 4, 10},
        {"split mix, lo 2 bits", 1, 2, 2, 10},
        {"split mix, lo 0 bits (powers of two)", 1, 0, 0, 10},
        {"split mix, lo = 0 (two of three products idle)", 2, 10, 10, 10},
        {"split mix, weights' lo 4 bits, activations' lo 10", 1, 4, 10, 10},
        {"split mix, weights' lo 10, activations' lo 4 bits", 1, 10, 4, 10},
        {"split mix, lo 10 bits, hi 7 bits (bf16-like hi)", 1, 10, 10, 7},
        {"split mix, lo 10 bits, only B operands change", 1, 10, 10, 10, 1},
        {"split mix, lo 10 bits, no operand changes", 1, 10, 10, 10, 0},
        {"split mix, lo 5 bits, only B operands change", 1, 5, 5, 10, 1},
        {"split mix, lo as is (10 bits), repeat", 1, 10, 10, 10},
    };
    for (const Mode& m : modes) {
        std::vector<_Float16> h(16 * 64 * 8);
        srand(1);
        for (int blk = 0; blk < 16; ++blk) {
            const int which = blk & 3;   // 0 a_hi 1 a_lo 2 b_hi 3 b_lo
            for (int i = 0; i < 64 * 8; ++i) {
                const float u = ((float)rand() / RAND_MAX * 2.f - 1.f) * (which < 2 ? 700.f : 20000.f);   // weights ~2^9.5, activations ~2^14: the kernels' pre-scaled ranges
                _Float16 hi = (_Float16)u;
                _Float16 lo = (_Float16)(u - (float)hi);
                hi = round_bits(hi, m.hi_bits);
                lo = round_bits(lo, (which == 1) ? m.lo_bits_w : m.lo_bits_x);
                _Float16 v = (which & 1) ? lo : hi;
                if (m.kind == 0) v = (_Float16)0.f;
                if (m.kind == 2 && (which & 1)) v = (_Float16)0.f;
                h[(size_t)blk * 64 * 8 + i] = v;
            }
        }
        hipMemcpy(d, h.data(), h.size() * sizeof(_Float16), hipMemcpyHostToDevice);
        const int iters = 60000;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        auto launch = [&](int n) {
            if (m.rot == 2) mfma_loop<2><<<cus, 512>>>(d, n, sink);
            else if (m.rot == 1) mfma_loop<1><<<cus, 512>>>(d, n, sink);
            else mfma_loop<0><<<cus, 512>>>(d, n, sink);
        };
        launch(4000);   // warm-up
        hipDeviceSynchronize();
        hipEventRecord(e0);
        launch(iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)cus * 8 * iters * 12 * (2.0 * 32 * 32 * 16);
        printf("%-58s %8.1f ms  %7.1f TFLOP/s  = %.3f of 2500\n", m.name, ms, flops / ms / 1e9, flops / ms / 1e9 / 2500.0);
        fflush(stdout);
    }
    return 0;
}
