// Is v_mfma_f32_4x4x1_16B_f32 (and 32x32x2) bitwise the fmaf chain?  Build: hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_exact tools/probes/mfma_f32_exact.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// A[64 rows][K], B[K][4 cols] -> D[64][4]; lane l supplies A[row l][k] and B[k][l & 3]; lane l gets D[4*(l>>2)+i][l&3] in reg i
__global__ void k4(const float* A, const float* Bm, const float* C0, float* D, int K) {
    const int l = threadIdx.x;
    f32x4 acc;
    for (int i = 0; i < 4; ++i) acc[i] = C0[(4 * (l >> 2) + i) * 4 + (l & 3)];
    for (int k = 0; k < K; ++k) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(A[l * K + k], Bm[k * 4 + (l & 3)], acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) D[(4 * (l >> 2) + i) * 4 + (l & 3)] = acc[i];
}
// 32x32x2: A[32][K], B[K][32]; lane l: a = A[l&31][2*kk + (l>>5)], b = B[2*kk + (l>>5)][l&31]
__global__ void k32(const float* A, const float* Bm, const float* C0, float* D, int K) {
    const int l = threadIdx.x;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = C0[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)];
    for (int kk = 0; kk < K / 2; ++kk)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(l & 31) * K + 2 * kk + (l >> 5)], Bm[(2 * kk + (l >> 5)) * 32 + (l & 31)], acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}
static float rnd(int mode) {
    float u = (float)rand() / RAND_MAX * 2.f - 1.f;
    if (mode == 1) return u * 1e-20f;            // products underflow into denormals
    if (mode == 2) return ldexpf(u, (rand() % 60) - 30);
    if (mode == 3) return u * 1e-38f * 4.f;       // denormal inputs
    return u;
}
int main() {
    int bad4 = 0, bad32 = 0;
    for (int mode = 0; mode < 4; ++mode) {
        const int K = 512;
        std::vector<float> A(64 * K), B4(K * 4), C4(64 * 4), D4(64 * 4), B32(K * 32), C32(32 * 32), D32(32 * 32);
        for (auto& v : A) v = rnd(mode);
        for (auto& v : B4) v = rnd(mode);
        for (auto& v : C4) v = rnd(mode);
        for (auto& v : B32) v = rnd(mode);
        for (auto& v : C32) v = rnd(mode);
        float *dA, *dB4, *dC4, *dD4, *dB32, *dC32, *dD32;
        hipMalloc(&dA, A.size() * 4); hipMalloc(&dB4, B4.size() * 4); hipMalloc(&dC4, C4.size() * 4); hipMalloc(&dD4, D4.size() * 4);
        hipMalloc(&dB32, B32.size() * 4); hipMalloc(&dC32, C32.size() * 4); hipMalloc(&dD32, D32.size() * 4);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB4, B4.data(), B4.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dC4, C4.data(), C4.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB32, B32.data(), B32.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dC32, C32.data(), C32.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k4, dim3(1), dim3(64), 0, 0, dA, dB4, dC4, dD4, K);
        hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, dA, dB32, dC32, dD32, K);
        hipMemcpy(D4.data(), dD4, D4.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(D32.data(), dD32, D32.size() * 4, hipMemcpyDeviceToHost);
        int b4 = 0, b32 = 0;
        for (int r = 0; r < 64; ++r) for (int c = 0; c < 4; ++c) {
            float x = C4[r * 4 + c];
            for (int k = 0; k < K; ++k) x = fmaf(A[r * K + k], B4[k * 4 + c], x);
            if (memcmp(&x, &D4[r * 4 + c], 4)) ++b4;
        }
        for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c) {
            float x = C32[r * 32 + c];
            for (int k = 0; k < K; ++k) x = fmaf(A[r * K + k], B32[k * 32 + c], x);
            if (memcmp(&x, &D32[r * 32 + c], 4)) ++b32;
        }
        printf("mode %d: 4x4x1 mismatches %d / 256, 32x32x2 mismatches %d / 1024\n", mode, b4, b32);
        bad4 += b4; bad32 += b32;
    }
    printf("RESULT 4x4x1 %s, 32x32x2 %s\n", bad4 ? "DIFFERS" : "bit-exact", bad32 ? "DIFFERS" : "bit-exact");
    return 0;
}
