#include <hip/hip_runtime.h>
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2(float v0, float v1, unsigned& hi, unsigned& lo) {
    const float2v p = {v0, v1};
    const half2v h = __builtin_convertvector(p, half2v);
    hi = __builtin_bit_cast(unsigned, h);
    float l0, l1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(hi), "v"(v0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(hi), "v"(v1));
    const float2v l = {l0, l1};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(l, half2v));
}
__global__ void k(const float* x, unsigned* hi, unsigned* lo) {
    int i = threadIdx.x;
    split2(x[2 * i], x[2 * i + 1], hi[i], lo[i]);
}
int main() {
    const int N = 256;
    float hx[2 * N];
    for (int i = 0; i < 2 * N; ++i) hx[i] = (float)((i * 2654435761u) % 100003) * 1e-3f - 50.f + (i % 7) * 1e-6f;
    float* dx; unsigned *dh, *dl;
    hipMalloc(&dx, sizeof(hx)); hipMalloc(&dh, N * 4); hipMalloc(&dl, N * 4);
    hipMemcpy(dx, hx, sizeof(hx), hipMemcpyHostToDevice);
    k<<<1, N>>>(dx, dh, dl);
    unsigned hh[N], hl[N];
    hipMemcpy(hh, dh, N * 4, hipMemcpyDeviceToHost); hipMemcpy(hl, dl, N * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < N; ++i) for (int e = 0; e < 2; ++e) {
        float v = hx[2 * i + e];
        _Float16 h = (_Float16)v; _Float16 l = (_Float16)(v - (float)h);
        unsigned short eh = __builtin_bit_cast(unsigned short, h), el = __builtin_bit_cast(unsigned short, l);
        unsigned short gh = (hh[i] >> (16 * e)) & 0xffff, gl = (hl[i] >> (16 * e)) & 0xffff;
        if (eh != gh || el != gl) ++bad;
    }
    printf("split2 mismatches: %d\n", bad);
    return bad != 0;
}
