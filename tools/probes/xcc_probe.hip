// Which XCD does workgroup i of a 1-D / 2-D launch run on?  (HW_REG_XCC_ID; the hand-off kernels place cooperating workgroups on one XCD.)
//   hipcc --offload-arch=gfx950 -O2 tools/probes/xcc_probe.hip -o tools/probes/xcc_probe && tools/probes/xcc_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned* out) {
    const int L = blockIdx.x + gridDim.x * blockIdx.y;
    if (threadIdx.x == 0) out[L] = __builtin_amdgcn_s_getreg(((16 - 1) << 11) | (0 << 6) | 20);
}
int main() {
    unsigned* d;
    hipMalloc(&d, 4096 * 4);
    for (int pass = 0; pass < 2; ++pass) {
        dim3 grid = pass == 0 ? dim3(64, 1) : dim3(16, 16);
        hipMemset(d, 0xff, 4096 * 4);
        hipLaunchKernelGGL(probe, grid, dim3(512), 0, 0, d);
        unsigned h[4096];
        hipMemcpy(h, d, 4096 * 4, hipMemcpyDeviceToHost);
        const int n = grid.x * grid.y;
        printf("grid %u x %u: raw XCC_ID register per linear workgroup index\n", grid.x, grid.y);
        for (int i = 0; i < (n < 64 ? n : 64); ++i) printf("%x%s", h[i], (i % 16 == 15) ? "\n" : " ");
    }
    return 0;
}
