// Hand-off latency between two workgroups, by memory scope and by XCD placement (gfx950).
//   hipcc --offload-arch=gfx950 -O2 -o tools/probes/handoff.bin tools/probes/handoff_probe.hip
// Pairs (A = block b, B = block b + stride): stride 8 -> same XCD if workgroups are dealt round-robin, stride 1 -> neighbours.
// A writes a 256-float payload + bumps flagA; B polls flagA, checks the payload, bumps flagB; A polls flagB.  Bounded spins.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int SCOPE> __device__ __forceinline__ void st(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, SCOPE); }
template <int SCOPE> __device__ __forceinline__ float ld(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, SCOPE); }
template <int SCOPE> __device__ __forceinline__ unsigned ldu(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, SCOPE); }
template <int SCOPE> __device__ __forceinline__ void stu(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, SCOPE); }

struct Args { float* data; unsigned* flags; unsigned* xcc; long long* cycles; unsigned* errors; int iters, stride, npairs; };

template <int SCOPE>
__global__ __launch_bounds__(256) void pingpong(Args a) {
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) a.xcc[b] = __builtin_amdgcn_s_getreg((3 << 11) | 20);   // HW_REG_XCC_ID[3:0]
    // pair index / role
    int pair = -1, role = 0;
    for (int p = 0; p < a.npairs; ++p) {
        const int A = (p / a.stride) * 2 * a.stride + (p % a.stride);
        if (b == A) { pair = p; role = 0; }
        if (b == A + a.stride) { pair = p; role = 1; }
    }
    if (pair < 0) return;
    float* data = a.data + (size_t)pair * 2 * 256;
    unsigned* fA = a.flags + pair * 64;
    unsigned* fB = a.flags + pair * 64 + 32;
    __shared__ int ok_s;
    unsigned err = 0;
    long long t0 = 0;
    for (int i = 1; i <= a.iters; ++i) {
        if (i == 11 && tid == 0) t0 = wall_clock64();
        if (role == 0) {
            st<SCOPE>(data + (i & 1) * 256 + tid, (float)i);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                stu<SCOPE>(fA, (unsigned)i);
                unsigned spins = 0;
                int ok = 1;
                while (ldu<SCOPE>(fB) < (unsigned)i) if (++spins > (1u << 20)) { ok = 0; break; }
                ok_s = ok;
            }
            __syncthreads();
            if (!ok_s) { err |= 2; break; }
        } else {
            if (tid == 0) {
                unsigned spins = 0;
                int ok = 1;
                while (ldu<SCOPE>(fA) < (unsigned)i) if (++spins > (1u << 20)) { ok = 0; break; }
                ok_s = ok;
            }
            __syncthreads();
            if (!ok_s) { err |= 2; break; }
            const float v = ld<SCOPE>(data + (i & 1) * 256 + tid);
            if (v != (float)i) err |= 1;
            __syncthreads();
            if (tid == 0) stu<SCOPE>(fB, (unsigned)i);
        }
    }
    if (tid == 0 && role == 0) a.cycles[pair] = wall_clock64() - t0;
    if (err) atomicOr(a.errors + pair, err);
}

template <int SCOPE>
static void run(const char* name, int stride) {
    const int npairs = 8, iters = 2010, nblk = 32;
    Args a;
    hipMalloc(&a.data, npairs * 2 * 256 * 4); hipMalloc(&a.flags, npairs * 64 * 4); hipMalloc(&a.xcc, nblk * 4);
    hipMalloc(&a.cycles, npairs * 8); hipMalloc(&a.errors, npairs * 4);
    hipMemset(a.data, 0, npairs * 2 * 256 * 4); hipMemset(a.flags, 0, npairs * 64 * 4); hipMemset(a.errors, 0, npairs * 4);
    hipMemset(a.cycles, 0, npairs * 8);
    a.iters = iters; a.stride = stride; a.npairs = npairs;
    hipLaunchKernelGGL(pingpong<SCOPE>, dim3(nblk), dim3(256), 0, 0, a);
    hipDeviceSynchronize();
    std::vector<long long> cyc(npairs); std::vector<unsigned> err(npairs), xcc(nblk);
    hipMemcpy(cyc.data(), a.cycles, npairs * 8, hipMemcpyDeviceToHost); hipMemcpy(err.data(), a.errors, npairs * 4, hipMemcpyDeviceToHost);
    hipMemcpy(xcc.data(), a.xcc, nblk * 4, hipMemcpyDeviceToHost);
    printf("%-10s stride %d: ", name, stride);
    for (int p = 0; p < npairs; ++p) printf("%.2fus%s ", cyc[p] / 100.0 / 2000.0, err[p] ? (err[p] & 2 ? "(TIMEOUT)" : "(STALE)") : "");   // wall_clock64: 100 MHz
    printf("| xcc of blocks 0..15:");
    for (int b = 0; b < 16; ++b) printf(" %u", xcc[b]);
    printf("\n");
}

int main() {
    run<__HIP_MEMORY_SCOPE_AGENT>("agent", 8);
    run<__HIP_MEMORY_SCOPE_AGENT>("agent", 1);
    run<__HIP_MEMORY_SCOPE_WORKGROUP>("workgroup", 8);
    run<__HIP_MEMORY_SCOPE_WORKGROUP>("workgroup", 1);
    run<__HIP_MEMORY_SCOPE_SYSTEM>("system", 8);
    return 0;
}
