// Shared matvec chain of the persistent recurrent kernels (lstm.hip, gru.hip): thread `row` accumulates NG gate rows over K
// inputs from a weight matrix packed [K/4][rows][4] (one 16-byte load per row per 4 inputs), for BT input vectors in LDS.
#pragma once
#include <hip/hip_runtime.h>

namespace ttsc {

template <int BT, int NG, int UN>
__device__ __forceinline__ void lstm_chain(float (&acc)[BT][NG], const float* __restrict__ wp, int rows, int gstride, int row,
                                             const float* v, int vstride, int K) {
    // Weight stream with EXPLICIT software pipelining: the 16-byte loads of a whole batch (UN k-blocks x NG rows) are
    // issued back to back into one register set while the fmaf chain consumes the other set.  Left to itself hipcc
    // places each load right before its use and waits vmcnt(0) per load, i.e. one L2 round trip per 16 bytes.
    // The chain order (k ascending, one fmaf per term) is unchanged.
    const float4* w4 = reinterpret_cast<const float4*>(wp) + row;
    const int KB = K >> 2;
    auto load = [&](float4 (&w)[UN][NG], int kb0) {
#pragma unroll
        for (int q = 0; q < UN; ++q)
#pragma unroll
            for (int g = 0; g < NG; ++g) w[q][g] = w4[(size_t)(kb0 + q) * rows + g * gstride];
    };
    auto fma_batch = [&](const float4 (&w)[UN][NG], int kb0) {
#pragma unroll
        for (int q = 0; q < UN; ++q) {
#pragma unroll
            for (int u = 0; u < BT; ++u) {
                const float4 hv = *reinterpret_cast<const float4*>(v + u * vstride + 4 * (kb0 + q));
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    float x = acc[u][g];
                    x = fmaf(w[q][g].x, hv.x, x);
                    x = fmaf(w[q][g].y, hv.y, x);
                    x = fmaf(w[q][g].z, hv.z, x);
                    x = fmaf(w[q][g].w, hv.w, x);
                    acc[u][g] = x;
                }
            }
        }
    };
    if (KB % UN == 0) {
        float4 wa[UN][NG], wb[UN][NG];
        const int NB = KB / UN;
        load(wa, 0);
        for (int bi = 0; bi < NB; bi += 2) {
            if (bi + 1 < NB) load(wb, (bi + 1) * UN);
            fma_batch(wa, bi * UN);
            if (bi + 2 < NB) load(wa, (bi + 2) * UN);
            if (bi + 1 < NB) fma_batch(wb, (bi + 1) * UN);
        }
    } else {
        for (int kb = 0; kb < KB; ++kb) {
            float4 w[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g) w[g] = w4[(size_t)kb * rows + g * gstride];
#pragma unroll
            for (int u = 0; u < BT; ++u) {
                const float4 hv = *reinterpret_cast<const float4*>(v + u * vstride + 4 * kb);
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    float x = acc[u][g];
                    x = fmaf(w[g].x, hv.x, x);
                    x = fmaf(w[g].y, hv.y, x);
                    x = fmaf(w[g].z, hv.z, x);
                    x = fmaf(w[g].w, hv.w, x);
                    acc[u][g] = x;
                }
            }
        }
    }
}

// ---- inter-workgroup hand-off used by the split recurrences (gru.hip, lstm.hip); counter protocol ----------
// payload: agent-scope relaxed atomic stores / loads (write-through, L1-bypassing); arrival: every storing wave drains
// vmcnt(0), one lane bumps a monotonic counter; consumers poll it from one lane with a bounded spin and a shared abort word.
__device__ __forceinline__ void g_st(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float g_ld(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
constexpr unsigned GS_SPIN_LIMIT = 1u << 22;

__device__ __forceinline__ bool g_wait(unsigned* cnt, unsigned want, unsigned* abort_word, int* ok_s) {
    if (threadIdx.x == 0) {
        int ok = 1;
        unsigned spins = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
            if (++spins > GS_SPIN_LIMIT || __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(abort_word + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sticky copy (common.hpp HandoffArea)
                ok = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        *ok_s = ok;
    }
    __syncthreads();
    return *ok_s != 0;
}
__device__ __forceinline__ void g_publish(unsigned* cnt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}


}  // namespace ttsc
