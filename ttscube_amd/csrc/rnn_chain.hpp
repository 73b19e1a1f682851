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
    // The lane's row index is made opaque at every call: the persistent kernels call this inside their time-step loop with the same
    // weights every step, so all UN x NG load addresses are loop-invariant — hoisted out of the step loop they are 64-bit per-lane
    // values that do not fit the register file (round-3 review: up to 225 VGPRs in scratch, each reload a dependent
    // scratch_load -> global_load pair inside the step).  Recomputing them per call is two VALU instructions per load.
    asm volatile("" : "+v"(row));
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

// The same chain with the weight address split into a WAVE-UNIFORM base (a kernel-argument pointer) and ONE 32-bit lane offset (in
// float4 units: the lane's row plus whatever k-slice offset it owns).  Every load is then `global_load_dwordx4 v, v_off, s[base]` with
// the k-block / gate part of the address folded into the scalar base: one address register for the whole chain instead of UN x NG
// running 64-bit per-lane pointers (which is what the persistent kernels spilled: see lstm_chain).
template <int BT, int NG, int UN>
__device__ __forceinline__ void lstm_chain_u(float (&acc)[BT][NG], const float* __restrict__ wbase, unsigned lane_off4, int rows, int gstride,
                                               const float* v, int vstride, int K) {
    asm volatile("" : "+v"(lane_off4));
    const int KB = K >> 2;
    const float4* wu = reinterpret_cast<const float4*>(wbase);
    auto load = [&](float4 (&w)[UN][NG], int kb0) {
#pragma unroll
        for (int q = 0; q < UN; ++q)
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const float4* ub = wu + ((size_t)(kb0 + q) * rows + (size_t)g * gstride);   // uniform
                w[q][g] = ub[lane_off4];
            }
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
    float4 wa[UN][NG], wb[UN][NG];
    const int NB = KB / UN;   // caller guarantees KB % UN == 0
    load(wa, 0);
    for (int bi = 0; bi < NB; bi += 2) {
        if (bi + 1 < NB) load(wb, (bi + 1) * UN);
        fma_batch(wa, bi * UN);
        if (bi + 2 < NB) load(wa, (bi + 2) * UN);
        if (bi + 1 < NB) fma_batch(wb, (bi + 1) * UN);
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
