// Internal layout of a conv layer handle, shared by conv1d.hip (single-layer launches) and resblock.hip (fused chains).
#pragma once
#include "common.hpp"

struct ConvPhase {
    float* wp_dev = nullptr;
    void* wph_dev = nullptr;  // f16x3 fragments
    int ntaps = 0, tap_base = 0, tap_step = 0;
    int out_stride = 1, out_off = 0;
    int r = 0;  // phase index (transposed)
    std::vector<int> taps;  // kernel taps (or tap indices when vfused) of this phase, in GEMM order
};

struct ttsc_conv1d {
    ttsc_conv1d_cfg cfg;
    int MT = 32, NT = 512, CinP = 0, CoutP = 0;
    std::vector<ConvPhase> phases;
    float* bias_dev = nullptr;
    bool has_weight = false;
    int precision = TTSC_PREC_FP32;
    bool vfused = false;   // ConvTranspose1d with Cout % 32 == 0: all `stride` phases as extra GEMM rows of ONE launch
    int CoutV = 0;         // stride * Cout when vfused
    bool vrow4 = false;    // vfused, f16x3 fragments packed with rows interleaved v = co * 4 + r (kernel_size == stride == 4, padding 0)
    float w_unscale = 1.f;
    float act_scale = 1.f;   // f16x3: power-of-two pre-scale of this layer's INPUT (see ttsc_conv1d_set_activation_scale)
    std::vector<float> w_host, b_host;  // kept so that the precision can be switched (repack) at any time
    bool has_bias = false;
    bool dev_weights = false;  // weights were last written by ttsc_conv1d_set_weight_device (host copy is stale)
    const float* bias_ext = nullptr;  // device-weight mode: the caller's bias tensor
    float* w_plain_dev = nullptr;     // out_channels == 1: weights in torch layout [1][Cin][K] for conv_cout1_kernel
    const float* w_plain_ext = nullptr;   // device-weight mode: the caller's weight tensor (same layout)
    // grouped Conv1d (torch `groups`, fp32 path only): channels per group, and the input channels ONE M tile stages — the groups its
    // MT rows belong to (a 32-row tile over 16-row groups stages two groups; the packed weights are block-diagonal inside the tile)
    int groups = 1, cin_g = 0, cout_g = 0, cin_tile = 0;
    unsigned* nf_flag = nullptr;          // out_channels == 1: device word that conv_cout1_kernel ORs with 1 when it emits a non-finite sample
};


#ifdef __HIPCC__
namespace ttsc {
// (hi, lo) fp16 halves of two fp32 values, packed: hi = rn16(v), lo = rn16(v - hi).  v - hi as v_fma_mix_f32 (an f16 operand read as f32
// inside the fma: exact, one rounding — the same bits as convert-back + subtract) saves the two back-conversions per pair.
__device__ __forceinline__ void split2_f16(float v0, float v1, unsigned& hi, unsigned& lo) {
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    typedef float f2_t __attribute__((ext_vector_type(2)));
    const f2_t p = {v0, v1};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(p, h2_t));
    float l0, l1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(hi), "v"(v0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(hi), "v"(v1));
    const f2_t l = {l0, l1};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(l, h2_t));
}
}  // namespace ttsc
#endif
