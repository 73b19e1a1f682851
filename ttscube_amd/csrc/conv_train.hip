// Split-precision convolutions for the TRAINING step: forward and data gradient of every dense Conv1d of the
// generator and of the MPD / MSD discriminators on v_mfma_f32_32x32x16_f16 (fp32 carried as fp16 hi + lo, three products, fp32
// accumulation: 22 significant bits, see conv_kernels.hpp) instead of the exact-fp32 MFMA kernel (157 TF/s peak, 45-90 TF/s measured on
// these shapes, profiles/r03_disc_layers_fp32.log).  What inference does with calibrated, sticky per-layer scales cannot work here:
// activations AND gradients move every step and span many orders of magnitude between layers.  So
//   * the range is measured on the device for every launch group: one reduction launch writes max |x| and max |w| (amax2_kernel), the weight
//     packing and the convolution derive their power-of-two scales from those words in-kernel — no host round trip, no state;
//   * the weights are split and laid out in fragment order from the live torch parameter by pack_wh_kernel (also for the data gradient:
//     roles of the channel dimensions swapped, taps reversed);
//   * the batch is folded into the GEMM's column dimension (ConvArgs::fold_S): the deep discriminator layers are 1024 x 1024 x 5 weights
//     over 100 .. 1 200 positions per sequence; one tile per (sequence, 64 rows) re-streams 1.3 MB of weights for a sliver of columns.
// The weight gradient has its own split-precision kernel (conv_wgrad.hip::wgrad_f16x3_kernel) that shares these range words.
#include <algorithm>
#include <cstdlib>

#include "common.hpp"
#include "conv_kernels.hpp"

namespace ttsc {

// *out_x = max |x|, *out_w = max |w| (the words are zeroed first; non-negative floats order like their bit patterns).  Blocks [0, bx) reduce x,
// the rest w; either tensor may be absent (bx == 0 / bx == gridDim.x).
__global__ __launch_bounds__(256) void amax2_kernel(const float* __restrict__ x, long nx, int bx, unsigned* __restrict__ out_x, const float* __restrict__ w,
                                                    long nw, unsigned* __restrict__ out_w) {
    const bool is_w = (int)blockIdx.x >= bx;
    const float* src = is_w ? w : x;
    const long n = is_w ? nw : nx;
    const long first = (long)(is_w ? blockIdx.x - bx : blockIdx.x) * 256 + threadIdx.x;
    const long stride = (long)(is_w ? gridDim.x - bx : bx) * 256;
    float m = 0.f;
    const long n4 = (((uintptr_t)src & 15) == 0) ? n / 4 : 0;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    for (long i = first; i < n4; i += stride) {
        const f32x4 v = reinterpret_cast<const f32x4*>(src)[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    for (long i = n4 * 4 + first; i < n; i += stride) m = fmaxf(m, fabsf(src[i]));
    // (fmaxf drops NaN elements: they reach the output through the data path itself; an inf makes the word inf -> scale 1, visible likewise)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    // one atomic per workgroup: thousands of same-address atomics serialise in L2 (the first version, one per wave, took 59 us flat)
    __shared__ float wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
        if (m > 0.f) atomicMax(is_w ? out_w : out_x, __float_as_uint(m));
    }
}

struct PackHArgs {
    const float* w;      // flip == 0: [Cout][Cin][K];  flip == 1: the forward weight [Cin][Cout][K] of the layer being differentiated
    _Float16* out;       // [K][CinP/16][CoutP/32][2 (hi, lo)][64 lanes][8 half]
    const float* amax_w;   // *amax_w = max |w|
    int Cin, Cout, K, nch, cotN, flip;
    int groups, cin_g, cout_g, cin_tile, MT;   // grouped layer (torch `groups`): a row tile of MT rows meets the cin_tile input channels of its group(s)
};
__global__ __launch_bounds__(256) void pack_wh_kernel(PackHArgs p) {
    const float scale = pow2_to(*p.amax_w, SPLIT_W_TARGET);
    const long total = (long)p.K * p.nch * p.cotN * 64;   // one thread per (tap, chunk, row tile, lane): 8 channels
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int lane = (int)(i & 63);
        long t = i >> 6;
        const int cot = (int)(t % p.cotN);
        t /= p.cotN;
        const int ch = (int)(t % p.nch);
        const int j = (int)(t / p.nch);
        const int co = cot * 32 + (lane & 31);
        half8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ci = ch * 16 + 8 * (lane >> 5) + e;
            float v = 0.f;
            if (p.groups > 1) {
                // ci counts from the first input channel of the row tile's group(s) (block-diagonal when a tile holds several groups).  Forward
                // weight [Cout][cin_g][K]; as a data gradient the source is the differentiated layer's weight [Cin(this)][cout_g(this)][K]
                const int gco = co / p.cout_g, cig = ((co / p.MT) * p.MT / p.cout_g) * p.cin_g + ci;
                if (co < p.Cout && ci < p.cin_tile && cig / p.cin_g == gco)
                    v = p.flip ? p.w[((size_t)cig * p.cout_g + (co - gco * p.cout_g)) * p.K + (p.K - 1 - j)]
                               : p.w[((size_t)co * p.cin_g + (cig - gco * p.cin_g)) * p.K + j];
            } else if (co < p.Cout && ci < p.Cin)
                v = p.flip ? p.w[((size_t)ci * p.Cout + co) * p.K + (p.K - 1 - j)] : p.w[((size_t)co * p.Cin + ci) * p.K + j];
            v *= scale;
            const _Float16 h = (_Float16)v;
            hi[e] = h;
            lo[e] = (_Float16)(v - (float)h);
        }
        half8* dst = reinterpret_cast<half8*>(p.out) + ((((size_t)j * p.nch + ch) * p.cotN + cot) * 2) * 64;
        dst[lane] = hi;
        dst[64 + lane] = lo;
    }
}

// zero the range words to be measured, then one launch over the tensors (also used by the split-precision weight gradient, conv_wgrad.hip)
int launch_amax2(const float* x, long nx, float* out_x, const float* w, long nw, float* out_w, hipStream_t s) {
    if (!x && !w) return TTSC_OK;
    hipError_t e = hipSuccess;
    if (x && w && out_w == out_x + 1)
        e = hipMemsetAsync(out_x, 0, 8, s);
    else {
        if (x) e = hipMemsetAsync(out_x, 0, 4, s);
        if (w && e == hipSuccess) e = hipMemsetAsync(out_w, 0, 4, s);
    }
    if (e != hipSuccess) {
        set_error("hipMemsetAsync: %s", hipGetErrorString(e));
        return TTSC_EHIP;
    }
    const int bx = x ? (int)std::min<long>((nx / 16 + 255) / 256 + 1, 256) : 0, bw = w ? (int)std::min<long>((nw / 16 + 255) / 256 + 1, 128) : 0;
    hipLaunchKernelGGL(amax2_kernel, dim3(bx + bw), dim3(256), 0, s, x, nx, bx, reinterpret_cast<unsigned*>(out_x), w, nw, reinterpret_cast<unsigned*>(out_w));
    return TTSC_OK;
}

template <int MI, int NJ, int TMAX>
static int launch_fold_t(const ConvArgs& a, hipStream_t s) {
    constexpr int NT = 4 * NJ * 32;
    dim3 grid((unsigned)ceil_div((long)a.fold_S * a.fold_B, NT), (unsigned)(a.CoutP / (32 * MI)), 1u);
    const size_t lds = (size_t)a.span_pad * 4 * 16 + (size_t)a.ntaps * MI * 2 * 64 * 16;
    TTSC_REQUIRE(lds <= 160 * 1024, "ttsc_conv_train: LDS footprint %zu", lds);
    if (int rc = ensure_full_lds((const void*)conv_f16x3_kernel<MI, NJ, TMAX, true>)) return rc;
    hipLaunchKernelGGL((conv_f16x3_kernel<MI, NJ, TMAX, true>), grid, dim3(256), lds, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("conv_f16x3_kernel (folded) launch failed: %s", hipGetErrorString(e));
        return TTSC_EHIP;
    }
    return TTSC_OK;
}
template <int MI, int NJ>
static int launch_fold(const ConvArgs& a, hipStream_t s) {
    if (a.ntaps <= 3) return launch_fold_t<MI, NJ, 3>(a, s);
    if (a.ntaps <= 7) return launch_fold_t<MI, NJ, 7>(a, s);
    if (a.ntaps <= 11) return launch_fold_t<MI, NJ, 11>(a, s);
    if (a.ntaps <= 16) return launch_fold_t<MI, NJ, 16>(a, s);
    return launch_fold_t<MI, NJ, 21>(a, s);   // MSD's k = 41, stride 2 layers: 21 taps after the de-interleave
}

}  // namespace ttsc

using namespace ttsc;

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// row tile of a (grouped) layer: 64 rows when they stay inside one group or hold whole groups, else 32; 0 = does not tile
static int train_mt(int Cout, int groups, int K) {
    const int cout_g = Cout / groups;
    if (K > 21)                                  // k = 41 at stride 1: all taps of a chunk must fit the LDS beside the window -> 32-row tiles;
        return (groups > 1 && !(cout_g % 32 == 0 || 32 % cout_g == 0)) ? 0 : 32;   // a tile must not straddle groups unevenly
    int mt = Cout >= 64 ? 64 : 32;
    if (groups > 1) {
        mt = cout_g % 64 == 0 ? 64 : 32;
        if (!(cout_g % mt == 0 || mt % cout_g == 0)) return 0;
    }
    return mt;
}

extern "C" int32_t ttsc_conv_train_supported(int32_t Cin, int32_t Cout, int32_t K, int32_t dilation, int32_t groups) {
    // (thin layers — the discriminators' first and last convolutions, 1 input or 1 output channel — are mostly channel padding on either kernel:
    // one 16-channel chunk / one 32-row tile; they are taken too, the launches around them cost more than their MFMAs)
    if (groups < 1 || Cin < 1 || Cout < 1 || Cin % groups || Cout % groups) return 0;
    if (groups > 1 && Cin / groups < 8) return 0;
    return K >= 1 && K <= 41 && dilation >= 1 && (K - 1) * dilation <= 64 && train_mt(Cout, groups, K) != 0;
}

extern "C" size_t ttsc_conv_train_workspace_bytes(int32_t Cin, int32_t Cout, int32_t K, int32_t groups) {
    if (groups < 1) groups = 1;
    const int mt = std::max(train_mt(Cout, groups, K), 32), cout_g = Cout / groups, cin_g = Cin / groups;
    const int cin_tile = groups > 1 ? (mt > cout_g ? mt / cout_g : 1) * cin_g : Cin;
    return 256 + (size_t)K * round_up(cin_tile, 16) * round_up(Cout, mt) * 2 * sizeof(_Float16);
}

extern "C" int ttsc_conv_train(const float* x, const float* w, const float* bias, const float* resid, const float* gate, float* y, int32_t B,
                               int32_t Cin, int32_t Cout, int32_t K, int64_t Lin, int32_t padding, int32_t dilation, int32_t groups, int32_t flip,
                               float in_scale, float in_slope, float out_scale, float gate_slope, float* amax_x, float* amax_w, int32_t measure, void* ws,
                               size_t ws_bytes, void* stream) {
    TTSC_REQUIRE(x && w && y && ws, "ttsc_conv_train: null argument");
    TTSC_REQUIRE(ttsc_conv_train_supported(Cin, Cout, K, dilation, groups), "ttsc_conv_train: shape not supported (Cin %d, Cout %d, K %d, dilation %d, groups %d)",
                 Cin, Cout, K, dilation, groups);
    TTSC_REQUIRE(B > 0 && Lin > 0 && padding >= 0, "ttsc_conv_train: bad B / Lin / padding");
    TTSC_REQUIRE(in_slope >= 0.f && in_slope <= 1.f && in_scale > 0.f, "ttsc_conv_train: in_slope must be in [0, 1], in_scale positive");
    const int64_t Lout = Lin + 2 * (int64_t)padding - (int64_t)dilation * (K - 1);
    TTSC_REQUIRE(Lout > 0, "ttsc_conv_train: output length %lld <= 0", (long long)Lout);
    const int64_t S = std::max<int64_t>(Lin + padding, Lout);
    TTSC_REQUIRE((int64_t)B * Cin * Lin < (1ll << 31) && (int64_t)B * Cout * Lout < (1ll << 31) && S * B < (1ll << 30), "ttsc_conv_train: tensor too large");
    TTSC_REQUIRE(ws_bytes >= ttsc_conv_train_workspace_bytes(Cin, Cout, K, groups), "ttsc_conv_train: workspace too small");
    TTSC_REQUIRE(((uintptr_t)ws & 15) == 0, "ttsc_conv_train: workspace must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    // range words: the caller's (shared between the launches of one layer: `measure` bit 0 = write max |x| now, bit 1 = write max |w| now; a clear
    // bit means an earlier launch of the layer measured that tensor) or, with null pointers, two words of the workspace measured here
    float* ws_words = reinterpret_cast<float*>(ws);
    if (!amax_x) { amax_x = ws_words; measure |= 1; }
    if (!amax_w) { amax_w = ws_words + 1; measure |= 2; }
    _Float16* wph = reinterpret_cast<_Float16*>(reinterpret_cast<char*>(ws) + 256);
    const int MT = train_mt(Cout, groups, K), MI = MT / 32;
    const int cin_g = Cin / groups, cout_g = Cout / groups;
    const int cin_tile = groups > 1 ? (MT > cout_g ? MT / cout_g : 1) * cin_g : Cin;
    const int CinP = round_up(cin_tile, 16), CoutP = round_up(Cout, MT);

    if (int rc = launch_amax2((measure & 1) ? x : nullptr, (long)B * Cin * Lin, amax_x, (measure & 2) ? w : nullptr, (long)cin_g * Cout * K, amax_w, s)) return rc;
    {
        PackHArgs p;
        p.w = w;
        p.out = wph;
        p.amax_w = amax_w;
        p.Cin = Cin;
        p.Cout = Cout;
        p.K = K;
        p.nch = CinP / 16;
        p.cotN = CoutP / 32;
        p.flip = flip;
        p.groups = groups;
        p.cin_g = cin_g;
        p.cout_g = cout_g;
        p.cin_tile = cin_tile;
        p.MT = MT;
        const long total = (long)K * p.nch * p.cotN * 64;
        hipLaunchKernelGGL(pack_wh_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0, s, p);
    }
    ConvArgs a{};
    a.x = x;
    a.y = y;
    a.resid = resid;
    a.wph = wph;
    a.w_unscale = 1.f;
    a.bias = bias;
    a.Cin = cin_tile;
    a.CinTot = Cin;
    a.CinP = CinP;
    a.Cout = Cout;
    a.CoutP = CoutP;
    a.groups = groups;
    a.cin_g = cin_g;
    a.cout_g = cout_g;
    a.Lin = (int)Lin;
    a.Lout = (int)Lout;
    a.ntaps = K;
    a.tap_base = -padding;
    a.tap_step = dilation;
    a.out_stride = 1;
    a.min_shift = -padding;
    a.in_scale = in_scale;
    a.in_slope = in_slope;
    a.out_scale = out_scale;
    a.out_act = TTSC_ACT_NONE;
    a.gate = gate;
    a.gate_slope = gate_slope;
    a.fold_S = (int)S;
    a.fold_B = B;
    a.amax_x = amax_x;
    a.amax_w = amax_w;
    a.q_cnt = (int)(S * B);
    // column tile: 256 wide when that still gives every CU two workgroups, else 128 (and 128 when 21+ taps of weights share the LDS)
    const long cols = S * B;
    const long wg256 = ceil_div(cols, 256) * (CoutP / MT);
    static const int nt_env = getenv("TTSC_TRAIN_NT") ? atoi(getenv("TTSC_TRAIN_NT")) : 0;
    const int nt = K > 16 ? 128 : (nt_env ? nt_env : (wg256 >= 512 ? 256 : 128));
    a.span = nt + (K - 1) * dilation;
    a.span_pad = a.span;
    int rc;
    if (K > 21)
        rc = launch_fold_t<1, 1, 41>(a, s);
    else if (MI == 2)
        rc = nt == 256 ? launch_fold<2, 2>(a, s) : launch_fold<2, 1>(a, s);
    else
        rc = nt == 256 ? launch_fold<1, 2>(a, s) : launch_fold<1, 1>(a, s);
    if (rc) return rc;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("ttsc_conv_train launch failed: %s", hipGetErrorString(e));
        return TTSC_EHIP;
    }
    return TTSC_OK;
}
