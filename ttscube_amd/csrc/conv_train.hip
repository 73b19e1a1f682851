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
int launch_amax2(const float* x, long nx, float* out_x, const float* w, long nw, float* out_w, hipStream_t s, bool prezeroed) {
    if (!x && !w) return TTSC_OK;
    hipError_t e = hipSuccess;
    if (prezeroed) {
        // (the caller handed out words of a pool it zeroed with one launch for the whole step)
    } else if (x && w && out_w == out_x + 1)
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

// `prepacked` != null: the weight fragments (and *amax_w) come from a weight bank (ttsc_wbank_prepare): no weight reduction, no packing launch,
// no workspace; `measure` bit 2: the caller zeroed *amax_x (a pooled word) — no memset launch either
static int conv_train_impl(const float* x, const float* w, const _Float16* prepacked, const float* bias, const float* resid, const float* gate, float* y,
                           int32_t B, int32_t Cin, int32_t Cout, int32_t K, int64_t Lin, int32_t padding, int32_t dilation, int32_t groups, int32_t flip,
                           float in_scale, float in_slope, float out_scale, float gate_slope, float* amax_x, float* amax_w, int32_t measure, void* ws,
                           size_t ws_bytes, void* stream, float* amax_y = nullptr) {
    TTSC_REQUIRE(x && (w || prepacked) && y && (ws || prepacked), "ttsc_conv_train: null argument");
    TTSC_REQUIRE(ttsc_conv_train_supported(Cin, Cout, K, dilation, groups), "ttsc_conv_train: shape not supported (Cin %d, Cout %d, K %d, dilation %d, groups %d)",
                 Cin, Cout, K, dilation, groups);
    TTSC_REQUIRE(B > 0 && Lin > 0 && padding >= 0, "ttsc_conv_train: bad B / Lin / padding");
    TTSC_REQUIRE(in_slope >= 0.f && in_slope <= 1.f && in_scale > 0.f, "ttsc_conv_train: in_slope must be in [0, 1], in_scale positive");
    const int64_t Lout = Lin + 2 * (int64_t)padding - (int64_t)dilation * (K - 1);
    TTSC_REQUIRE(Lout > 0, "ttsc_conv_train: output length %lld <= 0", (long long)Lout);
    const int64_t S = std::max<int64_t>(Lin + padding, Lout);
    TTSC_REQUIRE((int64_t)B * Cin * Lin < (1ll << 31) && (int64_t)B * Cout * Lout < (1ll << 31) && S * B < (1ll << 30), "ttsc_conv_train: tensor too large");
    TTSC_REQUIRE(prepacked || ws_bytes >= ttsc_conv_train_workspace_bytes(Cin, Cout, K, groups), "ttsc_conv_train: workspace too small");
    TTSC_REQUIRE(prepacked ? (((uintptr_t)prepacked & 15) == 0 && amax_x && amax_w) : (((uintptr_t)ws & 15) == 0),
                 "ttsc_conv_train: workspace / fragments must be 16-byte aligned (and a bank's launches carry both range words)");
    hipStream_t s = (hipStream_t)stream;
    // range words: the caller's (shared between the launches of one layer: `measure` bit 0 = write max |x| now, bit 1 = write max |w| now; a clear
    // bit means an earlier launch of the layer measured that tensor) or, with null pointers, two words of the workspace measured here
    float* ws_words = reinterpret_cast<float*>(ws);
    if (!amax_x) { amax_x = ws_words; measure |= 1; }
    if (!amax_w) { amax_w = ws_words + 1; measure |= 2; }
    if (prepacked) measure &= ~2;
    const _Float16* wph = prepacked ? prepacked : reinterpret_cast<_Float16*>(reinterpret_cast<char*>(ws) + 256);
    const int MT = train_mt(Cout, groups, K), MI = MT / 32;
    const int cin_g = Cin / groups, cout_g = Cout / groups;
    const int cin_tile = groups > 1 ? (MT > cout_g ? MT / cout_g : 1) * cin_g : Cin;
    const int CinP = round_up(cin_tile, 16), CoutP = round_up(Cout, MT);

    if (int rc = launch_amax2((measure & 1) ? x : nullptr, (long)B * Cin * Lin, amax_x, (measure & 2) ? w : nullptr, (long)cin_g * Cout * K, amax_w, s,
                              (measure & 4) != 0))
        return rc;
    if (!prepacked) {
        PackHArgs p;
        p.w = w;
        p.out = const_cast<_Float16*>(wph);
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
    a.amax_out = reinterpret_cast<unsigned*>(amax_y);
#ifdef TTSC_ABLATE
    if (const char* ev = getenv("TTSC_CONV_DBG")) a.dbg = atoi(ev);   // (measurement build only: see TTSC_DBG in conv_kernels.hpp)
#endif
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

extern "C" int ttsc_conv_train(const float* x, const float* w, const float* bias, const float* resid, const float* gate, float* y, int32_t B,
                               int32_t Cin, int32_t Cout, int32_t K, int64_t Lin, int32_t padding, int32_t dilation, int32_t groups, int32_t flip,
                               float in_scale, float in_slope, float out_scale, float gate_slope, float* amax_x, float* amax_w, int32_t measure, void* ws,
                               size_t ws_bytes, void* stream) {
    TTSC_REQUIRE(w && ws, "ttsc_conv_train: null argument");
    return conv_train_impl(x, w, nullptr, bias, resid, gate, y, B, Cin, Cout, K, Lin, padding, dilation, groups, flip, in_scale, in_slope, out_scale,
                           gate_slope, amax_x, amax_w, measure & 7, ws, ws_bytes, stream);   // (bit 2: the caller's words were zeroed by the caller — pooled words)
}

extern "C" int ttsc_conv_train_packed(const float* x, const void* wfrag, const float* bias, const float* resid, const float* gate, float* y, int32_t B,
                                      int32_t Cin, int32_t Cout, int32_t K, int64_t Lin, int32_t padding, int32_t dilation, int32_t groups,
                                      float in_scale, float in_slope, float out_scale, float gate_slope, float* amax_x, const float* amax_w,
                                      int32_t measure, float* amax_y, void* stream) {
    TTSC_REQUIRE(wfrag && amax_x && amax_w, "ttsc_conv_train_packed: null argument");
    return conv_train_impl(x, nullptr, reinterpret_cast<const _Float16*>(wfrag), bias, resid, gate, y, B, Cin, Cout, K, Lin, padding, dilation, groups, 0,
                           in_scale, in_slope, out_scale, gate_slope, amax_x, const_cast<float*>(amax_w), measure & 5, nullptr, 0, stream, amax_y);
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// Weight bank: everything the convolutions of ONE module (the generator; the five period / three scale discriminators) need from their
// parameters for a step, in THREE launches for the whole module instead of four per convolution launch.  Round 5's profile of the Cubegan step
// (profiles/r05_train_last300ms_kernel_stats.csv): 3 670 launches per 80 ms step, of which 388 pack_wh_kernel, 190 wn_fwd_kernel, 120
// deinterleave_w_kernel, and the weight half of 392 amax2_kernel launches — each a few microseconds of work behind a launch.
//   launch 1  zero the range words
//   launch 2  per weight row: w = g * v / ||v|| (weight norm, dim 0; plain weights pass through), the row norms for the backward pass, max |w|
//   launch 3  both operand orders of every layer (forward; data gradient: channel roles swapped, taps reversed) split into fp16 hi / lo and laid
//             out in MFMA fragment order, with the stride de-interleave of the discriminators' strided layers folded into the read
//             (w'[co, (r, ci), j] = w[co, ci, s j + r], zero beyond K) — the kernels of ttsc_conv_train_packed take the fragments as they are.
struct WBankDev {
    const float* v;
    const float* g;
    float* w;
    float* norm;
    float* amax;
    _Float16* pack[2];       // [0] forward, [1] data gradient (null = not wanted)
    int rows, cols;          // weight-norm view: rows = Cout, cols = Cg * K
    int Cg, K, stride, J;    // source layout [Cout][Cg][K]; the convolution sees [Cout][stride * Cg][J]
    // per orientation: the PackHArgs geometry of the stride-1 convolution that consumes the fragments
    int Cin[2], Cout[2], nch[2], cotN[2], groups, cin_g[2], cout_g[2], cin_tile[2], MT[2];
};

namespace ttsc {
__global__ __launch_bounds__(256) void wbank_norm_kernel(const WBankDev* __restrict__ tab) {
    const WBankDev& e = tab[blockIdx.y];
    __shared__ float red[4];
    float amax = 0.f;
    for (int r = blockIdx.x; r < e.rows; r += gridDim.x) {
        const float* vr = e.v + (size_t)r * e.cols;
        float k = 1.f;
        if (e.g) {   // same arithmetic, in the same order, as wn_fwd_kernel (train_ops.hip)
            float s = 0.f;
            for (int c = threadIdx.x; c < e.cols; c += 256) s = fmaf(vr[c], vr[c], s);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
            __syncthreads();
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
            __syncthreads();
            const float n = sqrtf(((0.f + red[0]) + red[1] + red[2]) + red[3]);
            k = e.g[r] / n;
            if (threadIdx.x == 0) e.norm[r] = n;
        }
        for (int c = threadIdx.x; c < e.cols; c += 256) {
            const float wv = e.g ? vr[c] * k : vr[c];
            if (e.g) e.w[(size_t)r * e.cols + c] = wv;
            amax = fmaxf(amax, fabsf(wv));
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
    __syncthreads();
    if (threadIdx.x == 0) {
        amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        if (amax > 0.f) atomicMax(reinterpret_cast<unsigned*>(e.amax), __float_as_uint(amax));
    }
}

// element [a][b][j] of the weight the convolution sees ([Cout][stride * Cg][J]) read from the source layout [Cout][Cg][K]
__device__ __forceinline__ float wbank_src(const WBankDev& e, const float* w, int a, int b, int j) {
    if (e.stride == 1) return w[((size_t)a * e.Cg + b) * e.K + j];
    const int r = b / e.Cg, ci = b - r * e.Cg, k = e.stride * j + r;
    return k < e.K ? w[((size_t)a * e.Cg + ci) * e.K + k] : 0.f;
}

__global__ __launch_bounds__(256) void wbank_pack_kernel(const WBankDev* __restrict__ tab) {
    const WBankDev& e = tab[blockIdx.y];
    const int o = blockIdx.z;                 // 0 forward, 1 data gradient
    if (!e.pack[o]) return;
    const float* w = e.g ? e.w : e.v;
    const float scale = pow2_to(*e.amax, SPLIT_W_TARGET);
    const int J = e.J, nch = e.nch[o], cotN = e.cotN[o], Cin = e.Cin[o], Cout = e.Cout[o];
    const int cgs = e.stride * e.Cg;          // input channels per group of the forward weight the convolution sees
    // one thread per (chunk, row tile, lane) walks the taps: a thread's 8 channels x J taps are 8 short contiguous runs of the source, so the
    // lines it touches for tap j serve taps j + 1 .. from the vector cache (with the taps outermost every line came from L2 once per tap)
    const long total = (long)nch * cotN * 64;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int lane = (int)(i & 63);
        long t = i >> 6;
        const int cot = (int)(t % cotN);
        const int ch = (int)(t / cotN);
        const int co = cot * 32 + (lane & 31);
        for (int j = 0; j < J; ++j) {
            half8 hi, lo;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int ci = ch * 16 + 8 * (lane >> 5) + q;
                float v = 0.f;
                // (index algebra of pack_wh_kernel; forward weight [CoutF][cgs][J], as a data gradient its roles are swapped and the taps reversed)
                if (e.groups > 1) {
                    const int cout_g = e.cout_g[o], cin_g = e.cin_g[o];
                    const int gco = co / cout_g, cig = ((co / e.MT[o]) * e.MT[o] / cout_g) * cin_g + ci;
                    if (co < Cout && ci < e.cin_tile[o] && cig / cin_g == gco)
                        v = o ? wbank_src(e, w, cig, co - gco * cout_g, J - 1 - j) : wbank_src(e, w, co, cig - gco * cin_g, j);
                } else if (co < Cout && ci < Cin)
                    v = o ? wbank_src(e, w, ci, co, J - 1 - j) : wbank_src(e, w, co, ci, j);
                v *= scale;
                const _Float16 h = (_Float16)v;
                hi[q] = h;
                lo[q] = (_Float16)(v - (float)h);
            }
            half8* dst = reinterpret_cast<half8*>(e.pack[o]) + ((((size_t)j * nch + ch) * cotN + cot) * 2) * 64;
            dst[lane] = hi;
            dst[64 + lane] = lo;
        }
    }
}
}  // namespace ttsc

struct ttsc_wbank {
    int n = 0;
    WBankDev* tab_dev = nullptr;
    float* amax0 = nullptr;     // the entries' range words are one contiguous run [amax0, amax0 + n) when the caller laid them out so
    bool amax_contiguous = false;
    std::vector<WBankDev> host;
};

extern "C" int ttsc_wbank_create(const ttsc_wbank_entry* entries, int32_t n, ttsc_wbank** out) {
    TTSC_REQUIRE(entries && out && n > 0 && n <= 4096, "ttsc_wbank_create: bad argument");
    auto* b = new ttsc_wbank();
    b->n = n;
    b->host.resize(n);
    b->amax_contiguous = true;
    for (int i = 0; i < n; ++i) {
        const ttsc_wbank_entry& s = entries[i];
        WBankDev& d = b->host[i];
        const bool ok = s.v && s.amax && s.Cout > 0 && s.Cin > 0 && s.K > 0 && s.groups >= 1 && s.stride >= 1 && s.Cin % s.groups == 0 &&
                        s.Cout % s.groups == 0 && (!s.g || (s.w && s.norm)) && (s.pack_fwd || s.pack_dgrad);
        const int J = ok ? (s.K + s.stride - 1) / s.stride : 0, CinE = ok ? s.stride * s.Cin : 0;
        if (!ok || !ttsc_conv_train_supported(CinE, s.Cout, J, 1, s.groups) || (s.pack_dgrad && !ttsc_conv_train_supported(s.Cout, CinE, J, 1, s.groups))) {
            delete b;
            set_error("ttsc_wbank_create: entry %d is not a layer ttsc_conv_train takes (Cin %d, Cout %d, K %d, stride %d, groups %d)", i, s.Cin, s.Cout, s.K,
                      s.stride, s.groups);
            return TTSC_EINVAL;
        }
        d.v = s.v;
        d.g = s.g;
        d.w = s.w;
        d.norm = s.norm;
        d.amax = s.amax;
        d.pack[0] = reinterpret_cast<_Float16*>(s.pack_fwd);
        d.pack[1] = reinterpret_cast<_Float16*>(s.pack_dgrad);
        d.rows = s.Cout;
        d.Cg = s.Cin / s.groups;
        d.K = s.K;
        d.cols = d.Cg * s.K;
        d.stride = s.stride;
        d.J = J;
        d.groups = s.groups;
        for (int o = 0; o < 2; ++o) {
            const int Ci = o ? s.Cout : CinE, Co = o ? CinE : s.Cout;      // the stride-1 convolution this orientation feeds
            const int MT = train_mt(Co, s.groups, J), cin_g = Ci / s.groups, cout_g = Co / s.groups;
            const int cin_tile = s.groups > 1 ? (MT > cout_g ? MT / cout_g : 1) * cin_g : Ci;
            d.Cin[o] = Ci;
            d.Cout[o] = Co;
            d.MT[o] = MT;
            d.cin_g[o] = cin_g;
            d.cout_g[o] = cout_g;
            d.cin_tile[o] = cin_tile;
            d.nch[o] = round_up(cin_tile, 16) / 16;
            d.cotN[o] = round_up(Co, MT) / 32;
        }
        if (i > 0 && s.amax != entries[i - 1].amax + 1) b->amax_contiguous = false;
    }
    b->amax0 = entries[0].amax;
    if (hipMalloc((void**)&b->tab_dev, sizeof(WBankDev) * n) != hipSuccess ||
        hipMemcpy(b->tab_dev, b->host.data(), sizeof(WBankDev) * n, hipMemcpyHostToDevice) != hipSuccess) {
        if (b->tab_dev) (void)hipFree(b->tab_dev);
        delete b;
        set_error("ttsc_wbank_create: device table allocation failed");
        return TTSC_EHIP;
    }
    *out = b;
    return TTSC_OK;
}

extern "C" void ttsc_wbank_destroy(ttsc_wbank* b) {
    if (!b) return;
    if (b->tab_dev) (void)hipFree(b->tab_dev);
    delete b;
}

extern "C" int ttsc_wbank_prepare(ttsc_wbank* b, void* stream) {
    TTSC_REQUIRE(b && b->tab_dev, "ttsc_wbank_prepare: null argument");
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipSuccess;
    if (b->amax_contiguous)
        e = hipMemsetAsync(b->amax0, 0, sizeof(float) * b->n, s);
    else
        for (int i = 0; i < b->n && e == hipSuccess; ++i) e = hipMemsetAsync(b->host[i].amax, 0, sizeof(float), s);
    if (e != hipSuccess) {
        set_error("ttsc_wbank_prepare: hipMemsetAsync: %s", hipGetErrorString(e));
        return TTSC_EHIP;
    }
    hipLaunchKernelGGL(wbank_norm_kernel, dim3(256, (unsigned)b->n), dim3(256), 0, s, b->tab_dev);
    hipLaunchKernelGGL(wbank_pack_kernel, dim3(128, (unsigned)b->n, 2), dim3(256), 0, s, b->tab_dev);
    e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("ttsc_wbank_prepare launch failed: %s", hipGetErrorString(e));
        return TTSC_EHIP;
    }
    return TTSC_OK;
}
