// HiFi-GAN generator forward (V1/V2/V3 style configs; ResBlock1 and ResBlock2) as a chain of fused
// conv launches on one stream.  Mirrors `hifigan.models.Generator.forward` [EXTERNAL submodule of the
// reference; call sites cube/networks/cubegan.py:72,83,131 and cube/io_utils/runtime.py:78]:
//
//   x = conv_pre(mel)
//   for each upsample i:  x = ups[i](lrelu(x, 0.1));  x = mean_j resblock[i*nk+j](x)
//   wav = tanh(conv_post(lrelu(x, 0.01)))
//
// Fusions (all inside the conv kernel's prologue/epilogue, see conv1d.hip):
//   - every leaky-relu is applied while staging the conv's input tile into LDS,
//   - the residual add `x = xt + x` is the epilogue of the second conv of each pair,
//   - the sum over the nk residual blocks is accumulated by the epilogue of each block's last conv and the
//     division by nk rides in the next layer's staging scale, i.e. (rb0 + rb1 + rb2) / nk exactly as
//     the reference computes it,
//   - tanh is the epilogue of conv_post.
//
// HBM layout: four activation buffers carved from the caller's workspace (X: stage input, XT: inner
// activation of a residual pair, R: running residual stream, S: sum over residual blocks), each
// [B, C_i, L_i] fp32 channel-major, sized for the largest stage.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <utility>

#include "common.hpp"

using namespace ttsc;

namespace {
struct Layer {
    ttsc_conv1d* c = nullptr;
    std::vector<float> w;  // staged host weight until bias+weight are both present
    std::vector<float> b;
    bool has_w = false, has_b = false, dirty = false, uploaded = false;
    std::vector<int64_t> wshape;
    ~Layer() { ttsc_conv1d_destroy(c); }
};
}  // namespace

struct ttsc_hifigan {
    ttsc_hifigan_cfg cfg;
    std::map<std::string, std::unique_ptr<Layer>> layers;
    std::vector<int> stage_ch;  // channels after upsample i
    bool use_fused = true;      // env TTSC_HIFIGAN_FUSED=0 disables the fused residual-pair kernel (A/B measurements)
    bool use_chain = true;      // env TTSC_HIFIGAN_CHAIN=0 disables the whole-ResBlock fused chain kernel (resblock.hip)
    bool use_chain128 = true;   // env TTSC_HIFIGAN_CHAIN128=0: keep the 128-channel K=3 block on the layer-by-layer wide kernel (A/B)
    int chain_shape = -1;       // env TTSC_HIFIGAN_CHAIN_SHAPE: tile shape of the chain kernel (-1 = by halo)
    int split_chain = 1;        // env TTSC_HIFIGAN_CHAIN_SPLIT=0: every chained ResBlock1 as ONE launch; 2: split whatever the batch size (see chain_first_pairs)
    bool pad_pitch = true;      // env TTSC_HIFIGAN_PITCH=0: intermediate tensors with their natural row pitch (rows not on 128-byte boundaries)
    bool fuse_post = true;      // env TTSC_HIFIGAN_FUSE_POST=0: conv_post + tanh as their own launch instead of the epilogue of the last chain launch
    // Branch streams (round 6): the ResBlocks of a stage that runs layer by layer (stages 1 - 2: six launches per block) are independent given the stage
    // input; on the caller's stream + two side streams their launches fill each other's tails — a launch whose workgroups are not a whole number of
    // rounds over the 2 x 256 resident slots idles most of the chip for its last round, which is the rule for ragged batches and for single sentences
    // (a launch of a few dozen workgroups) and the exception for the dense headline batch (64 x 8 s: exactly 4 rounds at stage 1).  The blocks' sums
    // still enter S in block order (an event between the blocks' LAST launches), so every output bit is what the one-stream schedule gives.
    // env TTSC_HIFIGAN_BRANCH_STREAMS: 0 = off, 1 = by the rule in branch_streams_for() (default), 2 = whenever the buffers allow.
    int branch_streams = 1;
    hipStream_t side[2] = {nullptr, nullptr};
    bool side_owned = false;   // false: handed in by ttsc_hifigan_set_branch_streams (not destroyed with the handle)
    hipEvent_t ev_fork = nullptr, ev_acc[TTSC_HIFIGAN_MAX_RB] = {nullptr};
    int side_dev = -1;
    int precision = TTSC_PREC_FP32;
    // Split precision keeps activations as fp16 (hi, lo) pairs, so every layer's input gets a power-of-two pre-scale that
    // centres it in fp16's range (ttsc_conv1d_set_activation_scale).  The scales come from ONE calibration forward, run layer
    // by layer with an abs-max reduction in front of every convolution, and stay fixed until the weights change.
    //   calib_mode 1 (default, env TTSC_HIFIGAN_CALIBRATE unset / "probe"): calibrated on a FIXED built-in probe mel (log-mel
    //       range [-5, 1] of cube/io_utils/vocoder.py:96-98: noise frames, all-floor frames, all-ceiling frames) the first time
    //       the weights are used — the scales are a function of the weights alone, so a handle's output never depends on which
    //       utterance it saw first;
    //   calib_mode 2 ("input"): calibrated on the first forward's own input (round 2's behaviour);
    //   calib_mode 0 ("0"): all scales 1.
    // Every forward is guarded: conv_post raises a device word when it emits a non-finite sample (= an fp16 overflow anywhere
    // upstream, see conv_cout1_kernel); the forward then re-calibrates on the offending input and reruns, and only reports
    // TTSC_ERANGE when the output is still non-finite (non-finite input or weights).  range_check: 1 = as described (one stream
    // synchronisation per forward; default), 2 = deferred: the forward never waits, the guard word stays sticky on the device and
    // ttsc_hifigan_range_status() reads and clears it whenever the caller can afford the synchronisation (pipelined callers whose
    // host work for the next batch overlaps this forward), 0 = off (env TTSC_HIFIGAN_RANGE_CHECK).
    int calib_mode = 1;
    bool calibrated = false;
    int range_check = 1;
    int recalibrations = 0;          // forwards that tripped the guard and were rerun after re-calibration
    unsigned* flag_dev = nullptr;    // [0] the guard word (non-finite sample), [1] max |mel| of the guarded forward's input (float bits)
    unsigned* flag_host = nullptr;   // pinned copy of both
    // Low side of the guard (round 4): the pre-scales were derived for inputs of a certain magnitude (calib_in_absmax = max |mel| of the
    // calibration data: 5 for the built-in probe).  An input whose largest value sits more than 2^-10 below that pushes every layer's
    // activations the same way (the generator is positively homogeneous up to its biases), the lo halves of the fp16 pairs drift into
    // subnormals and RELATIVE accuracy degrades silently (measured: 6e-5 relative at 2^-16, 2e-5 asked).  Such a forward is treated
    // like an overflow — rerun with scales derived from the offending input — except that the handle's own scales are restored afterwards.
    float calib_in_absmax = 0.f;
    ~ttsc_hifigan() {
        if (flag_dev) (void)hipFree(flag_dev);
        if (flag_host) (void)hipHostFree(flag_host);
        for (auto& st : side)
            if (st && side_owned) (void)hipStreamDestroy(st);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        for (auto& e : ev_acc)
            if (e) (void)hipEventDestroy(e);
    }
    // side streams and events of the branch schedule, created on first use on the current device
    int ensure_side_streams() {
        int dev = 0;
        TTSC_HIP_CHECK(hipGetDevice(&dev));
        if (side[0] && ev_fork && side_dev == dev) return TTSC_OK;
        TTSC_REQUIRE(!side[0] || !ev_fork, "ttsc_hifigan_forward: the handle's branch streams belong to device %d, the call runs on device %d", side_dev, dev);
        if (side[0]) {   // streams handed in by the caller: only the events are the handle's
            TTSC_REQUIRE(side_dev == dev, "ttsc_hifigan_forward: the branch streams handed in belong to device %d, the call runs on device %d", side_dev, dev);
            TTSC_HIP_CHECK(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
            for (auto& e : ev_acc) TTSC_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            return TTSC_OK;
        }
        for (auto& st : side) TTSC_HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        side_owned = true;
        TTSC_HIP_CHECK(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
        for (auto& e : ev_acc) TTSC_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        side_dev = dev;
        return TTSC_OK;
    }
    // pack + upload every layer whose host copy changed; returns the name of the first incomplete layer (or "")
    int flush_weights(std::string* missing) {
        for (auto& kv : layers) {
            Layer* L = kv.second.get();
            if (!(L->has_w && L->has_b)) {
                *missing = kv.first;
                return TTSC_ESTATE;
            }
            if (L->dirty) {
                int rc = ttsc_conv1d_set_weight(L->c, L->w.data(), L->b.data());
                if (rc) return rc;
                L->dirty = false;
                L->uploaded = true;
                calibrated = false;   // new weights, new activation statistics
            }
        }
        return TTSC_OK;
    }
};

static int add_layer(ttsc_hifigan* g, const std::string& name, int cin, int cout, int k, int stride, int pad, int dil,
                     int transposed) {
    ttsc_conv1d_cfg c{cin, cout, k, stride, pad, dil, transposed};
    std::unique_ptr<Layer> L(new Layer());
    int rc = ttsc_conv1d_create(&c, &L->c);
    if (rc) return rc;
    if (transposed)
        L->wshape = {cin, cout, k};
    else
        L->wshape = {cout, cin, k};
    g->layers[name] = std::move(L);
    return TTSC_OK;
}

extern "C" int ttsc_hifigan_create(const ttsc_hifigan_cfg* cfg, ttsc_hifigan** out) {
    TTSC_REQUIRE(cfg && out, "ttsc_hifigan_create: null argument");
    TTSC_REQUIRE(cfg->num_upsamples > 0 && cfg->num_upsamples <= TTSC_HIFIGAN_MAX_UPS, "bad num_upsamples %d", cfg->num_upsamples);
    TTSC_REQUIRE(cfg->num_kernels > 0 && cfg->num_kernels <= TTSC_HIFIGAN_MAX_RB, "bad num_kernels %d", cfg->num_kernels);
    TTSC_REQUIRE(cfg->resblock == 1 || cfg->resblock == 2, "resblock must be 1 or 2, got %d", cfg->resblock);
    TTSC_REQUIRE(cfg->num_mels > 0 && cfg->upsample_initial_channel > 0, "bad num_mels / upsample_initial_channel");
    TTSC_REQUIRE((cfg->upsample_initial_channel >> cfg->num_upsamples) >= 1, "upsample_initial_channel too small");
    std::unique_ptr<ttsc_hifigan> g(new ttsc_hifigan());
    g->cfg = *cfg;
    if (const char* ev = getenv("TTSC_HIFIGAN_FUSED")) g->use_fused = atoi(ev) != 0;
    if (const char* ev = getenv("TTSC_HIFIGAN_CHAIN")) g->use_chain = atoi(ev) != 0;
    if (const char* ev = getenv("TTSC_HIFIGAN_CHAIN128")) g->use_chain128 = atoi(ev) != 0;
    if (const char* ev = getenv("TTSC_HIFIGAN_CHAIN_SHAPE")) g->chain_shape = atoi(ev);
    if (const char* ev = getenv("TTSC_HIFIGAN_CHAIN_SPLIT")) g->split_chain = atoi(ev);
    if (const char* ev = getenv("TTSC_HIFIGAN_PITCH")) g->pad_pitch = atoi(ev) != 0;
    if (const char* ev = getenv("TTSC_HIFIGAN_FUSE_POST")) g->fuse_post = atoi(ev) != 0;
    if (const char* ev = getenv("TTSC_HIFIGAN_BRANCH_STREAMS")) g->branch_streams = atoi(ev);
    if (const char* ev = getenv("TTSC_HIFIGAN_CALIBRATE")) {
        const std::string v(ev);
        g->calib_mode = (v == "0" || v == "off") ? 0 : (v == "input" || v == "2") ? 2 : 1;
    }
    if (const char* ev = getenv("TTSC_HIFIGAN_RANGE_CHECK")) g->range_check = atoi(ev);
    int rc = add_layer(g.get(), "conv_pre", cfg->num_mels, cfg->upsample_initial_channel, 7, 1, 3, 1, 0);
    if (rc) return rc;
    int ch = cfg->upsample_initial_channel;
    for (int i = 0; i < cfg->num_upsamples; ++i) {
        const int u = cfg->upsample_rates[i], k = cfg->upsample_kernel_sizes[i];
        TTSC_REQUIRE(u >= 1 && k >= u, "upsample %d: need kernel_size >= rate >= 1 (k=%d,u=%d)", i, k, u);
        const int cout = ch / 2;
        rc = add_layer(g.get(), "ups." + std::to_string(i), ch, cout, k, u, (k - u) / 2, 1, 1);
        if (rc) return rc;
        for (int j = 0; j < cfg->num_kernels; ++j) {
            const int rk = cfg->resblock_kernel_sizes[j];
            TTSC_REQUIRE(rk % 2 == 1, "resblock kernel size must be odd, got %d", rk);
            TTSC_REQUIRE(cfg->num_dilations[j] > 0 && cfg->num_dilations[j] <= TTSC_HIFIGAN_MAX_DIL, "bad num_dilations");
            const std::string rb = "resblocks." + std::to_string(i * cfg->num_kernels + j);
            for (int m = 0; m < cfg->num_dilations[j]; ++m) {
                const int d = cfg->resblock_dilation_sizes[j][m];
                if (cfg->resblock == 1) {
                    rc = add_layer(g.get(), rb + ".convs1." + std::to_string(m), cout, cout, rk, 1, d * (rk - 1) / 2, d, 0);
                    if (rc) return rc;
                    rc = add_layer(g.get(), rb + ".convs2." + std::to_string(m), cout, cout, rk, 1, (rk - 1) / 2, 1, 0);
                    if (rc) return rc;
                } else {
                    rc = add_layer(g.get(), rb + ".convs." + std::to_string(m), cout, cout, rk, 1, d * (rk - 1) / 2, d, 0);
                    if (rc) return rc;
                }
            }
        }
        ch = cout;
        g->stage_ch.push_back(ch);
    }
    rc = add_layer(g.get(), "conv_post", ch, 1, 7, 1, 3, 1, 0);
    if (rc) return rc;
    TTSC_HIP_CHECK(hipMalloc((void**)&g->flag_dev, 64));
    TTSC_HIP_CHECK(hipMemset(g->flag_dev, 0, 64));
    {
        const unsigned inf_bits = 0x7f800000u;   // word 2: running minimum of the deferred forwards' input maxima
        TTSC_HIP_CHECK(hipMemcpy(g->flag_dev + 2, &inf_bits, sizeof(unsigned), hipMemcpyHostToDevice));
    }
    TTSC_HIP_CHECK(hipHostMalloc((void**)&g->flag_host, 64, hipHostMallocDefault));
    g->flag_host[0] = g->flag_host[1] = 0u;
    rc = ttsc_conv1d_set_nonfinite_flag(g->layers.at("conv_post")->c, g->flag_dev);
    if (rc) return rc;
    *out = g.release();
    return TTSC_OK;
}

extern "C" void ttsc_hifigan_destroy(ttsc_hifigan* g) { delete g; }

extern "C" int ttsc_hifigan_set_precision(ttsc_hifigan* g, int32_t precision) {
    TTSC_REQUIRE(g, "ttsc_hifigan_set_precision: null argument");
    for (auto& kv : g->layers) {
        int rc = ttsc_conv1d_set_precision(kv.second->c, precision);
        if (rc) return rc;
    }
    g->precision = precision;
    g->calibrated = false;
    return TTSC_OK;
}

extern "C" int ttsc_hifigan_set_weight(ttsc_hifigan* g, const char* name, const float* host, const int64_t* shape,
                                       int32_t nd) {
    TTSC_REQUIRE(g && name && host && shape, "ttsc_hifigan_set_weight: null argument");
    std::string n(name);
    const bool is_w = n.size() > 7 && n.compare(n.size() - 7, 7, ".weight") == 0;
    const bool is_b = n.size() > 5 && n.compare(n.size() - 5, 5, ".bias") == 0;
    TTSC_REQUIRE(is_w || is_b, "ttsc_hifigan_set_weight: '%s' is neither .weight nor .bias (fold weight_norm first)", name);
    std::string base = n.substr(0, n.size() - (is_w ? 7 : 5));
    auto it = g->layers.find(base);
    TTSC_REQUIRE(it != g->layers.end(), "ttsc_hifigan_set_weight: unknown layer '%s'", base.c_str());
    Layer* L = it->second.get();
    if (is_w) {
        TTSC_REQUIRE(nd == 3 && shape[0] == L->wshape[0] && shape[1] == L->wshape[1] && shape[2] == L->wshape[2],
                     "ttsc_hifigan_set_weight: '%s' expects shape [%lld,%lld,%lld]", name, (long long)L->wshape[0],
                     (long long)L->wshape[1], (long long)L->wshape[2]);
        L->w.assign(host, host + shape[0] * shape[1] * shape[2]);
        L->has_w = true;
    } else {
        TTSC_REQUIRE(nd == 1, "ttsc_hifigan_set_weight: '%s' must be 1-D", name);
        L->b.assign(host, host + shape[0]);
        L->has_b = true;
    }
    if (L->has_w && L->has_b) {
        // bias length check against the layer's Cout (weight dim 0 for Conv1d, dim 1 for ConvTranspose1d)
        const bool transposed = base.compare(0, 4, "ups.") == 0;
        const int64_t cout = transposed ? L->wshape[1] : L->wshape[0];
        TTSC_REQUIRE((int64_t)L->b.size() == cout, "ttsc_hifigan_set_weight: '%s.bias' expects %lld elements", base.c_str(),
                     (long long)cout);
    }
    L->dirty = true;  // packed + uploaded lazily by the next forward (flush_weights)
    return TTSC_OK;
}

extern "C" int64_t ttsc_hifigan_out_len(const ttsc_hifigan* g, int64_t T) {
    if (!g) return TTSC_EINVAL;
    int64_t L = T;
    for (int i = 0; i < g->cfg.num_upsamples; ++i) {
        const int u = g->cfg.upsample_rates[i], k = g->cfg.upsample_kernel_sizes[i];
        L = (L - 1) * u - 2 * ((k - u) / 2) + k;
    }
    return L;
}

// Row pitch of the intermediate tensors of stage i + 1 (the output of upsampler i) for rows of L samples: a multiple of 32 floats, so that
// every row starts on a 128-byte line (4001 / 12004 / 48016 samples per row at BASELINE config[1]: the 128-byte stores of a tile straddled two
// lines, the wide kernels wrote 1.23-1.26x their tensor — round-4 PMC).  The LAST stage keeps its natural pitch: its rows are the caller's waveform.
static int64_t stage_pitch(const ttsc_hifigan* g, int i, int64_t L) {
    return (g->pad_pitch && i + 1 < g->cfg.num_upsamples) ? round_up(L, 32) : L;
}

static size_t buf_elems(const ttsc_hifigan* g, int32_t B, int64_t T) {
    size_t mx = (size_t)B * g->cfg.upsample_initial_channel * T;
    int64_t L = T;
    for (int i = 0; i < g->cfg.num_upsamples; ++i) {
        const int u = g->cfg.upsample_rates[i], k = g->cfg.upsample_kernel_sizes[i];
        L = (L - 1) * u - 2 * ((k - u) / 2) + k;
        size_t e = (size_t)B * g->stage_ch[i] * stage_pitch(g, i, L);
        if (e > mx) mx = e;
    }
    return (size_t)round_up((int64_t)mx, 64);
}

// A chained ResBlock1 recomputes `halo` columns per tile side — the summed half receptive fields of ALL its convolutions: 36 / 60 columns at K = 7 / 11, 14 % / 23 %
// of the 512-column tile of the 64-channel kernel.  Cut in two launches (first `n` pairs X -> R, the rest R -> S) the halos are 6 + 30 (K = 7 as 1 + 2) / 30 + 30 columns (K = 11 as 2 + 1): 6 % /
// 13 % fewer MFMAs for one more tile fill and store, and at this operating point a chain launch costs what its MFMAs cost.  Measured at config[1]
// (tools/bench_chain_split.py, bit-identical results): 64 channels K = 11 5.30 -> 4.77 ms as 2 + 1, K = 7 3.60 -> 3.39 ms as 1 + 2; 32 channels (twice the bytes
// per MFMA, 1024-column tile): 3.65 -> 3.71 ms, 4.96 -> 5.96 ms — not split; K = 3: halo 12, nothing to win (64 channels 1.59 -> 1.83 ms; 128 channels 1.57 -> 1.42-1.52 ms
// alone, nothing in the whole forward).  Whole forward on one box, three alternations: 43.54-43.68 ms as single launches, 43.00-43.11 ms split.  Only when the launch fills the chip several times
// over: a short batch pays the second launch's latency instead.  Returns the number of pairs of the first launch, 0 = one launch.
// Branch streams for a layer-by-layer stage?  (ttsc_hifigan::branch_streams.)  Mode 1: ragged batches (their launches are never a whole number of
// workgroup rounds) and batches whose launches are short of two rounds of the wide kernel's 128 x 256 tiles; the dense headline batch keeps one stream
// (measured, tools/bench_streams.py / profiles/r06_stream_overlap_experiment.log: +1.7 % at stage 1, -4 .. -7 % at stage 2 — within what one box varies).
static bool branch_streams_for(const ttsc_hifigan* g, int32_t B, int ch, int64_t L, bool ragged) {
    if (g->branch_streams <= 0) return false;
    if (g->branch_streams >= 2) return true;
    const int64_t wgs = (int64_t)B * ceil_div(L, (int64_t)256) * std::max(1, ch / 128);
    return ragged || wgs < 2 * 512;
}

// Branch streams for a stage whose ResBlocks are ONE chain launch each?  Only where a launch leaves most of the chip idle (a sentence, a few short utterances:
// one 3 s utterance is 51 workgroups at 64 channels): blocks 1 and 2 then run all pairs but the last into temporaries (the R / XT buffers, unused by a chained
// stage) BESIDE block 0, and only their last pair — the launch that adds into S — waits for the block before.  A cut chain gives the bits of the whole one
// (tools/bench_chain_split.py), the sums enter S in block order: outputs identical bit for bit.  Large batches keep the single launches (the cut costs MFMAs
// at 32 channels: 4.96 -> 5.96 ms for the K = 11 block at config[1]).
static bool chain_branch_for(const ttsc_hifigan* g, int32_t B, int ch, int64_t L) {
    if (g->branch_streams <= 0) return false;
    if (g->branch_streams >= 2) return true;
    static const int tiles = getenv("TTSC_CHAIN_BRANCH_TILES") ? atoi(getenv("TTSC_CHAIN_BRANCH_TILES")) : 128;   // (measurement switch; tools/probes/gen_small_batches.py: the schedule pays for ONE utterance — 1.08 -> 0.92-0.99 ms at 3 s —, is a wash from two on)
    return (int64_t)B * ceil_div(L, (int64_t)(ch == 32 ? 900 : 400)) < tiles;
}

static int chain_first_pairs(const ttsc_hifigan* g, int ch, int k, int nd, int32_t B, int64_t L) {
    if (!g->split_chain || ch != 64 || nd != 3 || (k != 7 && k != 11)) return 0;
    if (g->split_chain < 2 && (int64_t)B * ceil_div(L, (int64_t)400) < 1024) return 0;   // < 4 rounds of workgroups over the 256 CUs
    return k == 11 ? 2 : 1;
}

// length table of a DENSE batch written on the device (no host buffer whose lifetime a stream would have to outlive): row i = stage i's length
__global__ void fill_lens_kernel(int32_t* tab, int B, int rows, const int32_t* __restrict__ unused, int l0, int l1, int l2, int l3, int l4, int l5, int l6, int l7, int l8) {
    const int v[9] = {l0, l1, l2, l3, l4, l5, l6, l7, l8};
    for (int i = threadIdx.x + blockIdx.x * blockDim.x; i < B * rows; i += blockDim.x * gridDim.x) tab[i] = v[i / B];
}

static size_t len_table_bytes(const ttsc_hifigan* g, int32_t B) {
    return (size_t)round_up((int64_t)(g->cfg.num_upsamples + 1) * B * sizeof(int32_t), 256);
}

extern "C" size_t ttsc_hifigan_workspace_bytes(const ttsc_hifigan* g, int32_t B, int64_t T) {
    if (!g || B <= 0 || T <= 0) return 0;
    return 4 * buf_elems(g, B, T) * sizeof(float) + len_table_bytes(g, B);
}

extern "C" int ttsc_hifigan_algorithmic_flops(const ttsc_hifigan* g, int32_t B, int64_t T, double* out) {
    TTSC_REQUIRE(g && out && B > 0 && T > 0, "ttsc_hifigan_algorithmic_flops: bad argument");
    const auto& c = g->cfg;
    double mac = (double)T * c.num_mels * c.upsample_initial_channel * 7;
    int64_t L = T;
    int ch = c.upsample_initial_channel;
    for (int i = 0; i < c.num_upsamples; ++i) {
        const int u = c.upsample_rates[i], k = c.upsample_kernel_sizes[i];
        mac += (double)L * ch * (ch / 2) * k;  // every input sample meets every tap once
        L = (L - 1) * u - 2 * ((k - u) / 2) + k;
        ch /= 2;
        for (int j = 0; j < c.num_kernels; ++j)
            mac += (double)L * ch * ch * c.resblock_kernel_sizes[j] * c.num_dilations[j] * (c.resblock == 1 ? 2 : 1);
    }
    mac += (double)L * ch * 7;
    *out = 2.0 * mac * B;
    return TTSC_OK;
}

extern "C" int ttsc_hifigan_forward(ttsc_hifigan* g, const float* mel, int32_t B, int64_t T, float* wav, void* ws,
                                    size_t ws_bytes, void* stream) {
    return ttsc_hifigan_forward_ragged(g, mel, B, T, nullptr, wav, ws, ws_bytes, stream);
}

static int hifigan_run(ttsc_hifigan* g, const float* mel, int32_t B, int64_t T, const int32_t* frames, float* wav, void* ws, size_t ws_bytes,
                       void* stream, float* calib_stat);

// scale that puts a layer's largest input magnitude m into [2^9, 2^10)
static float calib_scale(float m) {
    if (!(m > 0.f) || !std::isfinite(m)) return 1.f;
    int e = 0;
    (void)frexpf(m, &e);   // m = f * 2^e, f in [0.5, 1)
    e = 10 - e;
    e = e > 40 ? 40 : (e < -20 ? -20 : e);
    return ldexpf(1.f, e);
}

extern "C" int ttsc_hifigan_calibrate(ttsc_hifigan* g, const float* mel, int32_t B, int64_t T, float* wav, void* ws, size_t ws_bytes,
                                      void* stream) {
    TTSC_REQUIRE(g && mel && wav && ws, "ttsc_hifigan_calibrate: null argument");
    for (auto& kv : g->layers) {
        int rc = ttsc_conv1d_set_activation_scale(kv.second->c, 1.f);
        if (rc) return rc;
    }
    g->calibrated = true;   // (also keeps hifigan_run from recursing)
    if (g->precision != TTSC_PREC_F16X3) return TTSC_OK;
    // layer by layer: abs-max of the layer's input -> its scale -> the layer itself (so that every layer already runs inside
    // fp16's range and hands finite, accurate data to the next measurement); one host read per layer, calibration only
    float* stat = nullptr;
    TTSC_HIP_CHECK(hipMalloc((void**)&stat, sizeof(float)));
    float m = 0.f;
    if (hipMemsetAsync(stat, 0, sizeof(float), (hipStream_t)stream) != hipSuccess || ttsc_absmax(mel, (int64_t)B * g->cfg.num_mels * T, stat, stream) ||
        hipStreamSynchronize((hipStream_t)stream) != hipSuccess || hipMemcpy(&m, stat, sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) {
        (void)hipFree(stat);
        set_error("ttsc_hifigan_calibrate: measuring the input range failed");
        return TTSC_EHIP;
    }
    g->calib_in_absmax = std::isfinite(m) ? m : 0.f;
    int rc = hifigan_run(g, mel, B, T, nullptr, wav, ws, ws_bytes, stream, stat);
    (void)hipFree(stat);
    return rc;
}

// The built-in calibration probe: one utterance of 96 mel frames covering the log-mel range the generator is fed with
// (floor -5 = cube/io_utils/vocoder.py:96-98 / the collate's pad value, ceiling ~1): 32 frames of clip(N(-2,1)), 16 all-floor,
// 16 all-ceiling, 32 frames of clip(N(-1,1.5)).  Generated by a fixed integer recurrence (no library RNG: identical everywhere).
static void probe_mel(int num_mels, int T, std::vector<float>& out) {
    out.assign((size_t)num_mels * T, 0.f);
    uint64_t st = 0x9E3779B97F4A7C15ull;
    auto u01 = [&]() {
        st = st * 6364136223846793005ull + 1442695040888963407ull;
        return ((double)((st >> 11) & ((1ull << 53) - 1)) + 0.5) / (double)(1ull << 53);
    };
    auto normal = [&]() { return sqrt(-2.0 * log(u01())) * cos(6.283185307179586 * u01()); };
    for (int t = 0; t < T; ++t)
        for (int m = 0; m < num_mels; ++m) {
            double v;
            if (t < 32) v = -2.0 + normal();
            else if (t < 48) v = -5.0;
            else if (t < 64) v = 1.0;
            else v = -1.0 + 1.5 * normal();
            v = v < -5.0 ? -5.0 : (v > 1.0 ? 1.0 : v);
            out[(size_t)m * T + t] = (float)v;
        }
}

// weight-only calibration: the probe through ttsc_hifigan_calibrate on private buffers (once per weight change)
static int calibrate_on_probe(ttsc_hifigan* g, void* stream) {
    const int T = 96;
    std::vector<float> h;
    probe_mel(g->cfg.num_mels, T, h);
    const size_t wsb = ttsc_hifigan_workspace_bytes(g, 1, T);
    const int64_t Lout = ttsc_hifigan_out_len(g, T);
    float *mel = nullptr, *wav = nullptr;
    void* ws = nullptr;
    int rc = TTSC_OK;
    if (hipMalloc((void**)&mel, h.size() * sizeof(float)) != hipSuccess || hipMalloc((void**)&wav, (size_t)Lout * sizeof(float)) != hipSuccess ||
        hipMalloc(&ws, wsb) != hipSuccess) {
        set_error("ttsc_hifigan: cannot allocate the calibration probe buffers");
        rc = TTSC_ENOMEM;
    }
    if (!rc && hipMemcpy(mel, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) rc = TTSC_EHIP;
    if (!rc) rc = ttsc_hifigan_calibrate(g, mel, 1, T, wav, ws, wsb, stream);
    if (!rc && hipStreamSynchronize((hipStream_t)stream) != hipSuccess) rc = TTSC_EHIP;
    if (mel) (void)hipFree(mel);
    if (wav) (void)hipFree(wav);
    if (ws) (void)hipFree(ws);
    return rc;
}

extern "C" int ttsc_hifigan_set_activation_scales(ttsc_hifigan* g, const char* const* layers, const float* scales, int32_t n) {
    TTSC_REQUIRE(g && layers && scales && n > 0, "ttsc_hifigan_set_activation_scales: null argument");
    {
        std::string missing;
        int frc = g->flush_weights(&missing);   // the scales belong to the weights that are about to be used
        if (frc == TTSC_ESTATE) {
            set_error("ttsc_hifigan_set_activation_scales: weights missing (first: '%s')", missing.c_str());
            return TTSC_ESTATE;
        }
        if (frc) return frc;
    }
    TTSC_REQUIRE((size_t)n == g->layers.size(), "ttsc_hifigan_set_activation_scales: %d scales for %zu layers", n, g->layers.size());
    for (int i = 0; i < n; ++i) {
        auto it = g->layers.find(layers[i] ? layers[i] : "");
        TTSC_REQUIRE(it != g->layers.end(), "ttsc_hifigan_set_activation_scales: unknown layer '%s'", layers[i] ? layers[i] : "(null)");
        int rc = ttsc_conv1d_set_activation_scale(it->second->c, scales[i]);
        if (rc) return rc;
    }
    g->calibrated = true;
    if (!(g->calib_in_absmax > 0.f)) g->calib_in_absmax = 5.f;   // restored scales: assume the log-mel range of the built-in probe
    return TTSC_OK;
}

// Branch streams of the caller's choosing (see ttscube_hip.h).  Both null: back to streams the handle creates on first use.
extern "C" int ttsc_hifigan_set_branch_streams(ttsc_hifigan* g, void* stream_a, void* stream_b) {
    TTSC_REQUIRE(g, "ttsc_hifigan_set_branch_streams: null handle");
    TTSC_REQUIRE((stream_a == nullptr) == (stream_b == nullptr) && (!stream_a || stream_a != stream_b),
                 "ttsc_hifigan_set_branch_streams: two distinct streams, or two nulls");
    int dev = 0;
    TTSC_HIP_CHECK(hipGetDevice(&dev));
    TTSC_REQUIRE(!g->ev_fork || g->side_dev == dev, "ttsc_hifigan_set_branch_streams: the handle already ran its branch schedule on device %d", g->side_dev);
    for (auto& st : g->side) {
        if (st && g->side_owned) TTSC_HIP_CHECK(hipStreamDestroy(st));
        st = nullptr;
    }
    g->side_owned = false;
    g->side[0] = (hipStream_t)stream_a;
    g->side[1] = (hipStream_t)stream_b;
    if (stream_a) g->side_dev = dev;
    return TTSC_OK;
}

extern "C" int32_t ttsc_hifigan_recalibrations(const ttsc_hifigan* g) { return g ? g->recalibrations : -1; }

extern "C" int ttsc_hifigan_set_range_check(ttsc_hifigan* g, int32_t mode) {
    TTSC_REQUIRE(g && mode >= 0 && mode <= 2, "ttsc_hifigan_set_range_check: mode must be 0 (off), 1 (synchronous) or 2 (deferred)");
    g->range_check = mode;
    return TTSC_OK;
}

// Deferred guard: 1 when any forward since the last call emitted a non-finite sample (the word is then cleared and the handle
// marked un-calibrated, so the next forward calibrates afresh), 0 otherwise, < 0 on a HIP error.  Synchronises `stream`.
// deferred mode: fold one forward's max |mel| (word 1, then cleared) into the running MINIMUM over the forwards since the last status call (word 2)
__global__ void fold_input_range_kernel(unsigned* w) {
    const float m = __uint_as_float(w[1]);
    if (m < __uint_as_float(w[2])) w[2] = w[1];   // (a NaN / inf maximum is left to the non-finite guard)
    w[1] = 0u;
}

// the low side of the guard, from the pinned copy (word `idx`): the guarded forward's max |mel| sits more than 2^-10 below the calibration data's
static bool input_too_low(const ttsc_hifigan* g, int idx = 1) {
    float m;
    memcpy(&m, &g->flag_host[idx], sizeof(float));
    return g->calib_in_absmax > 0.f && std::isfinite(m) && m > 0.f && m < g->calib_in_absmax * (1.f / 1024.f);   // (an all-zero input has nothing to lose)
}

extern "C" int32_t ttsc_hifigan_range_status(ttsc_hifigan* g, void* stream) {
    if (!g) return TTSC_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemcpyAsync(g->flag_host, g->flag_dev, 3 * sizeof(unsigned), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
        set_error("ttsc_hifigan_range_status: reading the guard word failed");
        return TTSC_EHIP;
    }
    const bool low = input_too_low(g, 2);
    const unsigned inf_bits = 0x7f800000u;   // running minimum restarts at +inf
    if (hipMemcpyAsync(g->flag_dev + 2, &inf_bits, sizeof(unsigned), hipMemcpyHostToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return TTSC_EHIP;
    if (g->flag_host[0] == 0u && !low) return 0;
    if (hipMemsetAsync(g->flag_dev, 0, 2 * sizeof(unsigned), s) != hipSuccess) return TTSC_EHIP;
    g->calibrated = false;
    return 1;
}

extern "C" int ttsc_hifigan_get_activation_scale(const ttsc_hifigan* g, const char* layer, float* out) {
    TTSC_REQUIRE(g && layer && out, "ttsc_hifigan_get_activation_scale: null argument");
    auto it = g->layers.find(layer);
    TTSC_REQUIRE(it != g->layers.end(), "ttsc_hifigan_get_activation_scale: unknown layer '%s'", layer);
    *out = ttsc_conv1d_get_activation_scale(it->second->c);
    return TTSC_OK;
}

extern "C" int ttsc_hifigan_forward_ragged(ttsc_hifigan* g, const float* mel, int32_t B, int64_t T, const int32_t* frames,
                                           float* wav, void* ws, size_t ws_bytes, void* stream) {
    TTSC_REQUIRE(g, "ttsc_hifigan_forward: null argument");
    hipStream_t s = (hipStream_t)stream;
    const bool split_guarded = g->precision == TTSC_PREC_F16X3 && g->calib_mode != 0;   // (mode 0 is a measurement switch: scales stay 1)
    const int64_t nmel = (int64_t)B * g->cfg.num_mels * T;
    if (g->range_check == 2) {   // deferred: sticky words, see ttsc_hifigan_range_status
        if (split_guarded && mel && nmel > 0) {
            // word 1 may still hold the maximum of an earlier SYNCHRONOUS forward (that path clears it only at its own start): the
            // atomicMax below must start from zero, or a too-quiet input hides behind the stale value
            TTSC_HIP_CHECK(hipMemsetAsync(g->flag_dev + 1, 0, sizeof(unsigned), s));
            int arc = ttsc_absmax(mel, nmel, reinterpret_cast<float*>(g->flag_dev + 1), stream);
            if (arc) return arc;
            hipLaunchKernelGGL(fold_input_range_kernel, dim3(1), dim3(1), 0, s, g->flag_dev);
        }
        return hifigan_run(g, mel, B, T, frames, wav, ws, ws_bytes, stream, nullptr);
    }
    const bool guard = g->range_check == 1 && split_guarded;
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (guard) {
            TTSC_HIP_CHECK(hipMemsetAsync(g->flag_dev, 0, 2 * sizeof(unsigned), s));
            if (mel && nmel > 0) {
                int arc = ttsc_absmax(mel, nmel, reinterpret_cast<float*>(g->flag_dev + 1), stream);
                if (arc) return arc;
            }
        }
        int rc = hifigan_run(g, mel, B, T, frames, wav, ws, ws_bytes, stream, nullptr);
        if (rc || !guard) return rc;
        TTSC_HIP_CHECK(hipMemcpyAsync(g->flag_host, g->flag_dev, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, s));
        TTSC_HIP_CHECK(hipStreamSynchronize(s));
        const bool bad = g->flag_host[0] != 0u;
        const bool low = attempt == 0 && !bad && input_too_low(g);
        if (!bad && !low) return TTSC_OK;
        if (attempt == 1) break;
        g->recalibrations++;
        if (low) {
            // The input sits far BELOW the range the scales were derived for: this forward is rerun with scales derived from the input itself,
            // and the handle's own scales (normally the weight-only probe calibration) are put back afterwards — a quiet batch must not change
            // what the next ordinary batch computes (two handles with the same weights stay bit-identical whatever they saw before).
            std::vector<std::pair<ttsc_conv1d*, float>> saved;
            for (auto& kv : g->layers) saved.emplace_back(kv.second->c, ttsc_conv1d_get_activation_scale(kv.second->c));
            const float saved_absmax = g->calib_in_absmax;
            rc = ttsc_hifigan_calibrate(g, mel, B, T, wav, ws, ws_bytes, stream);
            if (!rc) {
                TTSC_HIP_CHECK(hipMemsetAsync(g->flag_dev, 0, 2 * sizeof(unsigned), s));
                rc = hifigan_run(g, mel, B, T, frames, wav, ws, ws_bytes, stream, nullptr);
            }
            if (!rc) {
                TTSC_HIP_CHECK(hipMemcpyAsync(g->flag_host, g->flag_dev, sizeof(unsigned), hipMemcpyDeviceToHost, s));
                TTSC_HIP_CHECK(hipStreamSynchronize(s));
            }
            for (auto& sv : saved) (void)ttsc_conv1d_set_activation_scale(sv.first, sv.second);
            g->calib_in_absmax = saved_absmax;
            if (rc) return rc;
            if (g->flag_host[0] == 0u) return TTSC_OK;
            break;
        }
        // a non-finite sample left conv_post: some layer's input overflowed the fp16 range its pre-scale was calibrated for
        // (or the input itself is non-finite).  Re-derive the scales on THIS input and run the forward again.
        rc = ttsc_hifigan_calibrate(g, mel, B, T, wav, ws, ws_bytes, stream);
        if (rc) return rc;
    }
    g->calibrated = false;   // the scales derived from a non-finite input are meaningless: the next forward calibrates afresh
    set_error("ttsc_hifigan_forward: non-finite output even after re-calibrating the split-precision scales on this input "
              "(non-finite values in the input or the weights?)");
    return TTSC_ERANGE;
}

static int hifigan_run(ttsc_hifigan* g, const float* mel, int32_t B, int64_t T, const int32_t* frames, float* wav, void* ws, size_t ws_bytes,
                       void* stream, float* calib_stat) {
    TTSC_REQUIRE(g && mel && wav && ws, "ttsc_hifigan_forward: null argument");
    TTSC_REQUIRE(B > 0 && T > 0, "ttsc_hifigan_forward: bad B/T (%d, %lld)", B, (long long)T);
    {
        std::string missing;
        int frc = g->flush_weights(&missing);
        if (frc == TTSC_ESTATE) {
            set_error("ttsc_hifigan_forward: weights missing (first: '%s')", missing.c_str());
            return TTSC_ESTATE;
        }
        if (frc) return frc;
    }
    const size_t need = ttsc_hifigan_workspace_bytes(g, B, T);
    if (ws_bytes < need) {
        set_error("ttsc_hifigan_forward: workspace %zu < required %zu bytes", ws_bytes, need);
        return TTSC_ENOMEM;
    }
    const bool calib = calib_stat != nullptr;
    if (!calib && !g->calibrated && g->calib_mode != 0 && g->precision == TTSC_PREC_F16X3) {   // first forward after new weights
        int crc = g->calib_mode == 1 ? calibrate_on_probe(g, stream) : ttsc_hifigan_calibrate(g, mel, B, T, wav, ws, ws_bytes, stream);
        if (crc) return crc;
    }
    const auto& c = g->cfg;
    const size_t be = buf_elems(g, B, T);
    float* X = (float*)ws;
    float* XT = X + be;
    float* R = XT + be;
    float* S = R + be;
    // per-stage valid lengths of every utterance: row 0 = mel frames, row i+1 = samples after upsample i
    const int32_t* lens[TTSC_HIFIGAN_MAX_UPS + 1] = {nullptr};
    bool last_lens_mult4 = true;   // (the fused conv_post epilogue moves the tile as 4-sample vectors)
    if (frames) {
        std::vector<int32_t> tab((size_t)(c.num_upsamples + 1) * B);
        for (int b = 0; b < B; ++b) {
            TTSC_REQUIRE(frames[b] >= 0 && frames[b] <= T, "ttsc_hifigan_forward_ragged: frames[%d]=%d outside [0,%lld]", b, frames[b], (long long)T);
            int64_t Lb = frames[b];
            tab[b] = (int32_t)Lb;
            for (int i = 0; i < c.num_upsamples; ++i) {
                const int u = c.upsample_rates[i], k = c.upsample_kernel_sizes[i];
                Lb = Lb > 0 ? (Lb - 1) * u - 2 * ((k - u) / 2) + k : 0;
                tab[(size_t)(i + 1) * B + b] = (int32_t)Lb;
            }
        }
        for (int b = 0; b < B; ++b) last_lens_mult4 = last_lens_mult4 && (tab[(size_t)c.num_upsamples * B + b] % 4 == 0);
        int32_t* dtab = (int32_t*)(S + be);   // behind the fourth buffer
        TTSC_HIP_CHECK(hipMemcpyAsync(dtab, tab.data(), tab.size() * sizeof(int32_t), hipMemcpyHostToDevice, (hipStream_t)stream));
        TTSC_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));  // `tab` is pageable host memory going out of scope
        for (int i = 0; i <= c.num_upsamples; ++i) lens[i] = dtab + (size_t)i * B;
    }
    // real length (of the longest utterance) and row pitch of every stage; the calibration forward reduces whole buffers, so it keeps natural pitches
    int64_t Lr[TTSC_HIFIGAN_MAX_UPS + 1], Pp[TTSC_HIFIGAN_MAX_UPS + 1];
    Lr[0] = Pp[0] = T;
    bool any_pad = false;
    for (int i = 0; i < c.num_upsamples; ++i) {
        const int u = c.upsample_rates[i], k = c.upsample_kernel_sizes[i];
        Lr[i + 1] = (Lr[i] - 1) * u - 2 * ((k - u) / 2) + k;
        Pp[i + 1] = calib ? Lr[i + 1] : stage_pitch(g, i, Lr[i + 1]);
        any_pad = any_pad || Pp[i + 1] != Lr[i + 1];
    }
    if (!frames) last_lens_mult4 = Lr[c.num_upsamples] % 4 == 0;
    if (any_pad && !frames) {   // a padded pitch needs per-utterance lengths: the dense batch's table, written by a kernel
        static_assert(TTSC_HIFIGAN_MAX_UPS + 1 <= 9, "fill_lens_kernel takes nine rows");
        int32_t* dtab = (int32_t*)(S + be);
        int v[9] = {0};
        for (int i = 0; i <= c.num_upsamples; ++i) v[i] = (int)Lr[i];
        hipLaunchKernelGGL(fill_lens_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, dtab, (int)B, c.num_upsamples + 1, (const int32_t*)nullptr, v[0], v[1],
                           v[2], v[3], v[4], v[5], v[6], v[7], v[8]);
        TTSC_HIP_CHECK(hipGetLastError());
        for (int i = 0; i <= c.num_upsamples; ++i) lens[i] = dtab + (size_t)i * B;
    }
    auto layer = [&](const std::string& n) -> ttsc_conv1d* { return g->layers.at(n)->c; };
    int rc;

    // Lin: row pitch of x (= its length when no lengths are given); Lout_pitch: row pitch of y (0 = natural)
    auto conv = [&](ttsc_conv1d* l, const float* x, int64_t Lin, float* y, const float* resid, const ttsc_conv1d_epilogue& e, const int32_t* il,
                    const int32_t* ol, int64_t Lout_pitch = 0, void* st = nullptr) {
        if (!st) st = stream;
        if (calib) {   // calibration forward: abs-max of this layer's input -> its pre-scale, then the layer
            float m = 0.f;
            if (hipMemsetAsync(calib_stat, 0, sizeof(float), (hipStream_t)stream) != hipSuccess) return (int)TTSC_EHIP;
            int arc = ttsc_absmax(x, (int64_t)B * ttsc_conv1d_in_channels(l) * Lin, calib_stat, stream);
            if (arc) return arc;
            if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess ||
                hipMemcpy(&m, calib_stat, sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) {
                set_error("ttsc_hifigan_calibrate: reading the activation statistics failed");
                return (int)TTSC_EHIP;
            }
            arc = ttsc_conv1d_set_activation_scale(l, calib_scale(m * fabsf(e.in_scale)));
            if (arc) return arc;
        }
        return ttsc_conv1d_forward_pitched(l, x, B, Lin, y, resid, &e, il, ol, (il && ol) ? Lout_pitch : 0, st);
    };
    const float inv_nk = 1.f / (float)c.num_kernels;
    ttsc_conv1d_epilogue ep{1.f, 1.f, 1.f, TTSC_ACT_NONE, 0};
    int64_t L = T;
    float sum_scale = 1.f;   // pending division by nk of the previous stage's block sum (fp32 consumers)
    rc = conv(layer("conv_pre"), mel, T, S, nullptr, ep, lens[0], lens[0]);
    if (rc) return rc;
    for (int i = 0; i < c.num_upsamples; ++i) {
        const int ch = g->stage_ch[i];
        ttsc_conv1d* up = layer("ups." + std::to_string(i));
        // is this stage's ResBlock1 chain run by the fused 32-channel pair kernel?
        bool fused_stage = !calib && (c.resblock == 1) && g->use_fused;
        for (int j = 0; fused_stage && j < c.num_kernels; ++j)
            for (int m = 0; fused_stage && m < c.num_dilations[j]; ++m) {
                const std::string rb = "resblocks." + std::to_string(i * c.num_kernels + j);
                fused_stage = ttsc_respair_supported(layer(rb + ".convs1." + std::to_string(m)), layer(rb + ".convs2." + std::to_string(m))) != 0;
            }
        // ... or, better, which ResBlock1s of the stage run as ONE fused chain launch each (32 / 64 channels: all of them; 128 channels:
        // the K = 3 block, whose image + halo still fit the LDS)?
        bool chain_rb[TTSC_HIFIGAN_MAX_RB] = {false};
        bool chain_stage = !calib && (c.resblock == 1) && g->use_chain;   // every block of the stage chained
        for (int j = 0; j < c.num_kernels; ++j) {
            const std::string rb = "resblocks." + std::to_string(i * c.num_kernels + j);
            const ttsc_conv1d *c1[TTSC_HIFIGAN_MAX_DIL], *c2[TTSC_HIFIGAN_MAX_DIL];
            const int nd = c.num_dilations[j];
            bool ok = !calib && (c.resblock == 1) && g->use_chain && nd <= 3;
            for (int m = 0; ok && m < nd; ++m) {
                c1[m] = layer(rb + ".convs1." + std::to_string(m));
                c2[m] = layer(rb + ".convs2." + std::to_string(m));
            }
            ok = ok && ttsc_rbchain_supported(c1, c2, nd) != 0;
            if (ok && ch >= 128 && !g->use_chain128) ok = false;
            chain_rb[j] = ok;
            chain_stage = chain_stage && ok;
        }
        if (chain_stage) fused_stage = false;
        // x = ups[i](lrelu(x / nk_prev, 0.1))
        ttsc_conv1d_epilogue eu{sum_scale, 0.1f, 1.f, TTSC_ACT_NONE, 0};
        rc = conv(up, S, L, X, nullptr, eu, lens[i], lens[i + 1], Pp[i + 1]);
        if (rc) return rc;
        TTSC_REQUIRE(ttsc_conv1d_out_len(up, Lr[i]) == Lr[i + 1], "ttsc_hifigan_forward: upsampler %d disagrees with the configured rates about its output length", i);
        L = Pp[i + 1];   // from here on L is the ROW PITCH of the stage's tensors; the real lengths are in `ln`
        const int32_t* ln = lens[i + 1];
        // ---- branch schedule of this stage (see ttsc_hifigan::branch_streams): block j on stream bst[j] with temporaries of its own ----
        const size_t n_i = (size_t)round_up((int64_t)B * ch * L, 64);
        const bool chain_branch = chain_stage && chain_branch_for(g, B, ch, L);
        bool branch = !calib && !fused_stage && c.resblock == 1 && c.num_kernels >= 2 && c.num_kernels <= 3 &&
                      (chain_stage ? chain_branch : (3 * n_i <= be && branch_streams_for(g, B, ch, L, ln != nullptr && frames != nullptr)));
        void* bst[TTSC_HIFIGAN_MAX_RB];
        float *bXT[TTSC_HIFIGAN_MAX_RB], *bR[TTSC_HIFIGAN_MAX_RB];
        for (int j = 0; j < c.num_kernels; ++j) {
            bst[j] = stream;
            bXT[j] = XT;
            bR[j] = R;
        }
        if (branch) {
            if ((rc = g->ensure_side_streams())) return rc;
            // blocks 1, 2: two temporaries each in the unused tails of the X and XT buffers (a stage-1 / stage-2 tensor is at most a quarter of a buffer)
            // (a chained stage needs one temporary per block, for the cut chains of blocks 1 and 2: the R and XT buffers, which it does not use otherwise)
            bst[1] = g->side[0];
            bXT[1] = chain_stage ? nullptr : X + n_i;
            bR[1] = chain_stage ? R : X + 2 * n_i;
            if (c.num_kernels > 2) {
                bst[2] = g->side[1];
                bXT[2] = chain_stage ? nullptr : XT + n_i;
                bR[2] = chain_stage ? XT : XT + 2 * n_i;
            }
            TTSC_HIP_CHECK(hipEventRecord(g->ev_fork, (hipStream_t)stream));   // the stage input (upsampler) is complete
            for (int j = 1; j < c.num_kernels; ++j) TTSC_HIP_CHECK(hipStreamWaitEvent((hipStream_t)bst[j], g->ev_fork, 0));
        }
        // block j's last launch adds into S: behind block j - 1's (same order of the sum as on one stream)
        auto before_acc = [&](int j) -> int {
            if (branch && j > 0) TTSC_HIP_CHECK(hipStreamWaitEvent((hipStream_t)bst[j], g->ev_acc[j - 1], 0));
            return TTSC_OK;
        };
        auto after_acc = [&](int j) -> int {
            if (branch) TTSC_HIP_CHECK(hipEventRecord(g->ev_acc[j], (hipStream_t)bst[j]));
            return TTSC_OK;
        };
        for (int j = 0; j < c.num_kernels; ++j) {
            const std::string rb = "resblocks." + std::to_string(i * c.num_kernels + j);
            const int nd = c.num_dilations[j];
            if (chain_rb[j]) {
                // the whole ResBlock1 in one launch: X -> S (+= for the second and third block)
                const ttsc_conv1d *c1[TTSC_HIFIGAN_MAX_DIL], *c2[TTSC_HIFIGAN_MAX_DIL];
                for (int m = 0; m < nd; ++m) {
                    c1[m] = layer(rb + ".convs1." + std::to_string(m));
                    c2[m] = layer(rb + ".convs2." + std::to_string(m));
                }
                if (g->fuse_post && i == c.num_upsamples - 1 && j == c.num_kernels - 1 && L % 4 == 0 && last_lens_mult4 &&
                    ((uintptr_t)X % 16 == 0) && ((uintptr_t)S % 16 == 0) && ttsc_rbchain_post_supported(c1, c2, nd, layer("conv_post"))) {
                    // the LAST block of the LAST stage: the block sum meets conv_post + tanh in the epilogue of this launch — S is read, never
                    // written, and the waveform is the only thing that leaves (one launch and two passes over the widest tensor less)
                    ttsc_conv1d_epilogue epost{inv_nk, 0.01f, 1.f, TTSC_ACT_TANH, 0};
                    if (branch && j > 0) {
                        // branch schedule: all pairs but the last beside the other blocks (-> this block's temporary), then — on the caller's stream, behind the
                        // block before — the last pair with conv_post in its epilogue
                        if (nd >= 2 && ttsc_rbchain_supported(c1, c2, nd - 1) && ttsc_rbchain_post_supported(c1 + nd - 1, c2 + nd - 1, 1, layer("conv_post")) &&
                            ((uintptr_t)bR[j] % 16 == 0)) {
                            rc = ttsc_rbchain_forward(c1, c2, nd - 1, X, B, L, bR[j], 0, ln, g->chain_shape, bst[j]);
                            if (rc) return rc;
                            TTSC_HIP_CHECK(hipEventRecord(g->ev_acc[j], (hipStream_t)bst[j]));
                            TTSC_HIP_CHECK(hipStreamWaitEvent((hipStream_t)stream, g->ev_acc[j - 1], 0));
                            TTSC_HIP_CHECK(hipStreamWaitEvent((hipStream_t)stream, g->ev_acc[j], 0));
                            return ttsc_rbchain_post_forward(c1 + nd - 1, c2 + nd - 1, 1, bR[j], B, L, S, layer("conv_post"), &epost, wav, ln, stream);
                        }
                        TTSC_HIP_CHECK(hipStreamWaitEvent((hipStream_t)stream, g->ev_acc[j - 1], 0));   // (the blocks before added into S on their own streams)
                    }
                    return ttsc_rbchain_post_forward(c1, c2, nd, X, B, L, j > 0 ? S : nullptr, layer("conv_post"), &epost, wav, ln, stream);
                }
                int nf = chain_first_pairs(g, ch, c.resblock_kernel_sizes[j], nd, B, L);
                if (branch && chain_stage) nf = (j > 0 && nd >= 2) ? nd - 1 : 0;   // (block 0 whole — R / XT belong to blocks 1 / 2; blocks 1, 2: only the last pair waits)
                if (nf > 0 && ttsc_rbchain_supported(c1, c2, nf) && ttsc_rbchain_supported(c1 + nf, c2 + nf, nd - nf)) {
                    rc = ttsc_rbchain_forward(c1, c2, nf, X, B, L, bR[j], 0, ln, g->chain_shape, bst[j]);
                    if (rc) return rc;
                    if ((rc = before_acc(j))) return rc;
                    rc = ttsc_rbchain_forward(c1 + nf, c2 + nf, nd - nf, bR[j], B, L, S, j > 0 ? 1 : 0, ln, g->chain_shape, bst[j]);
                } else {
                    if ((rc = before_acc(j))) return rc;
                    rc = ttsc_rbchain_forward(c1, c2, nd, X, B, L, S, j > 0 ? 1 : 0, ln, g->chain_shape, bst[j]);
                }
                if (rc) return rc;
                if ((rc = after_acc(j))) return rc;
                continue;
            }
            if (fused_stage) {
                // ONE fused launch per pair (inner activation stays in LDS); the residual stream ping-pongs between R and
                // XT because a fused tile reads its neighbours' halo of the input.
                const float* src = X;
                for (int m = 0; m < nd; ++m) {
                    const bool last = (m == nd - 1);
                    float* dst = last ? S : (src == R ? XT : R);
                    rc = ttsc_respair_forward(layer(rb + ".convs1." + std::to_string(m)), layer(rb + ".convs2." + std::to_string(m)), src, B,
                                              L, dst, (last && j > 0) ? 1 : 0, ln, stream);
                    if (rc) return rc;
                    src = dst;
                }
                continue;
            }
            for (int m = 0; m < nd; ++m) {
                float* const Rj = bR[j];
                float* const XTj = bXT[j];
                const float* src = (m == 0) ? X : Rj;
                const bool last = (m == nd - 1);
                float* dst = last ? S : Rj;
                ttsc_conv1d_epilogue e2{1.f, 0.1f, 1.f, TTSC_ACT_NONE, (last && j > 0) ? 1 : 0};
                if (c.resblock == 1) {
                    ttsc_conv1d_epilogue e1{1.f, 0.1f, 1.f, TTSC_ACT_NONE, 0};
                    rc = conv(layer(rb + ".convs1." + std::to_string(m)), src, L, XTj, nullptr, e1, ln, ln, 0, bst[j]);
                    if (rc) return rc;
                    if (last && (rc = before_acc(j))) return rc;
                    rc = conv(layer(rb + ".convs2." + std::to_string(m)), XTj, L, dst, src, e2, ln, ln, 0, bst[j]);   // + residual
                    if (rc) return rc;
                    if (last && (rc = after_acc(j))) return rc;
                } else {
                    // ResBlock2 reads src both as conv input and residual; dst != src unless m>0 && !last (R->R),
                    // where an in-place update would race with neighbouring tiles' halo reads -> ping-pong via XT.
                    float* d2 = dst;
                    if (dst == src) d2 = XT;
                    rc = conv(layer(rb + ".convs." + std::to_string(m)), src, L, d2, src, e2, ln, ln);
                    if (rc) return rc;
                    if (d2 != dst) std::swap(R, XT);
                }
            }
        }
        if (branch)   // the caller's stream goes on when every block has (block j's last launch waited for block j - 1's: the last event covers them all,
            for (int j = 1; j < c.num_kernels; ++j) TTSC_HIP_CHECK(hipStreamWaitEvent((hipStream_t)stream, g->ev_acc[j], 0));   // but each stream's own work needs its own)
        sum_scale = inv_nk;
    }
    ttsc_conv1d_epilogue epost{sum_scale, 0.01f, 1.f, TTSC_ACT_TANH, 0};
    return conv(layer("conv_post"), S, L, wav, nullptr, epost, lens[c.num_upsamples], lens[c.num_upsamples]);
}

