/*
 * ttscube_hip.h — C ABI of libttscube_hip.so: the MI355X (gfx950) waveform-synthesis hot path of
 * TTS-Cube, hand-written HIP behind plain pointers and sizes (no torch types).
 *
 * The reference (tiberiu44/TTS-Cube) is 100 % Python and has no FFI; the boundary it exposes for this
 * path is `nn.Module.forward/inference` + the `state_dict` key layout.  Each entry point
 * below names the reference interface it replaces; `ttscube_amd/` (Python, ctypes) mirrors the
 * reference classes on top of these symbols and INTEGRATION.md shows the reference-side binding.
 *
 * Conventions
 *   - every function returns TTSC_OK (0) or a negative TTSC_E* code; ttsc_last_error() returns a
 *     thread-local, human-readable message for the last failure on the calling thread.
 *   - `*_dev` pointers are device (HBM) pointers owned by the caller and valid for the call;
 *     host pointers are copied before the function returns.
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream); all work is
 *     enqueued on it, nothing synchronises the device unless stated.
 *   - handles are not thread-safe; distinct handles are independent.
 *   - activations are fp32, channel-major `[B, C, L]` contiguous (torch NCL), exactly what the
 *     reference modules exchange.
 */
#ifndef TTSCUBE_HIP_H
#define TTSCUBE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TTSC_OK 0
#define TTSC_EINVAL (-1)   /* bad argument / shape / name */
#define TTSC_EHIP (-2)     /* HIP runtime error (message has hipGetErrorString) */
#define TTSC_ESTATE (-3)   /* weights missing / handle not ready */
#define TTSC_ENOMEM (-4)   /* workspace too small */
#define TTSC_ERANGE (-5)   /* split-precision generator: non-finite output even after re-calibration (non-finite input / weights) */

const char* ttsc_version(void);
const char* ttsc_last_error(void);
/* number of visible HIP devices, or negative error (used by the loader to fail loudly) */
int ttsc_device_count(void);

/* ------------------------------------------------------------------------------------------------
 * Conv1d / ConvTranspose1d layer (fp32 MFMA implicit GEMM).
 * Replaces torch.nn.Conv1d / ConvTranspose1d as used by cube/networks/modules.py:37-55 (ConvNorm),
 * modules.py:117-145 (PostNet), modules.py:416-420 (WaveRNN low-res convs), textcoder.py:44-53 /
 * modules.py:850-871 (char CNN) and the [EXTERNAL] hifigan Generator convs (cubegan.py:43).
 * ------------------------------------------------------------------------------------------------ */
typedef struct ttsc_conv1d ttsc_conv1d;

typedef struct {
    int32_t in_channels;
    int32_t out_channels;
    int32_t kernel_size;
    int32_t stride;      /* Conv1d: must be 1.  ConvTranspose1d: upsample factor */
    int32_t padding;
    int32_t dilation;    /* ConvTranspose1d: must be 1 */
    int32_t transposed;  /* 0 = Conv1d (weight [Cout,Cin,K]), 1 = ConvTranspose1d (weight [Cin,Cout,K]) */
    int32_t groups;      /* Conv1d only: torch's `groups` (weight [Cout, Cin/groups, K]); 0 or 1 = dense.  TTSC_PREC_FP32 only — the
                          * grouped k=41 layers of the multi-scale discriminator (training, row f1) */
} ttsc_conv1d_cfg;

enum { TTSC_ACT_NONE = 0, TTSC_ACT_TANH = 1, TTSC_ACT_RELU = 2, TTSC_ACT_SIGMOID = 3 };
/* arithmetic of the implicit GEMM: exact fp32 MFMA (k-ordered fmaf chain), or split precision — every fp32 value as
 * fp16 hi + lo halves, three fp16 MFMAs per product with fp32 accumulation (~2^-21 relative, 5.3x the MFMA rate;
 * activations must stay inside fp16 range |x| < 65504). */
enum { TTSC_PREC_FP32 = 0, TTSC_PREC_F16X3 = 1 };

typedef struct {
    float in_scale;      /* x is staged as leaky_relu(x * in_scale, in_slope); 1 / 1 = identity */
    float in_slope;
    float out_scale;     /* y = act((conv + bias + resid) * out_scale)  */
    int32_t out_act;     /* TTSC_ACT_* */
    int32_t accumulate;  /* 1: y += result (running sum of residual blocks) */
    /* data-gradient launches (training, row a9): gate_dev = the pre-activation [B,Cout,Lout] the forward layer read through its
     * leaky-relu prologue; then y = ((conv + bias) * (gate > 0 ? 1 : gate_slope) + resid) * out_scale.  NULL = off. */
    const float* gate_dev;
    float gate_slope;
} ttsc_conv1d_epilogue;

int ttsc_conv1d_create(const ttsc_conv1d_cfg* cfg, ttsc_conv1d** out);
/* weight: host fp32 in torch layout (already weight-norm folded); bias: host [Cout] or NULL */
int ttsc_conv1d_set_weight(ttsc_conv1d* c, const float* weight_host, const float* bias_host);
/* training: (re)pack the weights from DEVICE memory (torch parameter layout, fp32) on `stream`, no host round trip;
 * TTSC_PREC_FP32 handles only.  bias_dev NULL = no bias. */
int ttsc_conv1d_set_weight_device(ttsc_conv1d* c, const float* weight_dev, const float* bias_dev, void* stream);
/* same, reading the weight of the FORWARD layer this handle is the data gradient of: a Conv1d(Ci -> Co, K) forward weight
 * [Co,Ci,K] is packed as the flipped, transposed Conv1d(Co -> Ci, K) weight  W'[ci,co,k] = W[co,ci,K-1-k]. */
int ttsc_conv1d_set_weight_device_dgrad(ttsc_conv1d* c, const float* fwd_weight_dev, void* stream);
/* switch the arithmetic (TTSC_PREC_*); weights are re-packed from the host copy kept by set_weight */
int ttsc_conv1d_set_precision(ttsc_conv1d* c, int32_t precision);
int64_t ttsc_conv1d_out_len(const ttsc_conv1d* c, int64_t Lin);
int32_t ttsc_conv1d_in_channels(const ttsc_conv1d* c);
/* TTSC_PREC_F16X3 carries activations as fp16 (hi, lo) pairs: below |x| ~ 2^-3 the low half goes subnormal (accuracy decays
 * towards fp16's), above 65504 the high half overflows.  `scale` (a power of two) multiplies this layer's INPUT while it is
 * staged and is divided out of the accumulator in the epilogue — both exact; 1 = off.  Ignored by TTSC_PREC_FP32. */
int ttsc_conv1d_set_activation_scale(ttsc_conv1d* c, float scale);
float ttsc_conv1d_get_activation_scale(const ttsc_conv1d* c);
/* out_channels == 1 layers (conv_post): `flag_dev` (device, 4 bytes, or NULL = off) is OR-ed with 1 by the forward kernel when it emits a
 * non-finite sample — the range guard of the split-precision generator (ttsc_hifigan_forward). */
int ttsc_conv1d_set_nonfinite_flag(ttsc_conv1d* c, uint32_t* flag_dev);
/* out_dev[0] = max(out_dev[0], max_i |x_i|) over n device floats (out_dev zeroed by the caller); feeds the calibration below */
int ttsc_absmax(const float* x_dev, int64_t n, float* out_dev, void* stream);
/* x_dev [B,Cin,Lin] -> y_dev [B,Cout,Lout]; resid_dev NULL or [B,Cout,Lout]; ep NULL = plain conv */
int ttsc_conv1d_forward(const ttsc_conv1d* c, const float* x_dev, int32_t B, int64_t Lin, float* y_dev,
                        const float* resid_dev, const ttsc_conv1d_epilogue* ep, void* stream);
/* ragged batch: in_len_dev / out_len_dev int32 [B] (device) give each utterance's valid input / output length inside
 * the padded [B,C,L] tensors; input beyond in_len[b] reads as zero (== the utterance run alone), tiles wholly beyond
 * out_len[b] are skipped (their output is unspecified).  NULL = dense. */
int ttsc_conv1d_forward_ragged(const ttsc_conv1d* c, const float* x_dev, int32_t B, int64_t Lin, float* y_dev,
                               const float* resid_dev, const ttsc_conv1d_epilogue* ep, const int32_t* in_len_dev,
                               const int32_t* out_len_dev, void* stream);
/* The same with an explicit ROW PITCH of the output tensors: y_dev / resid_dev are [B,Cout,Lout_pitch] (Lout_pitch = 0: the natural
 * output length of Lin).  With per-utterance lengths in in_len_dev / out_len_dev, Lin and Lout_pitch are pitches only — that is how
 * ttsc_hifigan_forward keeps the rows of its intermediate tensors on 128-byte boundaries (4001 / 12004 samples per row otherwise: every
 * 128-byte store of the 256- and 128-channel stages straddled two lines).  Lout_pitch must cover every utterance's real output length. 
 * Rounding note (TTSC_PREC_F16X3, square 128- / 256-channel layers on the wide-tile kernel, out_scale == 1, no activation): by default the
 * residual, the running sum and the bias are the accumulators' INITIAL value (TTSC_CONV_ACC_INIT=1), so the K * Cin / 16 accumulation steps
 * round at the magnitude of the whole sum rather than of the convolution alone (self-check RMS against the fp32 oracle 6.4e-7 instead of 4.3e-7
 * for the whole generator; per-layer tolerance in tests 3e-6 relative).  TTSC_CONV_ACC_INIT=0 adds them after the sum (prefetched epilogue; the
 * bias still starts the sum there, so the bits differ from the general kernel's, which adds the bias last).  A layer's bits therefore depend on
 * which kernel the machine-fill rule selects; results are deterministic for a given (shape, batch, environment). */
int ttsc_conv1d_forward_pitched(const ttsc_conv1d* c, const float* x_dev, int32_t B, int64_t Lin, float* y_dev,
                                const float* resid_dev, const ttsc_conv1d_epilogue* ep, const int32_t* in_len_dev,
                                const int32_t* out_len_dev, int64_t Lout_pitch, void* stream);
/* Fused residual pair  y = x + conv2(lrelu(conv1(lrelu(x), 0.1), 0.1)) [+ y if accumulate]  of a HiFi-GAN ResBlock1
 * (both layers 32 -> 32 channels, odd kernel 3/7/11, conv2 undilated, both in TTSC_PREC_F16X3): the inner activation
 * stays in LDS, which removes two of the five HBM passes of the unfused pair.  `supported` returns 1 when the fused
 * kernel applies to the two layers; y must not alias x. */
int ttsc_respair_supported(const ttsc_conv1d* conv1, const ttsc_conv1d* conv2);
int ttsc_respair_forward(const ttsc_conv1d* conv1, const ttsc_conv1d* conv2, const float* x_dev, int32_t B, int64_t L,
                         float* y_dev, int32_t accumulate, const int32_t* len_dev, void* stream);
/* Fused ResBlock1 chain  x <- x + conv2_p(lrelu(conv1_p(lrelu(x), 0.1), 0.1)),  p = 0..npairs-1 (npairs <= 3), y (+)= x:
 * hifigan.models.ResBlock1.forward [EXTERNAL; call sites cube/networks/cubegan.py:72,83 via Generator.forward] for the 32- and
 * 64-channel stages, where every single convolution is HBM-bound.  The residual stream of a time tile stays in registers and
 * the activations in LDS for the whole chain; HBM sees one read of x and one write (read-modify-write when `accumulate`) of y.
 * All layers C -> C with C in {32, 64}, one odd kernel size K in {3, 7, 11}, conv1 dilation <= 5, conv2 undilated, "same"
 * padding, TTSC_PREC_F16X3, host-set weights with bias.  `supported` returns 1 when the fused kernel applies.
 * tile_shape: -1 = pick by halo, 0 = small tile (512 columns at C=32 / 256 at C=64; two workgroups per CU), 1 = large tile
 * (1024 / 512 columns, one 8-wave workgroup per CU); 0 and 1 run with interleaved columns (a lane's column tiles are consecutive
 * samples: the tile moves as dwordx4 / dwordx2 accesses; bit-identical results) when the dilations are in {1, 3, 5} unless the
 * environment says TTSC_CHAIN_IL=0; 10 / 11 ask for the interleaved kernels explicitly, 12 = 11 with 6-step weight groups (C = 32),
 * 2 .. 4 = measurement variants (longer weight groups, 768-column tiles).  y must not alias x. */
int ttsc_rbchain_supported(const ttsc_conv1d* const* convs1, const ttsc_conv1d* const* convs2, int32_t npairs);
int ttsc_rbchain_forward(const ttsc_conv1d* const* convs1, const ttsc_conv1d* const* convs2, int32_t npairs, const float* x_dev,
                         int32_t B, int64_t L, float* y_dev, int32_t accumulate, const int32_t* len_dev, int32_t tile_shape,
                         void* stream);
/* The last ResBlock1 chain of the generator with conv_post + activation in its epilogue:
 *     wav = act((conv_post(lrelu((ysum + chain(x)) * in_scale, in_slope)) + bias) * out_scale)         (scales / activation of post_ep)
 * — `xs += resblocks[-1](x); x = xs / nk; x = conv_post(lrelu(x)); tanh` of hifigan.models.Generator.forward [EXTERNAL; call sites
 * cube/networks/cubegan.py:72,83,131, cube/io_utils/runtime.py:78] in one launch: ysum_dev [B,32,L] (the sum of the blocks before this one, or NULL)
 * is only read, wav_dev [B,1,L] is the only tensor written, conv_post's range-guard word is honoured; bit-identical to
 * ttsc_rbchain_forward(accumulate) + ttsc_conv1d_forward.  32 channels, dilations in {1, 3, 5}, conv_post 32 -> 1 with k = 7 and padding 3;
 * L and every entry of len_dev multiples of 4, x / ysum 16-byte aligned.  ttsc_hifigan_forward uses it unless TTSC_HIFIGAN_FUSE_POST=0. */
int ttsc_rbchain_post_supported(const ttsc_conv1d* const* convs1, const ttsc_conv1d* const* convs2, int32_t npairs, const ttsc_conv1d* post);
int ttsc_rbchain_post_forward(const ttsc_conv1d* const* convs1, const ttsc_conv1d* const* convs2, int32_t npairs, const float* x_dev, int32_t B,
                              int64_t L, const float* ysum_dev, const ttsc_conv1d* post, const ttsc_conv1d_epilogue* post_ep, float* wav_dev,
                              const int32_t* len_dev, void* stream);
void ttsc_conv1d_destroy(ttsc_conv1d* c);

/* Weight gradient of the generator's convolutions (training: `Cubegan.training_step`, cube/networks/cubegan.py:85-189,
 * where torch autograd differentiates Generator.forward):
 *   G[a, b, j] += sum_n sum_t P[n,a,t] * leaky_relu(q_scale * Q[n,b,t + base + j*step], q_slope)       (zero outside [0,LQ))
 * Conv1d(weight [Co,Ci,K], dilation d, padding p): P = dL/dy [N,Co,Lout], Q = layer input [N,Ci,Lin], base = -p, step = d,
 * G = dL/dW [Co,Ci,K].  G (device, [A,B,J] fp32) is overwritten.  The position axis is split over ~2048 waves whose partial
 * tiles go through `ws_dev` (>= ttsc_conv_wgrad_workspace_bytes) and are added in a fixed order (deterministic).
 * |(J-1)*step| <= 64 per group of 12 taps. */
/* torch.nn.utils.weight_norm(dim=0) of the generator convolutions [EXTERNAL hifigan/models.py], fused: v [rows, cols] (cols =
 * product of the other dims), g [rows];  forward: w = v * g / ||v||_row, norm = ||v||_row;  backward: dv, dg from dw. */
int ttsc_weight_norm_forward(const float* v_dev, const float* g_dev, float* w_dev, float* norm_dev, int32_t rows, int64_t cols, void* stream);
int ttsc_weight_norm_backward(const float* dw_dev, const float* v_dev, const float* g_dev, const float* norm_dev, float* dv_dev,
                              float* dg_dev, int32_t rows, int64_t cols, void* stream);
/* torch.nn.utils.spectral_norm of the first scale discriminator ([EXTERNAL hifigan/models.py] MultiScaleDiscriminator:
 * DiscriminatorS(use_spectral_norm=True), used by cube/networks/cubegan.py:40-41): wn = W / sigma, sigma = u^T W v after one power iteration (v <- normalize(W^T u), u <- normalize(W v)).  The two
 * mat-vecs and the pieces around them, all with fixed summation orders:
 *   ttsc_matvec         W [rows, cols] row-major: out = W x (transpose 0) or W^T x (transpose 1; ws_dev >= ttsc_matvec_workspace_bytes)
 *   ttsc_l2_normalize   out[n] = x / max(||x||, eps), norm_dev[0] = ||x|| (either output may be NULL)
 *   ttsc_dot            out_dev[0] = sum a[i] b[i] (ws_dev >= ttsc_dot_workspace_bytes(n))
 *   ttsc_div_scalar     out = w / sigma_dev[0]
 *   ttsc_spectral_norm_backward   dW[r,c] = dWn[r,c] / sigma - (dot_dev[0] / sigma^2) u[r] v[c], dot_dev[0] = sum(dWn . W) */
size_t ttsc_matvec_workspace_bytes(int32_t rows, int64_t cols);
int ttsc_matvec(const float* w_dev, int32_t rows, int64_t cols, const float* x_dev, int32_t transpose, float* out_dev, void* ws_dev, size_t ws_bytes,
                void* stream);
int ttsc_l2_normalize(const float* x_dev, int32_t n, float eps, float* out_dev, float* norm_dev, void* stream);
size_t ttsc_dot_workspace_bytes(int64_t n);
int ttsc_dot(const float* a_dev, const float* b_dev, int64_t n, float* out_dev, void* ws_dev, size_t ws_bytes, void* stream);
int ttsc_div_scalar(const float* w_dev, const float* sigma_dev, float* out_dev, int64_t n, void* stream);
int ttsc_spectral_norm_backward(const float* dwn_dev, const float* u_dev, const float* v_dev, const float* sigma_dev, const float* dot_dev,
                                float* dw_dev, int32_t rows, int64_t cols, void* stream);
/* bias gradient db[c] = sum_{b,t} dy[b,c,t] (fixed summation order).  ws_is_fresh = 1 when the workspace was not left by a
 * previous ttsc_bias_grad call with the same shape (its ticket counters are then zeroed on the stream first). */
size_t ttsc_bias_grad_workspace_bytes(int32_t B, int32_t C, int64_t L);
int ttsc_bias_grad(const float* dy_dev, float* db_dev, int32_t B, int32_t C, int64_t L, void* ws_dev, size_t ws_bytes, int32_t ws_is_fresh,
                   void* stream);
/* torch.optim.AdamW (cube/networks/cubegan.py:275-298: betas (0.8, 0.99)) over ONE flat fp32 arena: parameters, gradients and both
 * moment estimates of a parameter group are four contiguous, 16-byte-aligned device arrays of n elements (the gradient arena is the
 * gradient-exchange bucket itself); `step` is the 1-based step count of the bias corrections.  Same update rule and operation order
 * as torch's single-tensor AdamW (decoupled weight decay, no amsgrad). */
int ttsc_adamw_step(float* p_dev, const float* g_dev, float* m_dev, float* v_dev, int64_t n, float lr, float beta1, float beta2, float eps,
                    float weight_decay, int64_t step, void* stream);
/* The same update behind a device-side guard: when *guard_dev != 0 at execution time the launch leaves parameters and moments untouched (null = no
 * guard).  The word is what ttsc_split_status_collect left on the stream: a training step can then queue its update without the host waiting for the
 * backward pass to learn whether a split recurrence gave up (cube/networks/cubegan.py:172-180 is the update; the reference has no such failure mode). */
int ttsc_adamw_step_guarded(float* p_dev, const float* g_dev, float* m_dev, float* v_dev, int64_t n, float lr, float beta1, float beta2, float eps,
                            float weight_decay, int64_t step, const uint32_t* guard_dev, void* stream);
/* GAN loss terms over a LIST of tensors in one launch, value and gradient together — hifigan.models.feature_loss / generator_loss /
 * discriminator_loss [EXTERNAL; call sites cube/networks/cubegan.py:144-149,160-167]:
 *   kind 0:  out = sum_k w_k * mean|a_k - b_k|      gb_k = w_k sign(b_k - a_k) / n_k,  ga_k = -gb_k      (feature matching: w = 2)
 *   kind 1:  out = sum_k w_k * mean (a_k - target)^2   ga_k = 2 w_k (a_k - target) / n_k              (least-squares GAN terms)
 * a_dev / b_dev / ga_dev / gb_dev: host arrays of nseg device pointers (gradient pointers, or the arrays themselves, may be null);
 * numel / weight: host arrays.  1 <= nseg <= 64 per call.  out_dev: one float.  Deterministic (fixed-order sums). */
/* nn.Embedding of the mel decoder's text stacks (cube/networks/modules.py:869-872) in training: out[i,:] = table[idx[i],:] (rows outside
 * [0,V) read as zero) and its gradient gtable[v,:] = sum_{i: idx[i]==v} gout[i,:] (every row written; row `skip_row` — padding_idx — gets
 * zeros; -1 = none).  Deterministic summation order. */
int ttsc_rows_gather(const float* table_dev, const int32_t* idx_dev, float* out_dev, int64_t n, int32_t C, int32_t V, void* stream);
int ttsc_rows_scatter_add(const float* gout_dev, const int32_t* idx_dev, float* gtable_dev, int64_t n, int32_t C, int32_t V, int32_t skip_row,
                          void* stream);
/* the same adjoint for a NON-DECREASING index list (phoneme rows -> frame rows, cube/networks/modules.py:1043-1053): O(n C) instead of O(V n C) */
int ttsc_rows_segment_sum(const float* gout_dev, const int32_t* idx_sorted_dev, float* gtable_dev, int64_t n, int32_t C, int32_t V, void* stream);
/* Polyphase de-interleave of a strided Conv1d's operands (the discriminators' stride-2/3/4 layers [EXTERNAL hifigan/models.py DiscriminatorP /
 * DiscriminatorS; call sites cube/networks/cubegan.py:144-149,160-167]): the layer runs as a stride-1 convolution over
 *   xr[n, (g, r, ci), m P + w] = x[n, (g, ci), ((m s + r) - pad) P + w]   (zero outside; P = 1 or MPD's period, rows of P samples)
 * with taps  wp[co, (r, ci), j] = w[co, ci, s j + r]  (zero beyond K).  backward = 1 maps a gradient of xr / wp back onto x / w.
 * x [N, C, L P], xr [N, s C, M P]; w [Cout, Cg, K], wp [Cout, s Cg, ceil(K / s)]. */
int ttsc_deinterleave_x(const float* src_dev, float* dst_dev, int32_t N, int32_t C, int64_t L, int32_t groups, int32_t stride, int32_t period, int32_t pad,
                        int32_t M, int32_t backward, void* stream);
int ttsc_deinterleave_w(const float* src_dev, float* dst_dev, int32_t Cout, int32_t Cg, int32_t K, int32_t stride, int32_t backward, void* stream);
size_t ttsc_gan_loss_workspace_bytes(int32_t nseg);
int ttsc_gan_loss(int32_t kind, int32_t nseg, const void* const* a_dev, const void* const* b_dev, void* const* ga_dev, void* const* gb_dev,
                  const int64_t* numel, const float* weight, float target, float* out_dev, void* ws_dev, size_t ws_bytes, void* stream);
/* kind 0 over PRE-activations: slope[k] in [0, 1] (null = all 1) passes both operands of segment k through leaky_relu(., slope[k]) before the
 * difference and multiplies the gradients by the activation's derivative — the discriminators' feature maps are leaky_relu(conv output, 0.1)
 * ([EXTERNAL hifigan/models.py DiscriminatorP / DiscriminatorS forward]; feature_loss call sites cube/networks/cubegan.py:162-163): the training
 * step hands the convolution outputs over as they are instead of materialising ~90 activated copies and their backward launches per step. */
int ttsc_gan_loss_lrelu(int32_t kind, int32_t nseg, const void* const* a_dev, const void* const* b_dev, void* const* ga_dev, void* const* gb_dev,
                        const int64_t* numel, const float* weight, const float* slope, float target, float* out_dev, void* ws_dev, size_t ws_bytes,
                        void* stream);
size_t ttsc_conv_wgrad_workspace_bytes(int32_t N, int32_t A, int32_t B, int64_t LP, int32_t J);
/* grouped variant (torch Conv1d groups): P [N,A,LP], Q [N,groups*Bg,LQ], G [A,Bg,J] — row a only meets the Bg channels of its own
 * group.  Workspace: ttsc_conv_wgrad_workspace_bytes(N, A, Bg, LP, J). */
int ttsc_conv_wgrad_grouped(const float* p_dev, const float* q_dev, float* g_dev, int32_t N, int32_t A, int32_t Bg, int32_t groups, int64_t LP,
                            int64_t LQ, int32_t J, int32_t base, int32_t step, float q_scale, float q_slope, void* ws_dev, size_t ws_bytes,
                            void* stream);
int ttsc_conv_wgrad(const float* p_dev, const float* q_dev, float* g_dev, int32_t N, int32_t A, int32_t B, int64_t LP, int64_t LQ,
                    int32_t J, int32_t base, int32_t step, float q_scale, float q_slope, void* ws_dev, size_t ws_bytes, void* stream);

/* Split-precision weight gradient (csrc/conv_wgrad.hip::wgrad_f16x3_kernel): same contract as ttsc_conv_wgrad for the dense layers of the
 * training step (cubegan.py:137-170: every parameter gradient of MPD / MSD and of the generator), operands carried as fp16 hi + lo with
 * device-side ranges per launch, 128 x 64 tiles on v_mfma_f32_32x32x16_f16; agrees with ttsc_conv_wgrad to ~1e-6 of the largest entry.
 * `ttsc_conv_wgrad_split_supported` says whether a shape is taken (A >= 64 rows, Bc >= 32 columns, J <= 16 taps).  Range words as for
 * ttsc_conv_train: `measure` bit 0 = reduce max |Q| now, bit 1 = max |P| now, bit 2 = the caller's words were zeroed by the caller (pooled
 * words: no memset launch); null pointers = workspace words measured by this call. */
int32_t ttsc_conv_wgrad_split_supported(int32_t A, int32_t B, int32_t J, int32_t step);
size_t ttsc_conv_wgrad_split_workspace_bytes(int32_t N, int32_t A, int32_t B, int64_t LP, int32_t J);
int ttsc_conv_wgrad_split(const float* p_dev, const float* q_dev, float* g_dev, int32_t N, int32_t A, int32_t B, int64_t LP, int64_t LQ, int32_t J,
                          int32_t base, int32_t step, float q_scale, float q_slope, float* amax_q_dev, float* amax_p_dev, int32_t measure, void* ws_dev,
                          size_t ws_bytes, void* stream);
/* Grouped layers (MSD's k = 41 convolutions: groups of 16 .. 64 output channels) on the same split-precision scheme
 * (csrc/conv_wgrad.hip::wgrad_f16x3_grouped_kernel): p [N][A][LP], q [N][groups * Bg][LQ] -> g [A][Bg][J], group g = rows [g A / groups, (g + 1) A / groups)
 * against q's channels [g Bg, (g + 1) Bg).  Range words and `measure` as for ttsc_conv_wgrad_split. */
int32_t ttsc_conv_wgrad_split_grouped_supported(int32_t A, int32_t Bg, int32_t groups, int32_t J, int32_t step);
size_t ttsc_conv_wgrad_split_grouped_workspace_bytes(int32_t N, int32_t A, int32_t Bg, int32_t groups, int64_t LP, int32_t J);
int ttsc_conv_wgrad_split_grouped(const float* p_dev, const float* q_dev, float* g_dev, int32_t N, int32_t A, int32_t Bg, int32_t groups, int64_t LP, int64_t LQ,
                                  int32_t J, int32_t base, int32_t step, float q_scale, float q_slope, float* amax_q_dev, float* amax_p_dev, int32_t measure,
                                  void* ws_dev, size_t ws_bytes, void* stream);
/* The same call that also leaves the layer's bias gradient db[a] = sum_{n,t} p[n,a,t] in db_dev [A] (NULL = plain ttsc_conv_wgrad_split): its
 * slices are summed by surplus workgroups of the first weight-gradient launch and added in index order by surplus workgroups of the reduction —
 * the bits of ttsc_bias_grad, no launch of its own (same workspace size). */
int ttsc_conv_wgrad_split_bias(const float* p_dev, const float* q_dev, float* g_dev, float* db_dev, int32_t N, int32_t A, int32_t B, int64_t LP, int64_t LQ,
                               int32_t J, int32_t base, int32_t step, float q_scale, float q_slope, float* amax_q_dev, float* amax_p_dev, int32_t measure,
                               void* ws_dev, size_t ws_bytes, void* stream);

/* Split-precision convolution of the training step (csrc/conv_train.hip): forward and data gradient of a dense, stride-1, dilated Conv1d
 * of the generator [EXTERNAL hifigan/models.py; trained by cube/networks/cubegan.py:131-170] and of the MPD / MSD discriminators
 * (cubegan.py:144-149,160-167) on the fp16 hi/lo three-product MFMA path, stateless: weights come straight from the live torch parameter.
 *   y[b,co,t] = (( sum_ci sum_k W[co,ci,k] * lrelu(in_scale * x[b,ci,t - padding + k*dilation], in_slope) + bias[co] ) * g + resid[b,co,t]) * out_scale,
 *   g = 1 without `gate_dev`, else (gate[b,co,t] > 0 ? 1 : gate_slope)    (the leaky-relu derivative of a data-gradient launch),
 *   W = w_dev [Cout][Cin][K] (flip = 0), or W[co,ci,k] = w_dev[ci][co][K-1-k] with w_dev [Cin][Cout][K] (flip = 1: `w_dev` is the forward
 *   weight of the layer being differentiated).  x [B,Cin,Lin], y / resid / gate [B,Cout,Lout], Lout = Lin + 2 padding - dilation (K-1).
 *   groups > 1 (MSD's grouped k = 41 layers): torch Conv1d `groups`; w_dev [Cout][Cin/groups][K] (flip = 1: [Cin][Cout/groups][K]).
 * Both fp16 ranges are set per launch from device-side maxima of x and w (no host synchronisation, no calibration state); results agree
 * with the fp32 kernel to ~1e-6 relative.  The two range words (one float each, device) can be shared by the launches of one layer so that every
 * tensor is reduced once per step: forward measures max |x| and max |w| (`measure` = 3), the data gradient re-uses the weight word and measures
 * max |dy| (`measure` = 1), the weight gradient re-uses both; null pointers = measured into the workspace by this call; `measure` bit 2 = the words to be
 * measured were zeroed by the caller (words of a pool zeroed once per step: no memset launch).
 * `ttsc_conv_train_supported` says whether a shape is taken (receptive fields beyond 64 positions,
 * more than 41 taps and group sizes that do not tile into 32-row blocks stay on ttsc_conv1d_forward); the workspace holds the two range words and the packed weight fragments. */
int32_t ttsc_conv_train_supported(int32_t Cin, int32_t Cout, int32_t K, int32_t dilation, int32_t groups);
size_t ttsc_conv_train_workspace_bytes(int32_t Cin, int32_t Cout, int32_t K, int32_t groups);
int ttsc_conv_train(const float* x_dev, const float* w_dev, const float* bias_dev, const float* resid_dev, const float* gate_dev, float* y_dev,
                    int32_t B, int32_t Cin, int32_t Cout, int32_t K, int64_t Lin, int32_t padding, int32_t dilation, int32_t groups, int32_t flip, float in_scale,
                    float in_slope, float out_scale, float gate_slope, float* amax_x_dev, float* amax_w_dev, int32_t measure, void* ws_dev, size_t ws_bytes,
                    void* stream);

/* The same convolution on weight fragments prepared by a weight bank (below): no weight reduction, no packing launch, no workspace.
 * wfrag_dev / amax_w_dev: the bank entry's `pack_fwd` (forward) or `pack_dgrad` (data gradient: pass the differentiated layer's Cout as Cin and
 * vice versa, as for flip = 1) and its `amax` word.  `measure` bit 0 = reduce max |x| into *amax_x_dev now (clear: the word already holds it — e.g.
 * the `amax_y_dev` word of the launch that produced x); bit 2 = the caller zeroed that word (words handed out from a pool zeroed once per step: no
 * memset launch).  amax_y_dev (or NULL): a ZEROED word that receives max |y| over the stored elements from the convolution's own epilogue, so that
 * the launches that read y next (the following layer, the preceding layer's data gradient, the weight gradient) need no reduction launch over it. */
int ttsc_conv_train_packed(const float* x_dev, const void* wfrag_dev, const float* bias_dev, const float* resid_dev, const float* gate_dev, float* y_dev,
                           int32_t B, int32_t Cin, int32_t Cout, int32_t K, int64_t Lin, int32_t padding, int32_t dilation, int32_t groups,
                           float in_scale, float in_slope, float out_scale, float gate_slope, float* amax_x_dev, const float* amax_w_dev,
                           int32_t measure, float* amax_y_dev, void* stream);

/* Weight bank: the per-step weight preparation of ALL convolutions of one module in three launches (csrc/conv_train.hip) — what the reference
 * leaves to torch.nn.utils.weight_norm's pre-forward hooks and to the cuDNN / MIOpen filter transforms inside every nn.Conv1d / nn.Conv2d call of
 * Generator / MultiPeriodDiscriminator / MultiScaleDiscriminator [EXTERNAL hifigan/models.py; cubegan.py:131,144-149,160-167].
 * Per entry (all pointers device memory owned by the caller, valid until ttsc_wbank_destroy):
 *   v [Cout][Cin/groups][K] fp32 and g [Cout] (weight_v / weight_g; g NULL: v is the plain weight);
 *   w (same shape as v) and norm [Cout]: receive g * v / ||v|| and the row norms (ttsc_weight_norm_forward's outputs; unused when g is NULL);
 *   amax: one float, receives max |w|;  pack_fwd / pack_dgrad: ttsc_conv_train_workspace_bytes(stride * Cin, Cout, ceil(K / stride), groups) - 256
 *   bytes (resp. with the two channel counts swapped), either may be NULL;
 *   stride > 1: the layer is a strided convolution evaluated as a stride-1 convolution over the phase-de-interleaved input
 *   (ttsc_deinterleave_x); the fragments hold w'[co][(r, ci)][j] = w[co][ci][stride * j + r] (zero beyond K) = ttsc_deinterleave_w's output.
 * ttsc_wbank_prepare re-reads v / g and refills every output (call it after the parameters changed, on the stream the convolutions follow on). */
typedef struct ttsc_wbank ttsc_wbank;
typedef struct ttsc_wbank_entry {
    const float* v;
    const float* g;
    float* w;
    float* norm;
    float* amax;
    void* pack_fwd;
    void* pack_dgrad;
    int32_t Cin, Cout, K, groups, stride;
} ttsc_wbank_entry;
int ttsc_wbank_create(const ttsc_wbank_entry* entries_host, int32_t n, ttsc_wbank** out);
int ttsc_wbank_prepare(ttsc_wbank* bank, void* stream);
void ttsc_wbank_destroy(ttsc_wbank* bank);

/* ------------------------------------------------------------------------------------------------
 * HiFi-GAN generator.  Replaces `hifigan.models.Generator(h)` [EXTERNAL submodule]:
 * constructed cube/networks/cubegan.py:41-43 and cube/io_utils/runtime.py:49-51,
 * called cubegan.py:72,83,131,234 and runtime.py:78 as generator(mel[B,80,T]) -> [B,1,L].
 * ------------------------------------------------------------------------------------------------ */
typedef struct ttsc_hifigan ttsc_hifigan;

#define TTSC_HIFIGAN_MAX_UPS 8
#define TTSC_HIFIGAN_MAX_RB 8
#define TTSC_HIFIGAN_MAX_DIL 8

typedef struct {
    int32_t num_mels;                 /* 80 */
    int32_t upsample_initial_channel; /* 512 */
    int32_t resblock;                 /* 1 or 2 (config key "resblock") */
    int32_t num_upsamples;
    int32_t upsample_rates[TTSC_HIFIGAN_MAX_UPS];
    int32_t upsample_kernel_sizes[TTSC_HIFIGAN_MAX_UPS];
    int32_t num_kernels;
    int32_t resblock_kernel_sizes[TTSC_HIFIGAN_MAX_RB];
    int32_t num_dilations[TTSC_HIFIGAN_MAX_RB];
    int32_t resblock_dilation_sizes[TTSC_HIFIGAN_MAX_RB][TTSC_HIFIGAN_MAX_DIL];
} ttsc_hifigan_cfg;

int ttsc_hifigan_create(const ttsc_hifigan_cfg* cfg, ttsc_hifigan** out);
/* name = folded state_dict key: "conv_pre.weight", "ups.0.bias", "resblocks.3.convs1.2.weight",
 * "conv_post.weight" ...; host fp32, torch layout, shape checked.  The host copy is kept and
 * packed/uploaded lazily by the next forward, so weights may be updated any number of times. */
int ttsc_hifigan_set_weight(ttsc_hifigan* g, const char* name, const float* host, const int64_t* shape, int32_t nd);
/* TTSC_PREC_* for every conv of the generator (default: TTSC_PREC_F16X3, overridable with env TTSC_HIFIGAN_PRECISION=fp32) */
int ttsc_hifigan_set_precision(ttsc_hifigan* g, int32_t precision);
int64_t ttsc_hifigan_out_len(const ttsc_hifigan* g, int64_t T);
size_t ttsc_hifigan_workspace_bytes(const ttsc_hifigan* g, int32_t B, int64_t T);
/* mel_dev [B,num_mels,T] -> wav_dev [B,1,out_len(T)] (tanh output in (-1,1)) */
int ttsc_hifigan_forward(ttsc_hifigan* g, const float* mel_dev, int32_t B, int64_t T, float* wav_dev,
                         void* workspace_dev, size_t workspace_bytes, void* stream);
/* ragged batch: frames_host int32 [B] (HOST pointer, copied) = valid mel frames per utterance (<= T).  Every layer masks
 * its input beyond the utterance's own length, so wav_dev[b, 0, :out_len(frames[b])] equals that utterance run alone
 * and padding-only tiles are skipped.  The workspace must hold ttsc_hifigan_workspace_bytes(g,B,T) bytes. */
int ttsc_hifigan_forward_ragged(ttsc_hifigan* g, const float* mel_dev, int32_t B, int64_t T, const int32_t* frames_host,
                                float* wav_dev, void* workspace_dev, size_t workspace_bytes, void* stream);
/* algorithmic FLOPs (2 x MAC) of one forward at (B, T): the roofline numerator used by bench.py */
/* Activation pre-scales of the split-precision path: one layer-by-layer forward on `mel` with an abs-max reduction in front of
 * every convolution; every layer's input scale becomes the power of two that puts that maximum in [2^9, 2^10) (six binades of
 * head-room below fp16's 65504, full 22-bit accuracy for values down to 2^-13 of the maximum).  The first ttsc_hifigan_forward
 * after the weights or the precision changed calibrates by itself — by default on a FIXED built-in probe mel (log-mel range
 * [-5, 1]), so the scales depend on the weights alone and a handle's output never depends on which utterance came first
 * (env TTSC_HIFIGAN_CALIBRATE=input: on that call's own input; =0: all scales stay 1).  Call this function to calibrate on other
 * data.  wav_dev / workspace as for the forward (wav_dev receives that forward's output).
 * Range guard: every split-precision forward checks (one stream synchronisation) whether conv_post emitted a non-finite sample —
 * an fp16 overflow in any layer reaches the waveform as NaN; it then re-calibrates on that input, reruns, and returns TTSC_ERANGE
 * only if the output is still non-finite.  ttsc_hifigan_recalibrations counts the reruns (env TTSC_HIFIGAN_RANGE_CHECK=0: off). */
int ttsc_hifigan_calibrate(ttsc_hifigan* g, const float* mel_dev, int32_t B, int64_t T, float* wav_dev, void* workspace_dev,
                           size_t workspace_bytes, void* stream);
int ttsc_hifigan_get_activation_scale(const ttsc_hifigan* g, const char* layer_name, float* scale_out);
/* Restores persisted scales (the file a checkpoint keeps beside it): ALL n layers of the generator by name, powers of two.  Pending
 * weights are uploaded first; the handle then counts as calibrated until its weights change. */
int ttsc_hifigan_set_activation_scales(ttsc_hifigan* g, const char* const* layer_names, const float* scales, int32_t n);
int32_t ttsc_hifigan_recalibrations(const ttsc_hifigan* g);
/* The two side streams of the branch schedule (the ResBlocks of a layer-by-layer stage beside each other for ragged / small batches, TTSC_HIFIGAN_BRANCH_STREAMS):
 * by default the handle creates its own on first use; a host that keeps a stream pool hands two of ITS streams in (borrowed: never destroyed here; the caller
 * keeps them alive while the handle can run).  Why it matters: the runtime multiplexes all streams of a process onto four hardware queues, and two more
 * streams shift the queue every stream created after them lands on — a training step that followed a small-batch forward in the same process ran 61.5 -> 69 ms
 * (profiles/r06_branch_stream_queues.log).  Outputs are identical bit for bit whichever streams are used.  (nullptr, nullptr): back to the handle's own. */
int ttsc_hifigan_set_branch_streams(ttsc_hifigan* g, void* stream_a, void* stream_b);
/* Range guard mode: 1 (default) = every forward synchronises its stream, re-calibrates + reruns when tripped; 2 = deferred: forwards
 * never wait, the guard word stays sticky on the device until ttsc_hifigan_range_status() (which synchronises `stream`) reads and
 * clears it: 1 = some forward since the last call emitted a non-finite sample — its output must be discarded / recomputed —, 0 = all
 * clean; 0 = guard off. */
int ttsc_hifigan_set_range_check(ttsc_hifigan* g, int32_t mode);
int32_t ttsc_hifigan_range_status(ttsc_hifigan* g, void* stream);
int ttsc_hifigan_algorithmic_flops(const ttsc_hifigan* g, int32_t B, int64_t T, double* flops_out);
void ttsc_hifigan_destroy(ttsc_hifigan* g);

/* ------------------------------------------------------------------------------------------------
 * WaveRNN autoregressive vocoder network.  Replaces `cube.networks.modules.WaveRNN._inference`
 * (cube/networks/modules.py:453-503; constructed vocoder.py:45-57) — the per-sample python loop of
 * nn.GRU(seq=1) -> tanh(Linear) -> Linear -> output_functions.sample — with one persistent kernel.  Every output distribution
 * of cube/networks/loss.py is sampled in the kernel: MULAWOutput / RAWOutput (Categorical.sample as Gumbel-max, loss.py:218-307),
 * MOLOutput (the reference default; Gumbel-max over 10 mixtures + logistic inverse CDF, loss.py:163-201), GaussianOutput
 * (loss.py:50-52) and BetaOutput (two Marsaglia-Tsang gamma variates, loss.py:83-92).
 * Weight names are the reference state_dict keys: "_rnns.<l>.weight_ih_l0", "_rnns.<l>.weight_hh_l0",
 * "_rnns.<l>.bias_ih_l0", "_rnns.<l>.bias_hh_l0", "_lowres_conv.<i>.conv.weight|bias",
 * "_preoutput.linear_layer.weight|bias", "_output.linear_layer.weight|bias"; "_skip.*" is accepted and ignored
 * (dead layer, modules.py:424).  Arithmetic contract: include/ttscube_math.h + oracle/wavernn_ref.c.
 * ------------------------------------------------------------------------------------------------ */
typedef struct ttsc_wavernn ttsc_wavernn;

enum { TTSC_WR_OUT_MULAW = 0, TTSC_WR_OUT_RAW = 1, TTSC_WR_OUT_MOL = 2, TTSC_WR_OUT_GM = 3, TTSC_WR_OUT_BETA = 4 };
/* sampler noise: none (arg-max / distribution mode) | injected [B,L,W] | in-kernel Philox-4x32-10 counters.
 * W (floats per step) = 256 for mulaw/raw (Gumbel terms), 11 for MOL (10 Gumbel terms -log(-log u) + 1 logistic term
 * log u - log(1-u)), 1 for Gaussian (0.8 * N(0,1)), 18 for Beta (per gamma variate: boost uniform + 4 (normal, uniform) rounds);
 * defined next to the samplers in include/ttscube_math.h. */
enum { TTSC_WR_MODE_ARGMAX = 0, TTSC_WR_MODE_NOISE = 1, TTSC_WR_MODE_PHILOX = 2 };

typedef struct {
    int32_t H;            /* layer_size (GRU hidden), multiple of 8, <= 512 */
    int32_t num_layers;   /* stacked GRUs (1..4) */
    int32_t use_lowres;   /* 1: high-res net conditioned on x_low (in_dim 102); 0: low-res net (in_dim 81) */
    int32_t upsample;     /* mel repeat factor: 240 (hr) / 24 (lr) */
    int32_t upsample_low; /* 10 */
    int32_t S;            /* output_functions.sample_size: 256 mulaw/raw, 30 mol, 2 gm/beta */
    int32_t n_mel;        /* 80 */
    int32_t out_kind;     /* TTSC_WR_OUT_* */
} ttsc_wavernn_cfg;

int ttsc_wavernn_create(const ttsc_wavernn_cfg* cfg, ttsc_wavernn** out);
int ttsc_wavernn_set_weight(ttsc_wavernn* w, const char* name, const float* host, const int64_t* shape, int32_t nd);
/* samples emitted for T mel frames / Tl low-res samples: min(T*upsample, Tl*upsample_low) (modules.py:467) */
int64_t ttsc_wavernn_out_len(const ttsc_wavernn* w, int64_t T, int64_t Tl);
size_t ttsc_wavernn_workspace_bytes(const ttsc_wavernn* w, int32_t B, int64_t T, int64_t Tl);
/* mel_dev [B,T,n_mel]; xlow_dev [B,Tl] (hr net) or NULL; noise_dev [B,L,W] (mode NOISE) or NULL;
 * forced_x_dev [B,L] or NULL (teacher forcing: feedback = forced_x[t], logits then equal _train_forward);
 * outputs idx_dev uint8 [B,L] (class index; mixture index for MOL; 0 for gm/beta), wav_dev fp32 [B,L] (what _inference returns), logits_dev
 * fp32 [B,L,S] or NULL. */
int ttsc_wavernn_decode(ttsc_wavernn* w, const float* mel_dev, const float* xlow_dev, int32_t B, int64_t T, int64_t Tl,
                        int32_t mode, const float* noise_dev, uint64_t seed, const float* forced_x_dev,
                        uint8_t* idx_dev, float* wav_dev, float* logits_dev, void* workspace_dev, size_t workspace_bytes,
                        void* stream);
/* Which kernel ran the last decode and whether its hand-offs completed.  Synchronises `stream`, then: -1 = single-workgroup
 * streaming kernel; 2 = tile kernel (8 workgroups step 8 utterances, each owning an eighth of the weight rows; chosen for
 * one-layer networks (any output head) when ceil(B/8)*8 <= number of CUs, H <= 512; env TTSC_WR_TILE=0 turns it off);
 * 1 = the tile kernel gave up on an inter-workgroup hand-off (bounded spin timed out; outputs invalid). */
int ttsc_wavernn_last_status(ttsc_wavernn* w, void* stream);
void ttsc_wavernn_destroy(ttsc_wavernn* w);

/* ------------------------------------------------------------------------------------------------
 * Linear (fp32 MFMA NT GEMM): y[M, :N] = act(x[M, :K] . w[N,K]^T + bias) [+ y if accumulate].
 * Replaces torch.nn.Linear inside LinearNorm (cube/networks/modules.py:24-34) for _dur_output, _pitch_output,
 * _cond_output, _mel_output, PreNet, and carries the hoisted LSTM input projections.  All pointers are device
 * pointers (the weight is whatever torch holds: [out, in] row-major).  ldx is a row stride only: 0 < ldx < K reads
 * overlapping rows (STFT framing, io_utils/melspec.py); ldy >= N.
 * ------------------------------------------------------------------------------------------------ */
int ttsc_linear_forward(const float* x_dev, const float* w_dev, const float* bias_dev, float* y_dev, int64_t M, int32_t N,
                        int32_t K, int64_t ldx, int64_t ldy, int32_t act, int32_t accumulate, void* stream);
/* The same contract on the f16 matrix pipe with fp32-class accuracy (gemm.hip: gemm_nt_f16x3_kernel): operands split into fp16 hi / lo halves on
 * their way into LDS, three MFMA products per tile (lo.hi + hi.lo + hi.hi, fp32 accumulation) — ~2^-22 relative per product.  Operands must lie
 * inside the fp16 range (|v| <= 65504): anything beyond sets a sticky per-device status word, read (and cleared; synchronises) by
 * ttsc_gemm_split_status — 1 = a result since the last call is invalid.  lengths_dev [M / period] or NULL: rows are [utterance][period]; 128-row
 * tiles that lie wholly at or beyond their utterances' lengths are skipped and stay unwritten (the recurrences never read them).
 * ttsc_linear_split_supported: K >= 32, K % 4 == 0, ldx % 4 == 0, ldx >= K (and 16-byte aligned x / w); otherwise TTSC_EINVAL — use
 * ttsc_linear_forward.  Callers: the hoisted input projections of the text-side BiLSTMs at inference (cube/networks/modules.py:873-905). */
int32_t ttsc_linear_split_supported(int64_t M, int32_t N, int32_t K, int64_t ldx);
int ttsc_linear_forward_split(const float* x_dev, const float* w_dev, const float* bias_dev, float* y_dev, int64_t M, int32_t N, int32_t K,
                              int64_t ldx, int64_t ldy, int32_t act, int32_t accumulate, const int32_t* lengths_dev, int32_t period, void* stream);
int32_t ttsc_gemm_split_status(void);

/* General fp32-MFMA GEMM for the backward passes of the Linears and of the hoisted recurrent projections (training, row a9; the
 * reference leaves these to autograd over torch.nn.Linear / nn.GRU / nn.LSTM: cube/networks/modules.py:505-563, cubegan.py:85-189):
 *     C[M, :N] (+)= opA(A) . opB(B)     transA = 0: A is [M,K] (lda), 1: A is [K,M];   transB = 0: B is [K,N] (ldb), 1: B is [N,K]
 *   dx = dG . W (NN), dW = dG^T . x (TN), dW_hh = dG^T . h_prev (TN with b_row_shift = -1 / +1 and b_period = T: row r of the
 *   contraction reads B row r + shift when 0 <= r % period + shift < period and zero otherwise, so h_prev is never materialised).
 * When M x N tiles alone cannot fill the device the contraction is split over K; partial tiles go through ws_dev
 * (>= ttsc_gemm_workspace_bytes(M, N, K); may be NULL when that is 0) and are added in a fixed order (deterministic).
 * ttsc_colsum: out[c] (+)= sum_r x[r, c] (bias gradients), two fixed-order stages through ws_dev (>= ttsc_colsum_workspace_bytes). */
size_t ttsc_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K);
int ttsc_gemm(int32_t transA, int32_t transB, int64_t M, int64_t N, int64_t K, const float* a_dev, int64_t lda, const float* b_dev, int64_t ldb,
              float* c_dev, int64_t ldc, int32_t accumulate, int64_t b_row_shift, int64_t b_period, void* ws_dev, size_t ws_bytes, void* stream);
size_t ttsc_colsum_workspace_bytes(int64_t R, int64_t C);
int ttsc_colsum(const float* x_dev, int64_t R, int64_t C, int64_t ld, float* out_dev, int32_t accumulate, void* ws_dev, size_t ws_bytes,
                void* stream);

/* Measurement helper (bench.py `roofline`): executed dense f16 MFMA TFLOP/s that a register-only v_mfma_f32_32x32x16_f16 loop sustains on every
 * CU of the current device for ~ms_target milliseconds, by operand data — mode 0: all-zero operands, 1: random fp16 operands, 2: the
 * split-precision mix (hi x hi, hi x lo, lo x hi with lo at 2^-11 scale).  The chip clocks to its power budget: real data sustains ~2/3 of the
 * nominal peak (csrc/probe.hip).  Synchronises `stream`.  No counterpart in the reference (it has no kernels). */
int ttsc_probe_mfma_tflops(int32_t mode, double ms_target, double* tflops_out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * LSTM / BiLSTM recurrence.  Replaces the sequential part of torch.nn.LSTM (gate order i,f,g,o, `_reverse`
 * direction) for Languasito2 (modules.py:873-905) and CubenetTextcoder (textcoder.py:55-92).
 *   xg_dev        [B, T, ndir*4H]  pre-activations  x.W_ih^T + b_ih + b_hh  (from ttsc_linear_forward)
 *   whh_packed    from ttsc_lstm_pack_whh(weight_hh_l{k}[, weight_hh_l{k}_reverse] stacked [ndir,4H,H] on the host)
 *   y_dev         [B, T, ldy]; direction d writes columns [yoff + d*H, yoff + (d+1)*H)
 *   lengths_dev   int32 [B] or NULL: pack_padded_sequence semantics (outputs beyond the length are zero)
 *   h0/c0/hn/cn   optional [ndir, B, H] initial / final states (NULL = zeros / not returned)
 * ------------------------------------------------------------------------------------------------ */
int ttsc_lstm_pack_whh(const float* whh_host, int32_t ndir, int32_t H, float** whh_dev_out);
int ttsc_lstm_seq_forward(const float* xg_dev, const float* whh_packed_dev, float* y_dev, const int32_t* lengths_dev,
                          int32_t B, int32_t T, int32_t H, int32_t ndir, int64_t ldy, int32_t yoff, const float* h0_dev,
                          const float* c0_dev, float* hn_dev, float* cn_dev, void* stream);
/* Training of the mel-decoder stacks (Languasito2 / CubenetTextcoder LSTMs inside `Cubegan.training_step`, cubegan.py:85-189;
 * the reference differentiates torch.nn.LSTM):
 *   ttsc_lstm_pack_whh_device   weight_hh [ndir,4H,H] on the DEVICE -> out_dev [ndir*4H*H]: the forward packing (transpose = 0) or
 *                               the W_hh^T packing the backward kernel streams (transpose = 1); no host round trip
 *   ttsc_lstm_seq_forward_train same recurrence, additionally saves the post-activation gates [B,T,ndir*4H] (i,f,g,o) and the
 *                               cell states [B,T,ndir*H]
 *   ttsc_lstm_seq_backward      backward through time: dy [B,T,ldy] (columns yoff + d*H ...) -> gradient wrt the pre-activation
 *                               gates dgates [B,T,ndir*4H] (zero beyond lengths).  The caller finishes with plain GEMMs:
 *                               dx = dgates W_ih, dW_ih = dgates^T x, dW_hh = dgates^T h_prev, db = sum dgates. */
/* With few sequences (G * B * ndir <= number of CUs; G <= 4, env TTSC_LSTM_SPLIT caps it, 1 = off) the forward, train-forward and
 * backward recurrences split every (utterance, direction) over G workgroups (TTSC_LSTM_SPLIT_INFER=0 exempts inference) that exchange the state once per step (see
 * ttsc_gru_split_status); results then differ from the single-workgroup kernels in summation order only (<= 1e-6 relative).
 * ttsc_lstm_split_status: 0 = every hand-off since the last call completed, 1 = a bounded spin timed out (sticky until read:
 * later split launches give up at once, so callers that care check it once per step / synthesis).  Synchronises the device. */
int32_t ttsc_lstm_split_status(void);
/* The same verdict for the split recurrences (LSTM, GRU, mel-AR) launched on ONE stream, waiting for that stream only: bit 0 LSTM, bit 1 GRU,
 * bit 2 mel-AR; reported once, then re-armed; < 0 on a HIP error.  For a training step that runs its text side on a stream of its own and must know
 * that this stream's backward pass is sound BEFORE its gradients are exchanged and applied (cube/networks/cubegan.py:172-180 is the update it guards),
 * without draining the other streams. */
int32_t ttsc_split_status_stream(void* stream);
/* Device-side form of the same question: ONE launch on `stream` ORs the verdict bits of every split recurrence launched on that stream so far (bit 0 LSTM,
 * bit 1 GRU, bit 2 mel-AR, bit 3 other; bit 4 = the split-precision GEMM's range word) into *dst_dev and re-arms the sticky words; the host waits for
 * nothing.  flags: bit 0 = include the GEMM's range word, bit 1 = the recurrences of EVERY stream of the device (the caller has made `stream` wait for
 * them).  Returns the number of status words looked at (0: nothing to ask, no launch), < 0 on a HIP error. */
int32_t ttsc_split_status_collect(void* stream, uint32_t* dst_dev, int32_t flags);
/* Utterances per member group of the register-resident split recurrence (H = 256 / 512: 4 / 16 workgroups per group hold W_hh in registers).
 * 0 (default) = automatic: the smallest of 1 / 2 / 4 that takes the padded batch in one launch — the shortest step.  n = 1 / 2 / 4 / 8: n per
 * group whenever the batch has that many — n times fewer CUs held for a somewhat longer step, for callers that run the recurrence beside a
 * kernel that fills the chip (Cubegan.inference_pipelined).  Results do not depend on it (per utterance the arithmetic is the same).
 * Process-wide; returns the previous value, -1 for a bad n.  Not in the reference (B = 1 there, cube/networks/modules.py:946-953). */
int32_t ttsc_lstm_set_group_size(int32_t n);
int ttsc_lstm_pack_whh_device(const float* whh_dev, int32_t ndir, int32_t H, int32_t transpose, float* out_dev, void* stream);
int ttsc_lstm_seq_forward_train(const float* xg_dev, const float* whh_packed_dev, float* y_dev, const int32_t* lengths_dev,
                                int32_t B, int32_t T, int32_t H, int32_t ndir, int64_t ldy, int32_t yoff, float* gates_dev,
                                float* c_dev, void* stream);
int ttsc_lstm_seq_backward(const float* dy_dev, const float* gates_dev, const float* c_dev, const float* whhT_packed_dev,
                           float* dgates_dev, const int32_t* lengths_dev, int32_t B, int32_t T, int32_t H, int32_t ndir,
                           int64_t ldy, int32_t yoff, void* stream);
/* Training of the WaveRNN vocoder (`WaveRNN._train_forward` cube/networks/modules.py:505-539 + `training_step` 553-563: torch.nn.GRU
 * over the teacher-forced sequence, differentiated by torch autograd).  Gate order r,z,n; one layer, unidirectional:
 *   ttsc_gru_pack_whh_device  weight_hh [3H,H] (device) -> out_dev [3H*H]: forward (transpose=0) / backward (transpose=1) packing
 *   ttsc_gru_seq_forward      xg = W_ih x + b_ih [B,T,3H] -> y [B,T,H]; saved_dev [B,T,4H] (r,z,n,W_hn h + b_hn) or NULL (inference)
 *   ttsc_gru_seq_backward     dy [B,T,H] -> dgi [B,T,3H] (grad wrt W_ih x + b_ih), dgh [B,T,3H] (grad wrt W_hh h + b_hh);
 *                             the caller finishes with GEMMs: dx = dgi W_ih, dW_ih = dgi^T x, dW_hh = dgh^T h_prev, biases = sums */
/* With few utterances (B * G <= number of CUs) both recurrences split every utterance over G workgroups (env TTSC_GRU_SPLIT caps
 * G, default 4, 1 = off) that exchange the state through y / dgh once per step; launches of one process must then be
 * stream-ordered (shared hand-off counters).  ttsc_gru_split_status: 0 = every hand-off since the last call completed, 1 = a spin timed out
 * (sticky until read).  Synchronises the device. */
int32_t ttsc_gru_split_status(void);
int ttsc_gru_pack_whh_device(const float* whh_dev, int32_t H, int32_t transpose, float* out_dev, void* stream);
int ttsc_gru_seq_forward(const float* xg_dev, const float* whh_packed_dev, const float* bhh_dev, float* y_dev, float* saved_dev,
                         const float* h0_dev, int32_t B, int32_t T, int32_t H, void* stream);
int ttsc_gru_seq_backward(const float* dy_dev, const float* saved_dev, const float* y_dev, const float* h0_dev,
                          const float* whhT_packed_dev, float* dgi_dev, float* dgh_dev, int32_t B, int32_t T, int32_t H, void* stream);
/* frees a device buffer returned by a ttsc_*_pack_* function */
void ttsc_device_free(void* p);

/* ------------------------------------------------------------------------------------------------
 * Autoregressive mel decoder loop of CubenetTextcoder as one persistent kernel.  Replaces the python loop at
 * cube/networks/textcoder.py:174-185: per step PreNet (modules.py:159-164, dropout p=0.5 always on) -> 2-layer LSTM step
 * -> Linear H->O (three 80-bin frames) -> feedback of the last frame.  The caller hoists the overlay part of the layer-1
 * input projection: xg1_dev [B,S,4H] = overlay . W_ih1[:, :overlay]^T + b_ih1 + b_hh1 (ttsc_linear_forward).
 * masks_dev: {0,1} floats [B,S,2,P] (injected dropout masks) or NULL (Bernoulli(0.5) from Philox(seed)); steps_dev int32
 * [B] or NULL (all S); y_dev [B,S,O] pre-postnet mel (zeros beyond an utterance's steps).
 * ------------------------------------------------------------------------------------------------ */
typedef struct ttsc_melar ttsc_melar;
int ttsc_melar_create(int32_t H, int32_t P, int32_t M, int32_t O, ttsc_melar** out);
int ttsc_melar_set_weights(ttsc_melar* m, const float* w_ih1, int64_t ld1, const float* w_hh1, const float* w_ih2,
                           const float* w_hh2, const float* b_ih2, const float* b_hh2, const float* w_out, const float* b_out,
                           const float* pn_w1, const float* pn_b1, const float* pn_w2, const float* pn_b2);
/* With few utterances (G * B <= number of CUs; G <= 8, env TTSC_MELAR_SPLIT caps it, 1 = off) the decode loop of one utterance is
 * split over G workgroups (each owns H/G units of both LSTM layers; h1 / h2 exchanged every step, protocol of
 * ttsc_lstm_split_status); results differ from the single-workgroup kernel in summation order only. */
int32_t ttsc_melar_split_status(void);
int ttsc_melar_decode(const ttsc_melar* m, const float* xg1_dev, int32_t B, int32_t S, const float* masks_dev, uint64_t seed,
                      const int32_t* steps_dev, float* y_dev, void* stream);
void ttsc_melar_destroy(ttsc_melar* m);

/* Device-side duration -> alignment (row f2 of SURVEY.md §8; replaces the host round trip of
 * cube/networks/modules.py:946-953,1043-1053 and cube/networks/textcoder.py:160-166,291-302).
 * ttsc_align_durations: logits [B,N,D] fp32 (duration head) -> durs [B,N] int32 = argmax over D (first maximum; 0 beyond
 *   len_dev[b], len_dev NULL = N), f2p [B,Fcap] int32 frame -> phone map (phone p repeated durs[b,p] times, in order),
 *   flen [B] int32 = number of frames (clamped to Fcap).
 * ttsc_expand_rows: out [B,F,C] = rows of x [B,N,C] gathered through every stride-th entry of f2p; frames beyond an
 *   utterance's own count repeat its last aligned row (stride 1) or row N-1 (stride > 1), as the reference's gathers pad. */
int ttsc_align_durations(const float* logits_dev, const int32_t* len_dev, int32_t B, int32_t N, int32_t D, int32_t* durs_dev,
                         int32_t* f2p_dev, int32_t* flen_dev, int32_t Fcap, void* stream);
int ttsc_expand_rows(const float* x_dev, const int32_t* f2p_dev, const int32_t* flen_dev, int32_t B, int32_t N, int32_t C, int32_t Fcap,
                     int32_t stride, int32_t F, float* out_dev, void* stream);
/* Input of Languasito2's conditioning recurrence in one launch (cube/networks/modules.py:962-994: vuv = round(p[..., 1]), pitch = p[..., 0] * max_pitch * vuv,
 * cat[expand(g), pitch / max_pitch]): out [B,F,Cp] = rows of g [B,N,C] gathered through f2p (ttsc_expand_rows' rule at stride 1), column C = pitch * (1 / max_pitch),
 * columns C + 1 .. Cp - 1 zero (Cp = C + 1 rounded up to the split GEMM's multiple of 4); pitch [B,F] is written too.  pitch_out_dev [B,F,2] = the pitch head's
 * sigmoid outputs. */
int ttsc_cond_input(const float* g_dev, const int32_t* f2p_dev, const int32_t* flen_dev, const float* pitch_out_dev, float max_pitch, int32_t B, int32_t N,
                    int32_t C, int32_t Cp, int32_t Fcap, int32_t F, float* pitch_dev, float* out_dev, void* stream);

/* STFT-magnitude / mel-spectrogram helpers around ttsc_linear_forward (the DFT and the mel projection are GEMMs):
 * hifigan.meldataset.mel_spectrogram [EXTERNAL; cube/networks/cubegan.py:137-138,247-248 — the 45 x mel-L1 loss of the GAN step,
 * forward and backward] and MelVocoder.melspectrogram (cube/io_utils/vocoder.py:54-98, feature extraction, row f4).
 *   ttsc_stft_mag            reim [M, 2*NB] (re | im) -> mag [M, ldm] = sqrt(re^2 + im^2 + eps)  (columns >= NB zeroed)
 *   ttsc_stft_mag_backward   d(re|im) = dmag * (re|im) / mag
 *   ttsc_log_clamp           y = scale * ln(max(x, minv))  (scale 1: natural log; 1/ln 10: log10);  _backward: dx = dy*scale/x above minv
 *   ttsc_overlap_add         backward of the framing: y [B, Lp] += frames [B, F, n_fft] at hop (fixed summation order) */
int ttsc_stft_mag(const float* reim_dev, int64_t M, int32_t NB, int32_t ldm, float eps, float* mag_dev, void* stream);
int ttsc_stft_mag_backward(const float* dmag_dev, const float* reim_dev, const float* mag_dev, int64_t M, int32_t NB, int32_t ldm,
                           float* dreim_dev, void* stream);
int ttsc_log_clamp(const float* x_dev, int64_t n, float minv, float scale, float* y_dev, void* stream);
int ttsc_log_clamp_backward(const float* dy_dev, const float* x_dev, int64_t n, float minv, float scale, float* dx_dev, void* stream);
int ttsc_overlap_add(const float* frames_dev, int32_t B, int32_t F, int32_t n_fft, int32_t hop, int64_t Lp, float* y_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TTSCUBE_HIP_H */
