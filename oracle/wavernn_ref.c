/*
 * ORACLE (test infrastructure — never linked or called by the product path).
 *
 * Plain-C restatement of TTS-Cube's WaveRNN sampling loop and of the CubenetVocoder chunk folding:
 *   WaveRNN._inference        cube/networks/modules.py:453-503   (conditioning build + per-sample loop)
 *   UpsampleNetI / R          cube/networks/modules.py:346-354, 378-389
 *   ConvNorm low-res convs    cube/networks/modules.py:416-420, 459-461
 *   nn.GRU cell (r,z,n)       torch semantics used at modules.py:427,485
 *   MULAWOutput.sample/decode cube/networks/loss.py:227-230,257-269;  RAWOutput loss.py:288-299
 *   MOLOutput.sample          cube/networks/loss.py:163-201  (the default of WaveRNN, modules.py:398)
 *   GaussianOutput.sample     cube/networks/loss.py:50-52;   BetaOutput.sample loss.py:83-92
 *
 * Arithmetic contract (shared with the HIP kernel so that µ-law indices are bit-exact): every dot product of the GRU
 * cells is ONE k-ordered fp32 fmaf chain seeded with the bias; the two output Linears (pre-output H -> 256, output
 * 256 -> S) are FOUR k-ordered chains over consecutive quarters of the inputs — quarter q covers the 4-input blocks
 * [q KB / 4, (q + 1) KB / 4) of the KB = K / 4 blocks, the first chain seeded with the bias, the others with 0 — added as
 * ((p0 + p1) + p2) + p3 (round 6: the kernels run the four quarters side by side; the reference's own order is whatever
 * MKL's sgemv does on the host, neither is "the" order — the reference-made goldens are the arbiter);
 * sigmoid/tanh/log come from include/ttscube_math.h;
 * the categorical sample is the Gumbel-max  idx = argmax_s(logits_s + g_s)  (first maximum wins), which is
 * the reference's Categorical(logits).sample() == argmax_s softmax_s / E_s with g = -log E.
 * Pinned against the reference itself (imported, real torch RNG stream replayed) by
 * tools/gen_golden_wavernn.py -> tests/golden/wavernn_*.npz -> tests/test_oracle_wavernn.py.
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC (see oracle/Makefile).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/ttscube_math.h"

#define MODE_ARGMAX 0
#define MODE_NOISE 1   /* injected Gumbel noise [B, L, S] */
#define MODE_PHILOX 2  /* counter-based noise: philox(counter = (s/4, step, b, stream), key = seed) */

#define OUT_MULAW 0
#define OUT_RAW 1
#define OUT_MOL 2    /* S = 30: 10 mixture logits | 10 means | 10 log-scales; noise [B, L, 11] = 10 Gumbel terms + 1 logistic term */
#define OUT_GM 3     /* S = 2: mean, log-std; noise [B, L, 1] = 0.8 * N(0,1) */
#define OUT_BETA 4   /* S = 2: log alpha, log beta; noise [B, L, 18] = per gamma variate: boost uniform + 4 (normal, uniform) rounds */

typedef struct {
    int32_t H;            /* GRU size */
    int32_t num_layers;   /* 1 or 2 ... */
    int32_t use_lowres;   /* 1: hr net (in_dim 102), 0: lr net (in_dim 81) */
    int32_t upsample;     /* mel repeat factor (240 hr / 24 lr) */
    int32_t upsample_low; /* 10 */
    int32_t S;            /* logits per sample (256) */
    int32_t n_mel;        /* 80 */
    int32_t out_kind;     /* OUT_* */
} wr_cfg;

typedef struct {
    const float* w_ih[4];   /* [3H, in_l]  (torch weight_ih_l0 of _rnns.l) */
    const float* w_hh[4];   /* [3H, H] */
    const float* b_ih[4];
    const float* b_hh[4];
    const float* w_pre;     /* [256, H] */
    const float* b_pre;
    const float* w_out;     /* [S, 256] */
    const float* b_out;
    const float* lc_w[3];   /* low-res convs: [20,1,7], [20,20,7], [20,20,7] */
    const float* lc_b[3];
    const float* lut;       /* [256] µ-law decode table (tests/golden/mulaw_lut.npy, from the reference) */
} wr_weights;

static float dot_chain(const float* w, const float* x, int n, float acc) {
    for (int k = 0; k < n; ++k) acc = fmaf(w[k], x[k], acc);
    return acc;
}

/* out[r] = dot_chain(W + r*ld, x, n, bias[r]) for r < rows.  Eight rows advance in lockstep so that the CPU overlaps eight
 * independent fmaf chains (a single chain is bound by the fma latency); every row is still ONE k-ordered chain seeded with
 * its bias, i.e. bit-identical to dot_chain. */
static void matvec_chain(const float* W, int ld, const float* x, int n, const float* bias, int rows, float* out) {
    int r = 0;
    for (; r + 8 <= rows; r += 8) {
        float a0 = bias[r], a1 = bias[r + 1], a2 = bias[r + 2], a3 = bias[r + 3];
        float a4 = bias[r + 4], a5 = bias[r + 5], a6 = bias[r + 6], a7 = bias[r + 7];
        const float* w = W + (size_t)r * ld;
        for (int k = 0; k < n; ++k) {
            const float xk = x[k];
            a0 = fmaf(w[k], xk, a0);
            a1 = fmaf(w[(size_t)ld + k], xk, a1);
            a2 = fmaf(w[(size_t)2 * ld + k], xk, a2);
            a3 = fmaf(w[(size_t)3 * ld + k], xk, a3);
            a4 = fmaf(w[(size_t)4 * ld + k], xk, a4);
            a5 = fmaf(w[(size_t)5 * ld + k], xk, a5);
            a6 = fmaf(w[(size_t)6 * ld + k], xk, a6);
            a7 = fmaf(w[(size_t)7 * ld + k], xk, a7);
        }
        out[r] = a0, out[r + 1] = a1, out[r + 2] = a2, out[r + 3] = a3;
        out[r + 4] = a4, out[r + 5] = a5, out[r + 6] = a6, out[r + 7] = a7;
    }
    for (; r < rows; ++r) out[r] = dot_chain(W + (size_t)r * ld, x, n, bias[r]);
}

/* out[r] = ((p0 + p1) + p2) + p3, p_q = dot_chain over the q-th quarter of the inputs (blocks of 4), p0 seeded with bias[r]:
 * the two output Linears (see the contract above).  n must be a multiple of 4. */
static void matvec_chain4(const float* W, int ld, const float* x, int n, const float* bias, int rows, float* out) {
    const int KB = n / 4;
    int lo[5];
    for (int q = 0; q <= 4; ++q) lo[q] = 4 * ((q * KB) / 4);
    for (int r = 0; r < rows; ++r) {
        const float* w = W + (size_t)r * ld;
        float acc = dot_chain(w + lo[0], x + lo[0], lo[1] - lo[0], bias[r]);
        for (int q = 1; q < 4; ++q) acc += dot_chain(w + lo[q], x + lo[q], lo[q + 1] - lo[q], 0.f);
        out[r] = acc;
    }
}

/* F.interpolate(x[B,1,Tl], 10*Tl, mode='linear') (align_corners=False), modules.py:353 */
void wr_interp_linear(const float* x, int Tl, int up, float* out) {
    const int n = Tl * up;
    const float scale = (float)Tl / (float)n;
    for (int t = 0; t < n; ++t) {
        float src = scale * ((float)t + 0.5f) - 0.5f;
        if (src < 0.f) src = 0.f;
        int i0 = (int)src;
        if (i0 > Tl - 1) i0 = Tl - 1;
        const int i1 = i0 + (i0 < Tl - 1 ? 1 : 0);
        const float l1 = src - (float)i0;
        const float l0 = 1.0f - l1;
        out[t] = l0 * x[i0] + l1 * x[i1];
    }
}

/* tanh(Conv1d(k=7, pad=3)) with a (ci outer, k inner) fmaf chain seeded by the bias; out-of-range taps
 * multiply a 0 input (kept in the chain so host and device execute the same operations). */
void wr_lowres_conv(const float* x, int cin, int cout, int T, const float* w, const float* b, float* y) {
    for (int co = 0; co < cout; ++co)
        for (int t = 0; t < T; ++t) {
            float acc = b[co];
            for (int ci = 0; ci < cin; ++ci)
                for (int k = 0; k < 7; ++k) {
                    const int p = t + k - 3;
                    const float v = (p >= 0 && p < T) ? x[ci * T + p] : 0.f;
                    acc = fmaf(w[(co * cin + ci) * 7 + k], v, acc);
                }
            y[co * T + t] = ttsc_tanhf(acc);
        }
}

/* number of samples WaveRNN._inference emits (modules.py:464-471) */
int64_t wr_out_len(const wr_cfg* c, int T, int Tl) {
    int64_t L = (int64_t)T * c->upsample;
    if (c->use_lowres) {
        const int64_t l2 = (int64_t)Tl * c->upsample_low;
        if (l2 < L) L = l2;
    }
    return L;
}

int wr_noise_width(const wr_cfg* c) {
    switch (c->out_kind) {
        case OUT_MOL: return TTSC_MOL_NOISE;
        case OUT_GM: return TTSC_GM_NOISE;
        case OUT_BETA: return TTSC_BETA_NOISE;
        default: return c->S;
    }
}

/*
 * mel [B,T,n_mel]; x_low [B,Tl] (hr only); noise [B,L,wr_noise_width] (MODE_NOISE); forced_x [B,L] or NULL: when given,
 * the fed-back sample at step t is forced_x[b][t] (teacher forcing: logits then equal WaveRNN._train_forward).
 * Outputs: out_idx [B,L] uint8, out_wav [B,L] float, out_logits [B,L,S] or NULL.
 */
/* wr_decode_at: the arrays hold utterances b_offset .. b_offset+B-1 of a larger batch — only the counter-based noise depends
 * on the absolute utterance index (counter word 2), so a few utterances of a big batch can be checked on their own. */
int wr_decode_at(const wr_cfg* c, const wr_weights* w, const float* mel, const float* x_low, int B, int T, int Tl,
                 int mode, const float* noise, uint64_t seed, const float* forced_x, int b_offset, uint8_t* out_idx, float* out_wav,
                 float* out_logits) {
    const int H = c->H, S = c->S, NM = c->n_mel;
    const int I0 = NM + (c->use_lowres ? 21 : 0) + 1;
    const int64_t L = wr_out_len(c, T, Tl);
    float* interp = NULL;
    float* f1 = NULL;
    float* f2 = NULL;
    float* x = (float*)malloc(sizeof(float) * (size_t)(I0 > H ? I0 : H));
    float* h = (float*)calloc((size_t)c->num_layers * H, sizeof(float));
    float* hn = (float*)malloc(sizeof(float) * H);
    float* gi = (float*)malloc(sizeof(float) * 3 * H);
    float* gh = (float*)malloc(sizeof(float) * 3 * H);
    float* pre = (float*)malloc(sizeof(float) * 256);
    float* logits = (float*)malloc(sizeof(float) * S);
    if (c->use_lowres) {
        interp = (float*)malloc(sizeof(float) * (size_t)Tl * c->upsample_low);
        f1 = (float*)malloc(sizeof(float) * 20 * (size_t)Tl);
        f2 = (float*)malloc(sizeof(float) * 20 * (size_t)Tl);
    }
    for (int b = 0; b < B; ++b) {
        const float* melb = mel + (size_t)b * T * NM;
        if (c->use_lowres) {
            const float* xl = x_low + (size_t)b * Tl;
            wr_interp_linear(xl, Tl, c->upsample_low, interp);
            wr_lowres_conv(xl, 1, 20, Tl, w->lc_w[0], w->lc_b[0], f1);
            wr_lowres_conv(f1, 20, 20, Tl, w->lc_w[1], w->lc_b[1], f2);
            wr_lowres_conv(f2, 20, 20, Tl, w->lc_w[2], w->lc_b[2], f1);
        }
        memset(h, 0, sizeof(float) * (size_t)c->num_layers * H);
        float last_x = 0.f; /* modules.py:473 */
        for (int64_t t = 0; t < L; ++t) {
            /* cond = [mel(t//up) | lowres feats(t//10) | interp(t)] ++ last_x   (modules.py:468-481) */
            memcpy(x, melb + (size_t)(t / c->upsample) * NM, sizeof(float) * NM);
            int n = NM;
            if (c->use_lowres) {
                const int tl = (int)(t / c->upsample_low);
                for (int q = 0; q < 20; ++q) x[n++] = f1[q * Tl + tl];
                x[n++] = interp[t];
            }
            x[n++] = last_x;
            int in_l = I0;
            for (int l = 0; l < c->num_layers; ++l) {
                float* hl = h + (size_t)l * H;
                matvec_chain(w->w_ih[l], in_l, x, in_l, w->b_ih[l], 3 * H, gi);   /* rows: r | z | n (torch GRU layout) */
                matvec_chain(w->w_hh[l], H, hl, H, w->b_hh[l], 3 * H, gh);
                for (int j = 0; j < H; ++j) {
                    const float gi_r = gi[j], gi_z = gi[H + j], gi_n = gi[2 * H + j];
                    const float gh_r = gh[j], gh_z = gh[H + j], gh_n = gh[2 * H + j];
                    const float r = ttsc_sigmoidf(gi_r + gh_r);
                    const float z = ttsc_sigmoidf(gi_z + gh_z);
                    const float rg = r * gh_n;
                    const float nn = ttsc_tanhf(gi_n + rg);
                    const float d = hl[j] - nn;
                    hn[j] = fmaf(z, d, nn); /* (1-z)*n + z*h */
                }
                memcpy(hl, hn, sizeof(float) * H);
                memcpy(x, hn, sizeof(float) * H);
                in_l = H;
            }
            matvec_chain4(w->w_pre, H, x, H, w->b_pre, 256, pre);
            for (int j = 0; j < 256; ++j) pre[j] = ttsc_tanhf(pre[j]);
            matvec_chain4(w->w_out, 256, pre, 256, w->b_out, S, logits);
            if (out_logits) memcpy(out_logits + ((size_t)b * L + t) * S, logits, sizeof(float) * S);
            int best = 0;
            float wav;
            if (c->out_kind == OUT_MULAW || c->out_kind == OUT_RAW) {
                float bs = 0.f;
                for (int s = 0; s < S; ++s) {
                    float g = 0.f;
                    if (mode == MODE_NOISE) {
                        g = noise[((size_t)b * L + t) * S + s];
                    } else if (mode == MODE_PHILOX) {
                        uint32_t r4[4];
                        ttsc_philox4x32((uint32_t)(s >> 2), (uint32_t)t, (uint32_t)(b + b_offset), 0u,
                                        (uint32_t)seed, (uint32_t)(seed >> 32), r4);
                        g = ttsc_gumbel(r4[s & 3]);
                    }
                    const float sc = logits[s] + g;
                    if (s == 0 || sc > bs) {
                        bs = sc;
                        best = s;
                    }
                }
                if (c->out_kind == OUT_MULAW)
                    wav = w->lut[best];
                else
                    wav = (((float)best / 255.0f) - 0.5f) * 2.0f; /* loss.py:297-299 */
            } else {
                /* continuous outputs: MODE_ARGMAX = zero noise (the mode of the selected component / the mean) */
                float nz[TTSC_BETA_NOISE];
                const int nw = wr_noise_width(c);
                for (int i = 0; i < nw; ++i) nz[i] = 0.f;
                if (mode == MODE_NOISE) {
                    memcpy(nz, noise + ((size_t)b * L + t) * nw, sizeof(float) * nw);
                } else if (mode == MODE_PHILOX) {
                    if (c->out_kind == OUT_MOL) ttsc_noise_mol((uint32_t)t, (uint32_t)(b + b_offset), seed, nz);
                    else if (c->out_kind == OUT_GM) ttsc_noise_gm((uint32_t)t, (uint32_t)(b + b_offset), seed, nz);
                    else ttsc_noise_beta((uint32_t)t, (uint32_t)(b + b_offset), seed, nz);
                } else if (c->out_kind == OUT_BETA) {
                    for (int v = 0; v < 2; ++v) nz[v * (1 + 2 * TTSC_BETA_TRIES)] = 0.5f, nz[v * (1 + 2 * TTSC_BETA_TRIES) + 2] = 0.5f;
                }
                if (c->out_kind == OUT_MOL)
                    wav = ttsc_sample_mol(logits, nz, nz[TTSC_MOL_NMIX], &best);
                else if (c->out_kind == OUT_GM)
                    wav = ttsc_sample_gm(logits, nz[0]);
                else
                    wav = ttsc_sample_beta(logits, nz);
            }
            out_idx[(size_t)b * L + t] = (uint8_t)best;
            out_wav[(size_t)b * L + t] = wav;
            last_x = forced_x ? forced_x[(size_t)b * L + t] : wav;
        }
    }
    free(interp);
    free(f1);
    free(f2);
    free(x);
    free(h);
    free(hn);
    free(gi);
    free(gh);
    free(pre);
    free(logits);
    return 0;
}

int wr_decode(const wr_cfg* c, const wr_weights* w, const float* mel, const float* x_low, int B, int T, int Tl,
              int mode, const float* noise, uint64_t seed, const float* forced_x, uint8_t* out_idx, float* out_wav,
              float* out_logits) {
    return wr_decode_at(c, w, mel, x_low, B, T, Tl, mode, noise, seed, forced_x, 0, out_idx, out_wav, out_logits);
}

/* exported for the math pin test */
float wr_expf(float x) { return ttsc_expf(x); }
float wr_logf(float x) { return ttsc_logf(x); }
float wr_tanhf(float x) { return ttsc_tanhf(x); }
float wr_sigmoidf(float x) { return ttsc_sigmoidf(x); }
float wr_gumbel(uint32_t r) { return ttsc_gumbel(r); }
float wr_normal_icdf(float p) { return ttsc_normal_icdf(p); }
float wr_sample_beta(const float* y, const float* nz) { return ttsc_sample_beta(y, nz); }
float wr_sample_mol(const float* y, const float* nz, int* k) { return ttsc_sample_mol(y, nz, nz[TTSC_MOL_NMIX], k); }
void wr_noise_beta(uint32_t t, uint32_t b, uint64_t seed, float* nz) { ttsc_noise_beta(t, b, seed, nz); }
void wr_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* out) {
    ttsc_philox4x32(c0, c1, c2, c3, k0, k1, out);
}
