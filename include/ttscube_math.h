/*
 * ttscube_math.h — bit-exact fp32 transcendental definitions shared by the HIP kernels and the C oracle.
 *
 * WaveRNN decoding is autoregressive: one flipped argmax changes every later sample, so "bit-exact µ-law
 * indices" (BASELINE.json north_star) is only definable if sigmoid/tanh/exp/log have ONE definition on
 * device and on the host.  libm (glibc) and the ROCm device library round differently, so these are
 * written with nothing but IEEE-754 correctly-rounded operations (fmaf, +, *, /, rintf) and integer bit
 * moves, which gcc (-ffp-contract=off) and hipcc (gfx950, correctly-rounded division is hipcc's default)
 * evaluate identically.  Accuracy: expf <= 1 ulp-ish (rel 1.2e-7), logf rel 2e-7; they are pinned against
 * torch's fp32 results in tests/test_oracle_wavernn.py within 1e-6.
 *
 * Every function is `static inline`; in HIP translation units they are __host__ __device__.
 */
#ifndef TTSCUBE_MATH_H
#define TTSCUBE_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define TTSC_HD __host__ __device__ static inline
#define TTSC_UNROLL _Pragma("unroll")
#else
#define TTSC_HD static inline
#define TTSC_UNROLL
#endif

TTSC_HD float ttsc_bits2f(uint32_t u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}
TTSC_HD uint32_t ttsc_f2bits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}

/* exp(x) for fp32: Cody-Waite reduction x = n*ln2 + r, |r| <= ln2/2, degree-7 Horner polynomial in fmaf,
 * result scaled by 2^n through the exponent field.  Inputs are clamped to [-87, 88]. */
TTSC_HD float ttsc_expf(float x) {
    x = x < -87.0f ? -87.0f : x;
    x = x > 88.0f ? 88.0f : x;
    const float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693145751953125f, x);          /* ln2 high part (exact product for |n| < 2^11) */
    r = fmaf(n, -1.42860682030941723212e-6f, r);        /* ln2 low part */
    float p = 1.0f / 5040.0f;
    p = fmaf(p, r, 1.0f / 720.0f);
    p = fmaf(p, r, 1.0f / 120.0f);
    p = fmaf(p, r, 1.0f / 24.0f);
    p = fmaf(p, r, 1.0f / 6.0f);
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    const int32_t ni = (int32_t)n;                       /* in [-126, 127] after the clamp */
    /* split the scale so that 2^ni never leaves the normal range */
    const int32_t n1 = ni / 2, n2 = ni - n1;
    const float s1 = ttsc_bits2f((uint32_t)(n1 + 127) << 23);
    const float s2 = ttsc_bits2f((uint32_t)(n2 + 127) << 23);
    return (p * s1) * s2;
}

/* natural log for x > 0 (normal floats): x = m * 2^e with m in [sqrt(1/2), sqrt(2)), log(m) through the
 * atanh series in s = (m-1)/(m+1).  */
TTSC_HD float ttsc_logf(float x) {
    uint32_t u = ttsc_f2bits(x);
    int32_t e = (int32_t)(u >> 23) - 127;
    u = (u & 0x007fffffu) | 0x3f800000u;
    float m = ttsc_bits2f(u);
    if (m > 1.41421356237f) {
        m = m * 0.5f;
        e += 1;
    }
    const float s = (m - 1.0f) / (m + 1.0f);
    const float z = s * s;
    float p = 1.0f / 11.0f;
    p = fmaf(p, z, 1.0f / 9.0f);
    p = fmaf(p, z, 1.0f / 7.0f);
    p = fmaf(p, z, 1.0f / 5.0f);
    p = fmaf(p, z, 1.0f / 3.0f);
    p = fmaf(p, z, 1.0f);
    const float lm = 2.0f * (s * p);
    const float fe = (float)e;
    return fmaf(fe, 0.693145751953125f, fmaf(fe, 1.42860682030941723212e-6f, lm));
}

TTSC_HD float ttsc_sigmoidf(float x) { return 1.0f / (1.0f + ttsc_expf(-x)); }

/* tanh(x) = sign(x) * (1 - 2/(exp(2|x|)+1)); |x| < 2^-4 uses the odd Taylor polynomial to avoid cancellation */
TTSC_HD float ttsc_tanhf(float x) {
    const float ax = fabsf(x);
    float t;
    if (ax < 0.0625f) {
        const float z = ax * ax;
        float p = 17.0f / 315.0f;
        p = fmaf(p, z, -2.0f / 15.0f);
        p = fmaf(p, z, 1.0f / 3.0f);
        p = -p;
        t = fmaf(p * z, ax, ax);
    } else {
        const float e = ttsc_expf(2.0f * ax);
        t = 1.0f - 2.0f / (e + 1.0f);
    }
    return x < 0.0f ? -t : t;
}

/* ---- counter-based RNG (Philox-4x32-10) for the in-kernel sampler; integer-only, so trivially bit-exact ---- */
TTSC_HD void ttsc_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                             uint32_t out[4]) {
    for (int i = 0; i < 10; ++i) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* uniform in (0,1): 23 random bits + one half, exactly representable in fp32, never 0 or 1 */
TTSC_HD float ttsc_u01(uint32_t r) { return ((float)(r >> 9) + 0.5f) * (1.0f / 8388608.0f); }

/* Gumbel(0,1) noise from one uniform: g = -log(-log(u)) */
TTSC_HD float ttsc_gumbel(uint32_t r) { return -ttsc_logf(-ttsc_logf(ttsc_u01(r))); }


/* ---- continuous output distributions of the vocoder (cube/networks/loss.py:35-215) -------------------------------------
 * ONE definition for the HIP kernel and the C oracle, like the transcendentals above.  Every sampler is a pure function of
 * (the output layer's row, a few noise scalars): in MODE_NOISE the scalars are injected (the parity tests replay the random
 * TERMS the reference itself added), in MODE_PHILOX they come from the counter RNG below. */
#define TTSC_MOL_NMIX 10
#define TTSC_MOL_NOISE 11          /* 10 Gumbel terms + 1 logistic term */
#define TTSC_GM_NOISE 1            /* 0.8 * N(0,1) */
#define TTSC_BETA_TRIES 4
#define TTSC_BETA_NOISE (2 * (1 + 2 * TTSC_BETA_TRIES))   /* per gamma variate: boost uniform, then (normal, uniform) pairs */
#define TTSC_LOG_SCALE_MIN (-32.23619130191664f)          /* float(np.log(1e-14)), loss.py:176 */

/* uniform in [1e-5, 1 - 1e-5], the range loss.py:186,198 draws from */
TTSC_HD float ttsc_u01_clip(uint32_t r) { return fmaf(ttsc_u01(r), 1.0f - 2e-5f, 1e-5f); }
/* logistic noise log(u) - log(1 - u)  (loss.py:199) */
TTSC_HD float ttsc_logistic(float u) { return ttsc_logf(u) - ttsc_logf(1.0f - u); }

/* standard normal quantile (Acklam's rational approximation, |rel err| < 1.2e-9 in exact arithmetic; fp32 here) */
TTSC_HD float ttsc_normal_icdf(float p) {
    const float a1 = -3.969683028665376e+01f, a2 = 2.209460984245205e+02f, a3 = -2.759285104469687e+02f;
    const float a4 = 1.383577518672690e+02f, a5 = -3.066479806614716e+01f, a6 = 2.506628277459239e+00f;
    const float b1 = -5.447609879822406e+01f, b2 = 1.615858368580409e+02f, b3 = -1.556989798598866e+02f;
    const float b4 = 6.680131188771972e+01f, b5 = -1.328068155288572e+01f;
    const float c1 = -7.784894002430293e-03f, c2 = -3.223964580411365e-01f, c3 = -2.400758277161838e+00f;
    const float c4 = -2.549732539343734e+00f, c5 = 4.374664141464968e+00f, c6 = 2.938163982698783e+00f;
    const float d1 = 7.784695709041462e-03f, d2 = 3.224671290700398e-01f, d3 = 2.445134137142996e+00f, d4 = 3.754408661907416e+00f;
    const float plow = 0.02425f;
    if (p < plow) {
        const float q = sqrtf(-2.0f * ttsc_logf(p));
        return fmaf(fmaf(fmaf(fmaf(fmaf(c1, q, c2), q, c3), q, c4), q, c5), q, c6) / fmaf(fmaf(fmaf(fmaf(d1, q, d2), q, d3), q, d4), q, 1.0f);
    }
    if (p > 1.0f - plow) {
        const float q = sqrtf(-2.0f * ttsc_logf(1.0f - p));
        return -fmaf(fmaf(fmaf(fmaf(fmaf(c1, q, c2), q, c3), q, c4), q, c5), q, c6) / fmaf(fmaf(fmaf(fmaf(d1, q, d2), q, d3), q, d4), q, 1.0f);
    }
    const float q = p - 0.5f, r = q * q;
    return fmaf(fmaf(fmaf(fmaf(fmaf(a1, r, a2), r, a3), r, a4), r, a5), r, a6) * q /
           fmaf(fmaf(fmaf(fmaf(fmaf(b1, r, b2), r, b3), r, b4), r, b5), r, 1.0f);
}

/* MOLOutput.sample (loss.py:163-201): Gumbel-max over the mixture logits (first maximum wins), then one draw from the selected
 * logistic, clamped to [-1, 1].  y = the 3*nmix outputs, g = nmix Gumbel terms -log(-log(u)), lg = log(u) - log(1-u). */
TTSC_HD float ttsc_sample_mol(const float* y, const float* g, float lg, int* kout) {
    int k = 0;
    float best = y[0] + g[0];
    for (int i = 1; i < TTSC_MOL_NMIX; ++i) {
        const float v = y[i] + g[i];
        if (v > best) {
            best = v;
            k = i;
        }
    }
    const float mean = y[TTSC_MOL_NMIX + k];
    float ls = y[2 * TTSC_MOL_NMIX + k];
    ls = ls < TTSC_LOG_SCALE_MIN ? TTSC_LOG_SCALE_MIN : ls;
    float x = mean + ttsc_expf(ls) * lg;
    x = x < -1.0f ? -1.0f : x;
    x = x > 1.0f ? 1.0f : x;
    *kout = k;
    return x;
}

/* GaussianOutput.sample (loss.py:50-52): mean + z * exp(log_std), z = 0.8 * N(0,1) (the 0.8 is folded into the noise) */
TTSC_HD float ttsc_sample_gm(const float* y, float z08) { return y[0] + z08 * ttsc_expf(y[1]); }

/* Gamma(alpha, 1) by Marsaglia & Tsang (the algorithm behind torch's Beta / Dirichlet sampling, ATen Distributions.h
 * sample_gamma), with a BOUNDED number of rejection rounds fed from nz = [boost uniform, (normal, uniform) x TRIES]; the last
 * round is accepted unconditionally (probability of getting there < 1e-5). */
TTSC_HD float ttsc_gamma_mt(float alpha, const float* nz) {
    float scale = 1.0f;
    if (alpha < 1.0f) {
        scale = ttsc_expf(ttsc_logf(1.0f - nz[0]) / alpha);   /* (1 - u)^(1/alpha) */
        alpha += 1.0f;
    }
    const float d = alpha - 1.0f / 3.0f;
    const float c = 1.0f / sqrtf(9.0f * d);
    float v = 1.0f;
    int done = 0;
    /* branch-free form of: for each round { if (y <= 0) continue; v = y^3; if (accepted) break; } — fixed trip count, so the
     * noise array is indexed statically (registers on the device) */
    TTSC_UNROLL
    for (int i = 0; i < TTSC_BETA_TRIES; ++i) {
        const float x = nz[1 + 2 * i];
        const float u = 1.0f - nz[2 + 2 * i];
        const float yv = 1.0f + c * x;
        const float vv = yv * yv * yv;
        const float xx = x * x;
        const float lv = ttsc_logf(vv > 1e-30f ? vv : 1e-30f);
        const int acc = (u < 1.0f - 0.0331f * xx * xx) || (ttsc_logf(u) < 0.5f * xx + d * (1.0f - vv + lv));
        const int take = (!done) && (yv > 0.0f);
        v = take ? vv : v;
        done = done || (take && acc);
    }
    return scale * d * v;
}

/* BetaOutput.sample (loss.py:83-92): (Beta(exp(y0), exp(y1)) - 0.5) * 2 through two gamma variates */
TTSC_HD float ttsc_sample_beta(const float* y, const float* nz) {
    const float ga = ttsc_gamma_mt(ttsc_expf(y[0]), nz);
    const float gb = ttsc_gamma_mt(ttsc_expf(y[1]), nz + 1 + 2 * TTSC_BETA_TRIES);
    float s = ga / (ga + gb);
    s = s < 1.17549435e-38f ? 1.17549435e-38f : s;
    s = s > 0.99999994f ? 0.99999994f : s;
    return (s - 0.5f) * 2.0f;
}

/* MODE_PHILOX noise of the continuous samplers: scalar i of step t, utterance b (counter word 3 = 1 keeps the stream apart from
 * the categorical sampler's) */
TTSC_HD uint32_t ttsc_philox_word(uint32_t i, uint32_t t, uint32_t b, uint64_t seed) {
    uint32_t r4[4];
    ttsc_philox4x32(i >> 2, t, b, 1u, (uint32_t)seed, (uint32_t)(seed >> 32), r4);
    return r4[i & 3];
}
TTSC_HD void ttsc_noise_mol(uint32_t t, uint32_t b, uint64_t seed, float* nz) {
    for (uint32_t i = 0; i < TTSC_MOL_NMIX; ++i) nz[i] = -ttsc_logf(-ttsc_logf(ttsc_u01_clip(ttsc_philox_word(i, t, b, seed))));
    nz[TTSC_MOL_NMIX] = ttsc_logistic(ttsc_u01_clip(ttsc_philox_word(TTSC_MOL_NMIX, t, b, seed)));
}
TTSC_HD void ttsc_noise_gm(uint32_t t, uint32_t b, uint64_t seed, float* nz) {
    nz[0] = 0.8f * ttsc_normal_icdf(ttsc_u01(ttsc_philox_word(0, t, b, seed)));
}
TTSC_HD void ttsc_noise_beta(uint32_t t, uint32_t b, uint64_t seed, float* nz) {
    for (uint32_t v = 0; v < 2; ++v) {
        float* q = nz + v * (1 + 2 * TTSC_BETA_TRIES);
        q[0] = ttsc_u01(ttsc_philox_word(v * 16, t, b, seed));
        for (uint32_t i = 0; i < TTSC_BETA_TRIES; ++i) {
            q[1 + 2 * i] = ttsc_normal_icdf(ttsc_u01(ttsc_philox_word(v * 16 + 1 + 2 * i, t, b, seed)));
            q[2 + 2 * i] = ttsc_u01(ttsc_philox_word(v * 16 + 2 + 2 * i, t, b, seed));
        }
    }
}

#endif /* TTSCUBE_MATH_H */
