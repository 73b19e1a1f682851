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
#else
#define TTSC_HD static inline
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

#endif /* TTSCUBE_MATH_H */
