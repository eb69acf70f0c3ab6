/*
 * detmath.h -- the ARITHMETIC SPECIFICATION of the hot path.
 *
 * Every floating-point value the HIP kernels produce is defined by the functions in this header,
 * which use only IEEE-754 correctly-rounded primitives (add, mul, fma, int<->float conversions and
 * bit manipulation).  The same header compiles as device code (hipcc, gfx950) and as plain C
 * (gcc, the CPU restatement in oracle/exact.c), so GPU and CPU results are bit-identical -- this is
 * what makes "selected-region set bit-identical" a testable statement for a massively parallel
 * reduction (hard part 1).
 *
 * Rules for users of this header:
 *   - compile with -ffp-contract=off (no implicit fusion) and without -ffast-math;
 *   - fused operations are written explicitly with mas_fmaf();
 *   - region / loss accumulators are unsigned fixed-point integers (order-independent sums).
 *
 * exp/log follow the classic Cephes single-precision kernels (range reduction by ln2 split into
 * hi/lo, degree-5 / degree-8 minimax polynomials); accuracy is about 1 ulp, checked against libm in
 * tests/test_detmath.py.  They stand in for ATen's softmax / log in the reference
 * (active_selection/my_bvsb.py:20, utils/loss.py:563,581), whose own f32 bits are not reproducible
 * across thread counts either (tests/test_oracle_golden.py).
 */
#ifndef MULACTSEG_DETMATH_H
#define MULACTSEG_DETMATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define MAS_HD __host__ __device__ __forceinline__
#else
#define MAS_HD static inline
#endif

MAS_HD float mas_fmaf(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

MAS_HD uint32_t mas_f2u(float f) {
    union { float f; uint32_t u; } v;
    v.f = f;
    return v.u;
}
MAS_HD float mas_u2f(uint32_t u) {
    union { float f; uint32_t u; } v;
    v.u = u;
    return v.f;
}

/* 2^n for n in [-126, 127] */
MAS_HD float mas_pow2i(int n) { return mas_u2f((uint32_t)(n + 127) << 23); }

/* exp(x), f32, ~1 ulp.  x < -104 -> 0 ; x > 88.72 -> +inf ; NaN -> NaN.
 * Branch-free on purpose (64-lane waves: a divergent early return costs more than the selects). */
MAS_HD float mas_expf(float x) {
    float xc = (x < -104.0f) ? -104.0f : x;
    xc = (xc > 88.72f) ? 88.72f : xc;
    xc = (x != x) ? 0.0f : xc;
    /* n = round-to-nearest-even(x * log2(e)) via the 1.5*2^23 magic constant */
    float t = mas_fmaf(xc, 1.44269504088896341f, 12582912.0f);
    float n = t - 12582912.0f;
    float r = mas_fmaf(n, -0.693359375f, xc);
    r = mas_fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = mas_fmaf(p, r, 1.3981999507e-3f);
    p = mas_fmaf(p, r, 8.3334519073e-3f);
    p = mas_fmaf(p, r, 4.1665795894e-2f);
    p = mas_fmaf(p, r, 1.6666665459e-1f);
    p = mas_fmaf(p, r, 5.0000001201e-1f);
    float y = mas_fmaf(p, r * r, r) + 1.0f;
    int ni = (int)n;
    /* scale by 2^ni with ONE rounding: the first factor keeps the product normal (exact), the second
     * factor is 1 unless the result is subnormal (ni < -125) or ni = 128 */
    int h = ni < -125 ? -125 : (ni > 127 ? 127 : ni);
    float res = mas_u2f(mas_f2u(y) + ((uint32_t)h << 23)) * mas_pow2i(ni - h);
    res = (x < -104.0f) ? 0.0f : res;
    res = (x > 88.72f) ? mas_u2f(0x7f800000u) : res;
    return (x != x) ? x : res;
}

/* exp(max(x, -86)) for finite x <= 0 -- the softmax argument after max-subtraction.  Same values as
 * mas_expf on [-86, 0]; below -86 it saturates at exp(-86) ~ 4.5e-38, which keeps every result a NORMAL
 * float (the 2^n scaling is a single exponent add) and is invisible to every accumulator: such a
 * probability is < 2^-100 of the row sum, contributes 0 after mas_probq, and cannot change a row sum >= ~1. */
MAS_HD float mas_expf_np(float x) {
    float xc = (x < -86.0f) ? -86.0f : x;
    float t = mas_fmaf(xc, 1.44269504088896341f, 12582912.0f);
    float n = t - 12582912.0f;
    float r = mas_fmaf(n, -0.693359375f, xc);
    r = mas_fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = mas_fmaf(p, r, 1.3981999507e-3f);
    p = mas_fmaf(p, r, 8.3334519073e-3f);
    p = mas_fmaf(p, r, 4.1665795894e-2f);
    p = mas_fmaf(p, r, 1.6666665459e-1f);
    p = mas_fmaf(p, r, 5.0000001201e-1f);
    float y = mas_fmaf(p, r * r, r) + 1.0f;
    /* y * 2^n, exact (n >= -124, y in [0.7, 1.42]).  The bit pattern of t = 1.5*2^23 + n carries n mod 512 in its
     * low 9 bits, so (bits(t) << 23) IS (n << 23) mod 2^32: one shift-add, no float->int conversion. */
    return mas_u2f(mas_f2u(y) + (mas_f2u(t) << 23));
}

/* log(x) for finite x > 0, f32, ~1 ulp. */
MAS_HD float mas_logf(float x) {
    int e = 0;
    if (x < 1.17549435e-38f) {   /* subnormal: rescale exactly */
        x = x * 8388608.0f;
        e = -23;
    }
    uint32_t b = mas_f2u(x);
    e += (int)((b >> 23) & 0xffu) - 126;
    float m = mas_u2f((b & 0x007fffffu) | 0x3f000000u);   /* [0.5, 1) */
    if (m < 0.707106781186547524f) {
        e -= 1;
        m = (m + m) - 1.0f;
    } else {
        m = m - 1.0f;
    }
    float z = m * m;
    float p = 7.0376836292e-2f;
    p = mas_fmaf(p, m, -1.1514610310e-1f);
    p = mas_fmaf(p, m, 1.1676998740e-1f);
    p = mas_fmaf(p, m, -1.2420140846e-1f);
    p = mas_fmaf(p, m, 1.4249322787e-1f);
    p = mas_fmaf(p, m, -1.6668057665e-1f);
    p = mas_fmaf(p, m, 2.0000714765e-1f);
    p = mas_fmaf(p, m, -2.4999993993e-1f);
    p = mas_fmaf(p, m, 3.3333331174e-1f);
    float fe = (float)e;
    float y = (p * m) * z;
    y = mas_fmaf(fe, -2.12194440e-4f, y);
    y = mas_fmaf(z, -0.5f, y);
    float r = m + y;
    return mas_fmaf(fe, 0.693359375f, r);
}

/* Unsigned fixed point: floor(v * 2^FRAC) for finite v in [0, 2^(63-FRAC)); exact (no rounding other
 * than the floor); negative, zero and subnormal inputs give 0.  Written with integer shifts on the f32
 * bit pattern (branch-free) so that host and device agree without relying on float->int64 conversion
 * sequences. */
MAS_HD uint64_t mas_fix(float v, int frac) {
    const uint32_t b = mas_f2u(v);
    const int e = (int)((b >> 23) & 0xffu);
    const uint64_t m = (uint64_t)((b & 0x007fffffu) | 0x00800000u);
    const int sh = e - 150 + frac;                 /* value = m * 2^(e-150) */
    const int l = sh < 0 ? 0 : (sh > 39 ? 39 : sh);
    const int r = sh < 0 ? (-sh > 63 ? 63 : -sh) : 0;
    uint64_t q = (m << l) >> r;
    q = (sh > 39) ? ~(uint64_t)0 : q;              /* saturate (never reached by the accumulators below) */
    return ((b >> 31) | (uint32_t)(e == 0)) ? (uint64_t)0 : q;
}

/* Class-probability quantum of the class-prior pass: round-to-nearest-even of p * 2^23 with p = e * rinv never
 * materialised: t = fma(e, R, 2^23) with R = rinv * 2^23 (exact scaling) lands in [2^23, 2^24] where floats are
 * integers, so q = bits(t) - bits(2^23).  One fma; the constant can be subtracted once per accumulator.  23
 * fractional bits keep a per-thread 32-bit accumulator exact over 511 pixels, the rounding is unbiased, and the
 * quantisation error of the pool-wide class prior (~1e-9) is two orders below the f32 rounding of the
 * reference's own mean. */
#define MAS_PROBQ_BIAS 0x4B000000u
MAS_HD uint32_t mas_probq(float e, float R) { return mas_f2u(mas_fmaf(e, R, 8388608.0f)) - MAS_PROBQ_BIAS; }

#define MAS_SCORE_FRAC 40   /* per-region sum of weighted BvSB: v in (0, 1.0000001], <= 2^23 px/region */
#define MAS_PROB_FRAC  23   /* per-image class-probability sums (mas_probq quanta), <= 2^40 px/image */
#define MAS_LOSS_FRAC  32   /* loss sums: l in [0, 18.5], <= 2^27 selected px / batch */

/* mean = floor(sum / count) * 2^-FRAC rounded once to f32 (count > 0) */
MAS_HD float mas_fixed_mean(uint64_t sum, uint64_t count, int frac) {
    uint64_t q = sum / count;
    /* q < 2^53 always holds for the accumulators above, so the u64 -> f64 conversion is exact */
    union { double d; uint64_t u; } s;
    s.u = (uint64_t)(1023 - frac) << 52;     /* 2^-frac, exact scaling */
    return (float)((double)q * s.d);
}

/* Per-pixel Best-vs-Second-Best margin from the two largest logits (z1 >= z2):
 *   reference: softmax(z/T) -> top-2 -> p2/p1 + 1e-8   (active_selection/my_bvsb.py:20-24)
 *   here:      exp(z2*invT - z1*invT) + 1e-8           (same quantity, softmax denominator cancels) */
MAS_HD float mas_bvsb(float z1, float z2, float invT) {
    float d = (z2 * invT) - (z1 * invT);
    return mas_expf_np(d) + 1e-8f;     /* d <= 0; below -86 the exp term is far below ulp(1e-8) either way */
}

/* (a * w) accumulated into a 128-bit unsigned (hi:lo) -- the single-pass scorer's weighted region sum
 * sum_c class_sum[c] * W31[c] needs up to 93 bits. */
MAS_HD void mas_mac_u64_u32(uint64_t a, uint32_t w, uint64_t* hi, uint64_t* lo) {
    const uint64_t p0 = (a & 0xffffffffu) * (uint64_t)w;     /* < 2^64 */
    const uint64_t p1 = (a >> 32) * (uint64_t)w;             /* < 2^64, weight 2^32 */
    uint64_t l = *lo + p0;
    uint64_t h = *hi + (l < p0 ? 1u : 0u);
    const uint64_t p1lo = p1 << 32;
    l += p1lo;
    h += (l < p1lo ? 1u : 0u) + (p1 >> 32);
    *lo = l;
    *hi = h;
}

/* (hi:lo) >> 31 as u64 (callers guarantee the result fits) */
MAS_HD uint64_t mas_shr31_u128(uint64_t hi, uint64_t lo) { return (hi << 33) | (lo >> 31); }

#endif /* MULACTSEG_DETMATH_H */
