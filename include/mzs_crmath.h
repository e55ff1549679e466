/*
 * mzs_crmath.h — log and pow in double-double arithmetic, accurate to ~2^-75 before the final rounding:
 * the correctly rounded double except with probability ~1e-5 per call.
 *
 * Why: the device generates each game's root Dirichlet noise as numpy.random.RandomState(seed).dirichlet
 * does (monte_carlo_tree_search.py:140; NumPy's legacy_standard_gamma calls libm's log and pow).  The third
 * party arithmetic on that path is glibc's libm (2.35 in this image), whose log / pow are correctly
 * rounded in all but 0.06 % / 0.08 % of the calls (measured against decimal arithmetic: see
 * tests/test_crmath.py).  A device libm within 1-2 ulp differs from it in a large fraction of the calls;
 * these functions differ only where glibc itself misses the correctly rounded result.
 *
 * Shared by the HIP kernels (device functions) and the CPU oracle / tests (plain C, gcc).  Needs
 * -ffp-contract=off (error-free transformations) and fma().  Tables: mzs_crmath_tables.h
 * (tools/gen_crmath_tables.py).
 *
 * Domain: positive normal doubles (the sampler's arguments are k / 2^53, 1 - k / 2^53 and O(1) values);
 * results far inside the normal range (no subnormal rounding).
 */
#ifndef MZS_CRMATH_H
#define MZS_CRMATH_H

#if defined(__HIPCC__)
#define MZS_CR_FN __device__ __forceinline__
#define MZS_CR_CONST __device__ const
#else
#define MZS_CR_FN static inline
#define MZS_CR_CONST const
#endif

#include "mzs_crmath_tables.h"

typedef struct mzs_dd {
    double hi, lo;
} mzs_dd;

MZS_CR_FN double mzs_cr_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

MZS_CR_FN long long mzs_cr_bits(double x) {
    long long i;
    __builtin_memcpy(&i, &x, 8);
    return i;
}
MZS_CR_FN double mzs_cr_from_bits(long long i) {
    double x;
    __builtin_memcpy(&x, &i, 8);
    return x;
}

/* error-free transformations */
MZS_CR_FN mzs_dd mzs_two_sum(double a, double b) {
    mzs_dd r;
    r.hi = a + b;
    const double bb = r.hi - a;
    r.lo = (a - (r.hi - bb)) + (b - bb);
    return r;
}
MZS_CR_FN mzs_dd mzs_fast_two_sum(double a, double b) { /* |a| >= |b| or a == 0 */
    mzs_dd r;
    r.hi = a + b;
    r.lo = b - (r.hi - a);
    return r;
}
MZS_CR_FN mzs_dd mzs_two_prod(double a, double b) {
    mzs_dd r;
    r.hi = a * b;
    r.lo = mzs_cr_fma(a, b, -r.hi);
    return r;
}
MZS_CR_FN mzs_dd mzs_dd_add(mzs_dd a, mzs_dd b) {
    mzs_dd s = mzs_two_sum(a.hi, b.hi);
    const mzs_dd t = mzs_two_sum(a.lo, b.lo);
    s.lo += t.hi;
    s = mzs_fast_two_sum(s.hi, s.lo);
    s.lo += t.lo;
    return mzs_fast_two_sum(s.hi, s.lo);
}
MZS_CR_FN mzs_dd mzs_dd_add_d(mzs_dd a, double b) {
    mzs_dd s = mzs_two_sum(a.hi, b);
    s.lo += a.lo;
    return mzs_fast_two_sum(s.hi, s.lo);
}
MZS_CR_FN mzs_dd mzs_dd_mul(mzs_dd a, mzs_dd b) {
    mzs_dd p = mzs_two_prod(a.hi, b.hi);
    p.lo += a.hi * b.lo + a.lo * b.hi;
    return mzs_fast_two_sum(p.hi, p.lo);
}
MZS_CR_FN mzs_dd mzs_dd_mul_d(mzs_dd a, double b) {
    mzs_dd p = mzs_two_prod(a.hi, b);
    p.lo += a.lo * b;
    return mzs_fast_two_sum(p.hi, p.lo);
}

/* log(x) as a double-double, x a positive normal double.
 * x = 2^k z, z in [0.75, 1.5); r = z * invc - 1 exactly (two_prod), |r| < 2^-7, with invc = 1 on the two
 * intervals next to 1.0 so that log(x) = log1p(x - 1) keeps its RELATIVE accuracy there;
 * log(x) = k ln2 + logc + log1p(r), log1p(r) = r - r^2/2 + r^3/3 (double-double) + r^4 P(r) (double). */
MZS_CR_FN mzs_dd mzs_log_dd(double x) {
    const long long ix = mzs_cr_bits(x);
    const long long tmp = ix - 0x3fe8000000000000LL;
    const int i = (int)((tmp >> 45) & 127);
    const long long k = tmp >> 52; /* arithmetic shift */
    const double z = mzs_cr_from_bits(ix - (long long)((unsigned long long)tmp & 0xfff0000000000000ULL));
    const double invc = mzs_cr_log_tab[i][0];
    const mzs_dd p = mzs_two_prod(z, invc);
    const mzs_dd r = mzs_fast_two_sum(p.hi - 1.0, p.lo); /* p.hi - 1 is exact (p.hi within 1 % of 1) */
    const double rh = r.hi;
    /* tail: sum_{j=4..13} (-1)^(j+1) r^j / j, plain double (<= 2^-21 of the result: 2^-74 relative) */
    double t = -1.0 / 14.0;
    t = mzs_cr_fma(t, rh, 1.0 / 13.0);
    t = mzs_cr_fma(t, rh, -1.0 / 12.0);
    t = mzs_cr_fma(t, rh, 1.0 / 11.0);
    t = mzs_cr_fma(t, rh, -1.0 / 10.0);
    t = mzs_cr_fma(t, rh, 1.0 / 9.0);
    t = mzs_cr_fma(t, rh, -1.0 / 8.0);
    t = mzs_cr_fma(t, rh, 1.0 / 7.0);
    t = mzs_cr_fma(t, rh, -1.0 / 6.0);
    t = mzs_cr_fma(t, rh, 1.0 / 5.0);
    t = mzs_cr_fma(t, rh, -1.0 / 4.0);
    const double r2d = rh * rh;
    t = t * (r2d * r2d);
    /* r - r^2/2 + r^3/3 in double-double */
    const mzs_dd r2 = mzs_dd_mul(r, r);
    const mzs_dd r3 = mzs_dd_mul(r2, r);
    mzs_dd third;
    third.hi = MZS_CR_THIRD_HI;
    third.lo = MZS_CR_THIRD_LO;
    mzs_dd half_r2;
    half_r2.hi = -0.5 * r2.hi; /* exact */
    half_r2.lo = -0.5 * r2.lo;
    mzs_dd s = mzs_dd_add(mzs_dd_mul(r3, third), mzs_dd_add_d(half_r2, t));
    s = mzs_dd_add(r, s); /* log1p(r) */
    /* k ln2 + logc: k * LN2_HI is exact (32-bit constant, |k| < 2^11) */
    const double kd = (double)k;
    mzs_dd c;
    c.hi = mzs_cr_log_tab[i][1];
    c.lo = mzs_cr_log_tab[i][2];
    mzs_dd kl = mzs_two_sum(kd * MZS_CR_LN2_HI, kd * MZS_CR_LN2_MID);
    kl.lo += kd * MZS_CR_LN2_LO;
    kl = mzs_fast_two_sum(kl.hi, kl.lo);
    return mzs_dd_add(mzs_dd_add(kl, c), s);
}

/* the double nearest log(x) */
MZS_CR_FN double mzs_cr_log(double x) {
    if (x == 1.0) return 0.0;
    const mzs_dd l = mzs_log_dd(x);
    return l.hi + l.lo;
}

/* exp(l.hi + l.lo) rounded to double, |l| < 700.
 * l = n (ln2 / 32) + u, |u| <= ln2 / 64; exp = 2^(n div 32) * T[n mod 32] * (1 + (u + u^2/2 + u^3 Q(u))) */
MZS_CR_FN double mzs_exp_dd(mzs_dd l) {
    const double nd = __builtin_rint(l.hi * MZS_CR_32_LN2);
    const long long n = (long long)nd;
    /* u = l - n c, c = c_hi + c_mid + c_lo; n * c_hi exact (36-bit constant), l.hi - n c_hi exact (Sterbenz) */
    const double a = l.hi - nd * MZS_CR_LN2_32_HI;
    mzs_dd m = mzs_two_prod(nd, MZS_CR_LN2_32_MID);
    m.lo += nd * MZS_CR_LN2_32_LO;
    m.hi = -m.hi;
    m.lo = -m.lo;
    mzs_dd u = mzs_dd_add(mzs_two_sum(a, l.lo), m);
    const double uh = u.hi;
    /* tail: u^3 (1/6 + u/24 + ... + u^6/9!), plain double (<= 2^-22 of the result) */
    double q = 1.0 / 362880.0;
    q = mzs_cr_fma(q, uh, 1.0 / 40320.0);
    q = mzs_cr_fma(q, uh, 1.0 / 5040.0);
    q = mzs_cr_fma(q, uh, 1.0 / 720.0);
    q = mzs_cr_fma(q, uh, 1.0 / 120.0);
    q = mzs_cr_fma(q, uh, 1.0 / 24.0);
    q = mzs_cr_fma(q, uh, 1.0 / 6.0);
    q = q * (uh * uh * uh);
    mzs_dd u2 = mzs_dd_mul(u, u);
    u2.hi *= 0.5;
    u2.lo *= 0.5;
    const mzs_dd s = mzs_dd_add(u, mzs_dd_add_d(u2, q)); /* exp(u) - 1 */
    const int j = (int)(n & 31);
    const long long e = (n - j) / 32;
    mzs_dd T;
    T.hi = mzs_cr_exp_tab[j][0];
    T.lo = mzs_cr_exp_tab[j][1];
    const mzs_dd y = mzs_dd_add(T, mzs_dd_mul(T, s));
    const double v = y.hi + y.lo;
    /* scale by 2^e in two exact steps (|e| <= 1100) */
    const long long e1 = e / 2, e2 = e - e1;
    return (v * mzs_cr_from_bits((e1 + 1023) << 52)) * mzs_cr_from_bits((e2 + 1023) << 52);
}

/* the double nearest x^y for x >= 0, y > 0 (legacy_standard_gamma: pow(U, 1 / shape)) */
MZS_CR_FN double mzs_cr_pow(double x, double y) {
    if (x == 0.0) return 0.0;
    if (x == 1.0) return 1.0;
    return mzs_exp_dd(mzs_dd_mul_d(mzs_log_dd(x), y));
}

#endif /* MZS_CRMATH_H */
