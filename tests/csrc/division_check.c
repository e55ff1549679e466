/* division_check.c — host check of the reciprocal + fma-residual divisions used by
 * muzero_amd/csrc/mzs_quad.h (div_r1 / div_r2).  Test infrastructure only.
 *   part 1 (exhaustive): every fp32 significand x every integer divisor 1..512, ONE correction
 *                        step == correctly rounded a / d  (exponent scaling is exact)
 *   part 2 (sampled):    two steps, fp32 with arbitrary divisors and fp64 with integer divisors
 * Prints "mismatches <n1> <n2> <n3>"; exit status 0 iff all zero. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline uint64_t rng(uint64_t* s) {
    uint64_t x = *s;
    x ^= x << 13, x ^= x >> 7, x ^= x << 17;
    return *s = x;
}
static inline float div32_1(float a, float d, float r) {
    float q = a * r;
    return fmaf(fmaf(-q, d, a), r, q);
}
static inline float div32_2(float a, float d, float r) {
    float q = div32_1(a, d, r);
    return fmaf(fmaf(-q, d, a), r, q);
}
static inline double div64_2(double a, double d, double r) {
    double q = a * r;
    q = fma(fma(-q, d, a), r, q);
    return fma(fma(-q, d, a), r, q);
}

int main(int argc, char** argv) {
    long samples = argc > 1 ? atol(argv[1]) : 20000000L;
    long bad1 = 0, bad2 = 0, bad3 = 0;
#pragma omp parallel for reduction(+ : bad1)
    for (int d = 1; d <= 512; d++) {
        float df = (float)d, r = 1.0f / df;
        for (uint32_t m = 0; m < (1u << 23); m++) {
            uint32_t bits = (127u << 23) | m;
            float a;
            memcpy(&a, &bits, 4);
            if (div32_1(a, df, r) != a / df) bad1++;
        }
    }
    uint64_t s = 88172645463325252ULL;
    for (long i = 0; i < samples; i++) {
        uint64_t x = rng(&s);
        uint32_t bits = (uint32_t)x, ex = 64 + ((bits >> 23) & 0xff) % 128;
        bits = (bits & 0x807fffffu) | (ex << 23);
        float a;
        memcpy(&a, &bits, 4);
        uint32_t b2 = (uint32_t)(x >> 17), e2 = 100 + ((b2 >> 23) & 0xff) % 56;
        b2 = (b2 & 0x007fffffu) | (e2 << 23);
        float g;
        memcpy(&g, &b2, 4);
        float tg = a / g;
        if (isfinite(tg) && fabsf(tg) > 1e-30f && div32_2(a, g, 1.0f / g) != tg) bad2++;
        uint64_t y = rng(&s), e3 = 900 + ((y >> 52) & 0x7ff) % 250;
        y = (y & 0x800fffffffffffffULL) | (e3 << 52);
        double A;
        memcpy(&A, &y, 8);
        double D = (double)(1 + (int)((x >> 40) % 512));
        if (div64_2(A, D, 1.0 / D) != A / D) bad3++;
    }
    printf("mismatches %ld %ld %ld\n", bad1, bad2, bad3);
    return (bad1 || bad2 || bad3) ? 1 : 0;
}
