/*
 * mz_stream_oracle.c — CPU restatement of the per-game random streams (TEST INFRASTRUCTURE ONLY; see
 * mz_oracle.h): what numpy.random.RandomState(seed) yields for dirichlet([alpha] * A) followed by
 * uniform(lo, hi, size = n) — the draws the reference makes from NumPy's global legacy stream after
 * np.random.seed(seed) (/root/reference monte_carlo_tree_search.py:140, :163).
 *
 * The algorithm lives in a third-party dependency, NumPy (2.2.6 in this image; the legacy generator is
 * frozen by NumPy's stream-compatibility guarantee), restated from its published sources:
 *   numpy/random/src/mt19937/mt19937.c        mt19937_seed (init_genrand), mt19937_gen, tempering
 *   numpy/random/src/legacy/legacy-distributions.c   legacy_double (53-bit double from two words),
 *                                             legacy_standard_exponential, legacy_standard_gamma
 *   numpy/random/mtrand.pyx                   RandomState.dirichlet (gamma draws, * 1 / sum), uniform
 * and, beneath NumPy, libm's log and pow (glibc 2.35 here), which are not restated: include/mzs_crmath.h
 * evaluates both to ~2^-75 and rounds once.  glibc's own results are the correctly rounded ones in all but
 * ~0.07 % of the calls, so this oracle equals NumPy on the uniforms bit for bit and on the Dirichlet rows
 * except where glibc misrounds (pinned by tests/test_crmath.py against NumPy itself).
 */
#include <stdint.h>
#include <string.h>

#include "../include/mzs_crmath.h"

#define MT_N 624
#define MT_M 397

typedef struct mt_state {
    uint32_t key[MT_N];
    int pos;
} mt_state;

static void mt_seed(mt_state* s, uint32_t seed) { /* mt19937_seed */
    for (int i = 0; i < MT_N; ++i) {
        s->key[i] = seed;
        seed = 1812433253u * (seed ^ (seed >> 30)) + (uint32_t)i + 1u;
    }
    s->pos = MT_N;
}

static void mt_gen(mt_state* s) { /* mt19937_gen */
    uint32_t* k = s->key;
    int i;
    for (i = 0; i < MT_N - MT_M; ++i) {
        const uint32_t y = (k[i] & 0x80000000u) | (k[i + 1] & 0x7fffffffu);
        k[i] = k[i + MT_M] ^ (y >> 1) ^ (-(int32_t)(y & 1) & 0x9908b0dfu);
    }
    for (; i < MT_N - 1; ++i) {
        const uint32_t y = (k[i] & 0x80000000u) | (k[i + 1] & 0x7fffffffu);
        k[i] = k[i + (MT_M - MT_N)] ^ (y >> 1) ^ (-(int32_t)(y & 1) & 0x9908b0dfu);
    }
    const uint32_t y = (k[MT_N - 1] & 0x80000000u) | (k[0] & 0x7fffffffu);
    k[MT_N - 1] = k[MT_M - 1] ^ (y >> 1) ^ (-(int32_t)(y & 1) & 0x9908b0dfu);
    s->pos = 0;
}

static uint32_t mt_next(mt_state* s) {
    if (s->pos == MT_N) mt_gen(s);
    uint32_t y = s->key[s->pos++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

static double mt_double(mt_state* s) { /* legacy_double: (a * 2^26 + b) / 2^53 */
    const uint32_t a = mt_next(s) >> 5, b = mt_next(s) >> 6;
    return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
}

static double std_exponential(mt_state* s) { return -mzs_cr_log(1.0 - mt_double(s)); }

static double std_gamma(mt_state* s, double shape) { /* legacy_standard_gamma, shape <= 1 */
    if (shape == 1.0) return std_exponential(s);
    if (shape == 0.0) return 0.0;
    for (;;) {
        const double U = mt_double(s);
        const double V = std_exponential(s);
        if (U <= 1.0 - shape) {
            const double X = mzs_cr_pow(U, 1.0 / shape);
            if (X <= V) return X;
        } else {
            const double Y = -mzs_cr_log((1.0 - U) / shape);
            const double X = mzs_cr_pow(1.0 - shape + shape * Y, 1.0 / shape);
            if (X <= V + Y) return X;
        }
    }
}

/* One game's stream.  dirichlet [A] is written only if train != 0 (monte_carlo_tree_search.py:135-141);
 * alpha <= 1 (larger shapes take NumPy's Marsaglia-Tsang branch, which the device does not generate
 * either: muzero_amd.rng.noise_streams draws those on the host).  Returns 0, or -1 for alpha > 1. */
int mzo_stream(int64_t seed, int A, double alpha, int train, double lo, double hi, int64_t n_uniform,
               double* dirichlet, double* uniform) {
    if (alpha > 1.0) return -1;
    mt_state s;
    mt_seed(&s, (uint32_t)seed);
    if (train) {
        double acc = 0.0;
        for (int a = 0; a < A; ++a) {
            dirichlet[a] = std_gamma(&s, alpha);
            acc += dirichlet[a];
        }
        const double inv = 1.0 / acc;
        for (int a = 0; a < A; ++a) dirichlet[a] = dirichlet[a] * inv;
    }
    const double range = hi - lo;
    for (int64_t i = 0; i < n_uniform; ++i) uniform[i] = lo + range * mt_double(&s);
    return 0;
}

double mzo_cr_log(double x) { return mzs_cr_log(x); }
double mzo_cr_pow(double x, double y) { return mzs_cr_pow(x, y); }
