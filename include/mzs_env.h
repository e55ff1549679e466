/*
 * mzs_env.h — arithmetic shared by the self-play kernels (muzero_amd/csrc/mzs_selfplay.h) and
 * their CPU oracle (oracle/mz_selfplay_oracle.c): the batched CartPole-v1 environment, the
 * counter-based random numbers of the batched driver, and the scalar pieces of
 * Game.store_search_statistics / Game.policy_step / Game.make_target.
 *
 * Every operation is an individually rounded IEEE-754 op (compile with -ffp-contract=off; fused
 * multiply-adds only where written), so GPU and oracle agree bit for bit.
 *
 * Reference code restated (file:line in /root/reference):
 *   game.py:154-168   store_search_statistics      game.py:170-188  softmax_stable / policy_step
 *   game.py:237-260   make_target (dtype behaviour under NumPy 2 noted at mzs_target_value)
 * The environment itself is NOT in the reference (it calls gymnasium, which is not installed in
 * the build container): mzs_cartpole_step restates Gymnasium's CartPole-v1 (classic_control/
 * cartpole.py: Euler integrator, force 10 N, tau 0.02 s, termination at |x| > 2.4 or
 * |theta| > 12 degrees, reward 1 per step) from its published equations — parity unpinned.
 */
#ifndef MZS_ENV_H
#define MZS_ENV_H

#include <stdint.h>
#if !defined(__HIPCC__)
#include <math.h>
#endif

#include "mzs_numeric.h"

/* ---------------------------------------------------------------- random numbers
 * The reference draws environment resets from the env's own generator and action samples from
 * NumPy's global stream, one game at a time; thousands of concurrent games cannot share one
 * sequential stream, so the batched driver uses a counter-based generator instead: the k-th
 * number of (seed, episode, move, purpose) is splitmix64 of a key, no state to carry. */
MZS_FN uint64_t mzs_splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

enum { MZS_RNG_RESET = 0, MZS_RNG_ACTION = 4 }; /* RESET uses purposes 0..3 (one per state variable) */

MZS_FN double mzs_uniform01(uint64_t seed, uint32_t episode, uint32_t move, uint32_t purpose) {
    uint64_t k = mzs_splitmix64(seed);
    k = mzs_splitmix64(k ^ (((uint64_t)episode << 32) | move));
    k = mzs_splitmix64(k + purpose);
    return (double)(k >> 11) * (1.0 / 9007199254740992.0);
}

/* ---------------------------------------------------------------- CartPole-v1 */
typedef struct mzs_cartpole {
    double x, x_dot, theta, theta_dot;
} mzs_cartpole;

/* sin and cos for |t| <= 1 (the pole angle never leaves +-0.3 rad before the episode ends):
 * Taylor polynomials in t^2, Horner with fma, truncation error < 1e-17 at |t| = 0.5 */
MZS_FN void mzs_sincos_small(double t, double* s, double* c) {
    const double z = t * t;
    double ps = -1.0 / 1307674368000.0;                 /* -1/15! */
    ps = __builtin_fma(ps, z, 1.0 / 6227020800.0);      /*  1/13! */
    ps = __builtin_fma(ps, z, -1.0 / 39916800.0);       /* -1/11! */
    ps = __builtin_fma(ps, z, 1.0 / 362880.0);          /*  1/9!  */
    ps = __builtin_fma(ps, z, -1.0 / 5040.0);           /* -1/7!  */
    ps = __builtin_fma(ps, z, 1.0 / 120.0);             /*  1/5!  */
    ps = __builtin_fma(ps, z, -1.0 / 6.0);              /* -1/3!  */
    ps = __builtin_fma(ps, z, 1.0);
    *s = t * ps;
    double pc = 1.0 / 20922789888000.0;                 /*  1/16! */
    pc = __builtin_fma(pc, z, -1.0 / 87178291200.0);    /* -1/14! */
    pc = __builtin_fma(pc, z, 1.0 / 479001600.0);       /*  1/12! */
    pc = __builtin_fma(pc, z, -1.0 / 3628800.0);        /* -1/10! */
    pc = __builtin_fma(pc, z, 1.0 / 40320.0);           /*  1/8!  */
    pc = __builtin_fma(pc, z, -1.0 / 720.0);            /* -1/6!  */
    pc = __builtin_fma(pc, z, 1.0 / 24.0);              /*  1/4!  */
    pc = __builtin_fma(pc, z, -0.5);
    *c = __builtin_fma(pc, z, 1.0);
}

/* reset: four uniforms in [-0.05, 0.05) */
MZS_FN void mzs_cartpole_reset(mzs_cartpole* st, uint64_t seed, uint32_t episode) {
    st->x = -0.05 + 0.1 * mzs_uniform01(seed, episode, 0, MZS_RNG_RESET + 0);
    st->x_dot = -0.05 + 0.1 * mzs_uniform01(seed, episode, 0, MZS_RNG_RESET + 1);
    st->theta = -0.05 + 0.1 * mzs_uniform01(seed, episode, 0, MZS_RNG_RESET + 2);
    st->theta_dot = -0.05 + 0.1 * mzs_uniform01(seed, episode, 0, MZS_RNG_RESET + 3);
}

/* one env.step(action): returns 1 when the episode terminated; the reward is always 1.0 */
MZS_FN int mzs_cartpole_step(mzs_cartpole* st, int action) {
    const double gravity = 9.8, masspole = 0.1, total_mass = 1.1, length = 0.5;
    const double polemass_length = 0.05, force_mag = 10.0, tau = 0.02;
    const double force = action == 1 ? force_mag : -force_mag;
    double sn, cs;
    mzs_sincos_small(st->theta, &sn, &cs);
    const double temp = (force + polemass_length * (st->theta_dot * st->theta_dot) * sn) / total_mass;
    const double thetaacc =
        (gravity * sn - cs * temp) / (length * (4.0 / 3.0 - masspole * (cs * cs) / total_mass));
    const double xacc = temp - polemass_length * thetaacc * cs / total_mass;
    st->x = st->x + tau * st->x_dot;
    st->x_dot = st->x_dot + tau * xacc;
    st->theta = st->theta + tau * st->theta_dot;
    st->theta_dot = st->theta_dot + tau * thetaacc;
    const double x_threshold = 2.4, theta_threshold = 12.0 * 2.0 * 3.141592653589793 / 360.0;
    return st->x < -x_threshold || st->x > x_threshold || st->theta < -theta_threshold ||
           st->theta > theta_threshold;
}

/* ---------------------------------------------------------------- Game.policy_step pieces */

/* softmax_stable's power (game.py:171-172): policy ** (1 / temperature) for temperature >= 0.3.
 * Integer exponents (T = 1, 0.5, ...) are evaluated by repeated multiplication so that GPU and
 * oracle agree bit for bit; NumPy's pow may differ from that by an ulp, which only perturbs a
 * sampling probability by 1e-16. */
MZS_FN double mzs_policy_power(double p, double temperature) {
    if (temperature < 0.3) return p;
    const double e = 1.0 / temperature;
    const int n = (int)e;
    if ((double)n == e && n >= 1 && n <= 8) {
        double r = p;
        for (int i = 1; i < n; ++i) r = r * p;
        return r;
    }
    return pow(p, e);
}

#endif /* MZS_ENV_H */
