/*
 * mz_selfplay_oracle.c — CPU oracle of the batched self-play bookkeeping (test infrastructure,
 * see mz_oracle.h).  Restates, one game at a time:
 *   Game.store_search_statistics   game.py:154-168
 *   Game.policy_step               game.py:176-225 (softmax_stable :170-173, np.random.choice's
 *                                  cdf / searchsorted, argmax below temperature 0.1)
 *   play_game's loop condition     self_play.py:62
 *   ReplayBuffer.sample_batch      replay_buffer.py:147-151 (make_image, fill_gap_empty_action)
 *   Game.make_target               game.py:237-260 (incl. its fp32 / fp64 accumulation under NumPy 2)
 * Pinned to the reference's own Game class by tests/test_selfplay_oracle.py (game.py imported in
 * the build container with torchvision stubbed).  The CartPole physics and the counter-based
 * random numbers live in include/mzs_env.h (not in the reference; physics parity unpinned).
 */
#include <math.h>
#include <string.h>

#include "mz_oracle.h"
#include "../include/mzs_env.h"

double mzo_cartpole_step(double* s, int32_t action, int32_t* terminated) {
    mzs_cartpole st = {s[0], s[1], s[2], s[3]};
    *terminated = mzs_cartpole_step(&st, action);
    s[0] = st.x, s[1] = st.x_dot, s[2] = st.theta, s[3] = st.theta_dot;
    return 1.0;
}

/* Game.store_search_statistics (game.py:154-168): the policy target of one move, as the fp32
 * values reshape_batch later makes of it */
void mzo_store_policy(int32_t A, const int32_t* v, const double* prior, float* pol) {
    double vsum = 0.0;
    for (int i = 0; i < A; ++i) vsum += (double)v[i];
    if (vsum >= 3.0) {
        for (int i = 0; i < A; ++i) pol[i] = (float)((double)v[i] / vsum);
    } else {
        double psum = 0.0;
        for (int i = 0; i < A; ++i) psum += prior[i];
        for (int i = 0; i < A; ++i) pol[i] = (float)(prior[i] / psum);
    }
}

/* Game.policy_step's action (game.py:184-188) from the policy MCTS.run returns (the visit
 * counts, or the root priors when num_simulations == 0: monte_carlo_tree_search.py:218-221)
 * and the uniform number np.random.choice would draw */
int32_t mzo_policy_action(int32_t A, const double* v, double temperature, double u) {
    double tsum = 0.0;
    int all_equal = 1;
    for (int i = 0; i < A; ++i) {
        tsum += mzs_policy_power((double)v[i], temperature);
        all_equal = all_equal && v[i] == v[0];
    }
    int act = 0;
    if (temperature > 0.1 || all_equal) {
        /* RandomState.choice: cdf = p.cumsum(); cdf /= cdf[-1]; cdf.searchsorted(u, side='right') */
        double cl = 0.0;
        for (int i = 0; i < A; ++i) cl += mzs_policy_power((double)v[i], temperature) / tsum;
        double c = 0.0;
        act = A - 1;
        for (int i = 0; i < A; ++i) {
            c += mzs_policy_power((double)v[i], temperature) / tsum;
            if (u < c / cl) {
                act = i;
                break;
            }
        }
    } else {
        double best = mzs_policy_power((double)v[0], temperature) / tsum;
        for (int i = 1; i < A; ++i) {
            const double p = mzs_policy_power((double)v[i], temperature) / tsum;
            if (p > best) best = p, act = i; /* np.argmax: first maximum */
        }
    }
    return act;
}

static int64_t stream_seed(int64_t seed, int32_t episode, int32_t move) {
    return (seed + 0x9E3779B1LL * (int64_t)(move + 1) + 0x85EBCA77LL * (int64_t)episode) & 0xFFFFFFFFLL;
}

static void start_game(mzo_arena* ar, int g, int vrow) {
    const int row = vrow % ar->n_rows;
    mzs_cartpole st;
    mzs_cartpole_reset(&st, (uint64_t)ar->row_seed[row], (uint32_t)ar->row_episode[row]);
    double* s = ar->state + (size_t)g * 4;
    s[0] = st.x, s[1] = st.x_dot, s[2] = st.theta, s[3] = st.theta_dot;
    for (int j = 0; j < 4; ++j) ar->obs[(size_t)g * 4 + j] = (float)s[j];
    ar->alive[g] = 1;
    ar->row[g] = row;
    ar->vrow[g] = vrow;
    ar->stream_seed[g] = stream_seed(ar->row_seed[row], ar->row_episode[row], 0);
}

/* every game below this virtual row has finished: the queue's head, or the oldest game still running */
static void update_done(mzo_arena* ar) {
    int d = ar->next_row;
    for (int g = 0; g < ar->G; ++g)
        if (ar->alive[g] && ar->vrow[g] < d) d = ar->vrow[g];
    ar->done_upto = d;
}

/* a round of n_games episodes in rows [row0, row0 + n_games): the first min(G, n_games) start now */
void mzo_arena_begin(mzo_arena* ar, int32_t row0, int32_t n_games, const int64_t* seeds, int32_t episode) {
    ar->episode = episode;
    for (int i = 0; i < n_games; ++i) {
        ar->row_seed[row0 + i] = seeds[i];
        ar->length[row0 + i] = 0;
        ar->row_episode[row0 + i] = episode;
        ar->row_temp[row0 + i] = NAN;
    }
    ar->next_row = row0 + (n_games < ar->G ? n_games : ar->G);
    ar->row_end = row0 + n_games;
    for (int g = 0; g < ar->G; ++g) {
        if (g < n_games)
            start_game(ar, g, row0 + g);
        else
            ar->alive[g] = 0;
    }
    update_done(ar);
}

/* more games behind the queue's end (they start as slots free up, from the next step on) */
void mzo_arena_enqueue(mzo_arena* ar, int32_t n_games, const int64_t* seeds, int32_t episode, double temperature) {
    for (int i = 0; i < n_games; ++i) {
        const int row = (ar->row_end + i) % ar->n_rows;
        ar->row_seed[row] = seeds[i];
        ar->length[row] = 0;
        ar->row_episode[row] = episode;
        ar->row_temp[row] = temperature;
    }
    ar->row_end += n_games;
}

/* one move of every live slot, then free slots take the next unplayed games in slot order;
 * returns the number of slots with a running game */
int32_t mzo_arena_step(mzo_arena* ar, const int32_t* visits, const float* root_value, const double* root_prior,
                       double temperature, int32_t from_prior) {
    const int T = ar->T, A = ar->A;
    for (int g = 0; g < ar->G; ++g) {
        if (!ar->alive[g]) continue;
        const int row = ar->row[g], t = ar->length[row];
        const int32_t* v = visits + (size_t)g * A;
        mzo_store_policy(A, v, root_prior + (size_t)g * A, ar->traj_policy + ((size_t)row * T + t) * A);
        ar->traj_root[(size_t)row * T + t] = root_value[g];
        const double u = mzs_uniform01((uint64_t)ar->row_seed[row], (uint32_t)ar->row_episode[row], (uint32_t)t, MZS_RNG_ACTION);
        double pv[64];
        for (int i = 0; i < A; ++i) pv[i] = from_prior ? root_prior[(size_t)g * A + i] : (double)v[i];
        const double temp = isnan(ar->row_temp[row]) ? temperature : ar->row_temp[row];
        const int act = mzo_policy_action(A, pv, temp, u);
        double* s = ar->state + (size_t)g * 4;
        mzs_cartpole st = {s[0], s[1], s[2], s[3]};
        const int terminated = mzs_cartpole_step(&st, act);
        s[0] = st.x, s[1] = st.x_dot, s[2] = st.theta, s[3] = st.theta_dot;
        for (int j = 0; j < 4; ++j) {
            ar->obs[(size_t)g * 4 + j] = (float)s[j];
            ar->traj_obs[((size_t)row * T + t) * 4 + j] = (float)s[j];
        }
        ar->traj_action[(size_t)row * T + t] = act;
        ar->traj_reward[(size_t)row * T + t] = 1.0f;
        ar->length[row] = t + 1;
        ar->alive[g] = !terminated && t + 1 < T; /* self_play.py:62 */
        ar->stream_seed[g] = stream_seed(ar->row_seed[row], ar->row_episode[row], t + 1);
    }
    int32_t n_alive = 0;
    for (int g = 0; g < ar->G; ++g) {
        if (!ar->alive[g] && ar->next_row < ar->row_end) start_game(ar, g, ar->next_row++);
        n_alive += ar->alive[g];
    }
    update_done(ar);
    return n_alive;
}

static double target_value(const float* rw, const float* rootv, int len, int ci, int td, const double* disc_pow) {
    const int bi = ci + td, hi = bi < len ? bi : len;
    if (bi < len) {
        float v = rootv[bi] * (float)disc_pow[td];
        for (int i = 0; ci + i < hi; ++i) v = v + (float)((double)rw[ci + i] * disc_pow[i]);
        return (double)v;
    }
    double v = 0.0;
    for (int i = 0; ci + i < hi; ++i) v = v + (double)rw[ci + i] * disc_pow[i];
    return v; /* rounded to fp32 by make_target's consumer (reshape_batch), kept by make_priority */
}

/* Game.make_priority (game.py:262-284): out [n_rows][T] */
void mzo_make_priority(int32_t n_rows, int32_t T, const int32_t* length, const float* traj_reward,
                       const float* traj_root, int32_t td, const double* disc_pow, double* out) {
    for (int r = 0; r < n_rows; ++r)
        for (int pos = 0; pos < T; ++pos) {
            double v = 0.0;
            if (pos < length[r]) {
                const double t = target_value(traj_reward + (size_t)r * T, traj_root + (size_t)r * T, length[r],
                                             pos, td, disc_pow);
                v = (double)traj_root[(size_t)r * T + pos] - t;
                v = v < 0 ? -v : v;
            }
            out[(size_t)r * T + pos] = v;
        }
}

void mzo_make_batch(int32_t T, int32_t A, int32_t obs_dim, const int32_t* length, const float* traj_obs,
                    const int32_t* traj_action, const float* traj_reward, const float* traj_root,
                    const float* traj_policy, int32_t B, const int32_t* row_pos, int32_t K, int32_t td,
                    const double* disc_pow, float* obs, float* actions, float* tv, float* tr, float* tp) {
    for (int b = 0; b < B; ++b) {
        const int row = row_pos[2 * b], pos = row_pos[2 * b + 1], len = length[row];
        memcpy(obs + (size_t)b * obs_dim, traj_obs + ((size_t)row * T + pos) * obs_dim,
               sizeof(float) * (size_t)obs_dim);
        for (int k = 0; k < K; ++k) {
            const int ci = pos + k;
            const int act = ci < len ? traj_action[(size_t)row * T + ci] : -1;
            for (int j = 0; j < A; ++j) actions[((size_t)k * B + b) * A + j] = j == act ? 1.0f : 0.0f;
            tr[(size_t)k * B + b] = (ci > 0 && ci <= len) ? traj_reward[(size_t)row * T + ci - 1] : 0.0f;
            if (ci < len) {
                tv[(size_t)k * B + b] = (float)target_value(traj_reward + (size_t)row * T, traj_root + (size_t)row * T,
                                                     len, ci, td, disc_pow);
                for (int j = 0; j < A; ++j)
                    tp[((size_t)k * B + b) * A + j] = traj_policy[((size_t)row * T + ci) * A + j];
            } else {
                tv[(size_t)k * B + b] = 0.0f;
                for (int j = 0; j < A; ++j) tp[((size_t)k * B + b) * A + j] = 0.0f;
            }
        }
    }
}
