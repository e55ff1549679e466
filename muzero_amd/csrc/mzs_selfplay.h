/*
 * mzs_selfplay.h — the callers on either side of the search, batched on the GPU (SURVEY.md §8f
 * N1-N3): G concurrent CartPole episodes stepped by a kernel, the reference's per-move
 * bookkeeping (Game.store_search_statistics, Game.policy_step, game.py:154-225) applied to the
 * search results where they already are (HBM), trajectories kept in HBM, and the training
 * batch (ReplayBuffer.sample_batch's gather + Game.make_target, replay_buffer.py:131-166,
 * game.py:237-260) assembled by a kernel straight into the buffers mzs_train_grad reads.
 *
 * One thread per game / per (sample, unroll step): this is HBM-bound gather / scatter work of a
 * few bytes per item, two orders of magnitude below the search in cost; lanes are laid out so
 * that neighbouring threads touch neighbouring rows.  Scalar arithmetic: include/mzs_env.h.
 * Included by mzsearch.hip (one library).
 */
#ifndef MZS_SELFPLAY_H
#define MZS_SELFPLAY_H

#include <hip/hip_runtime.h>

#include "../../include/mzs_env.h"

namespace mzs {

struct ArenaArrays {
    int G, n_rows, T, obs_dim, A;
    /* per slot (one running episode each) */
    double* state;          /* [G][4] CartPole state                                        */
    int* alive;             /* [G]    1 while the slot's episode runs                       */
    int* row;               /* [G]    trajectory row the slot writes                        */
    long long* seed;        /* [G]    seed of the slot's current game                       */
    int* episode;           /* [G]    round counter (random-number key)                     */
    long long* stream_seed; /* [G]    RandomState seed of the slot's next search streams    */
    int* vrow;              /* [G]    virtual row of the slot's game (row = vrow % n_rows)   */
    int* ctl;               /* [4]    VIRTUAL rows (they never wrap): next unplayed game, end of the queue,
                                      and the row every game below which has finished; [3] = lock-step
                                      moves since mzs_arena_begin                               */
    int* status_ring;       /* [kArenaRing][8] pinned HOST memory: what move m left behind, written by the
                                      move's last kernel: {m, ctl[0..2], alive slots, max draws consumed, m, 0} */
    /* per trajectory row (replaces the Game object's lists, game.py:46-55) */
    long long* row_seed;    /* [n_rows]          seed of the game recorded in the row       */
    int* row_episode;       /* [n_rows]          random-number key: the round the game was queued for */
    double* row_temp;       /* [n_rows]          action-selection temperature of the game; NaN: the step's */
    int* length;            /* [n_rows]          game_length                                */
    float* obs;             /* [n_rows][T][obs]  observations (state AFTER move t, as Game stores it) */
    int* action;            /* [n_rows][T]       action_history (index; one-hot when gathered) */
    float* reward;          /* [n_rows][T]       rewards                                    */
    float* root_value;      /* [n_rows][T]       root_values                                */
    float* policy;          /* [n_rows][T][A]    child_visits                               */
};

constexpr int kArenaRing = 256; /* moves whose status entries stay readable */

/* seed of the (game, round, move) stream the search draws its Dirichlet noise and tie-break
 * uniforms from (mzs_fill_streams seeds a NumPy RandomState with it) */
__device__ __forceinline__ long long arena_stream_seed(long long seed, int episode, int move) {
    return (seed + 0x9E3779B1LL * (long long)(move + 1) + 0x85EBCA77LL * (long long)episode) & 0xFFFFFFFFLL;
}

/* Game() + Game.observation(iteration=0) (game.py:126-141) for slot g, recording into `row` */
__device__ __forceinline__ void arena_start_game(const ArenaArrays& a, int g, int vrow, float* obs_io) {
    const int row = vrow % a.n_rows;
    const long long seed = a.row_seed[row];
    const int episode = a.row_episode[row];
    mzs_cartpole st;
    mzs_cartpole_reset(&st, (uint64_t)seed, (uint32_t)episode);
    double* s = a.state + (size_t)g * 4;
    s[0] = st.x, s[1] = st.x_dot, s[2] = st.theta, s[3] = st.theta_dot;
    a.alive[g] = 1;
    a.row[g] = row;
    a.vrow[g] = vrow;
    a.seed[g] = seed;
    a.episode[g] = episode;
    a.stream_seed[g] = arena_stream_seed(seed, episode, 0);
    float* o = obs_io + (size_t)g * a.obs_dim;
    o[0] = (float)st.x, o[1] = (float)st.x_dot, o[2] = (float)st.theta, o[3] = (float)st.theta_dot;
}

/* Start of a round of n_games episodes recorded in rows [row0, row0 + n_games): the first
 * min(G, n_games) start now, the rest when a slot frees up (k_arena_refill). */
__global__ void __launch_bounds__(256) k_arena_rows(const ArenaArrays a, int vrow0, int n_games,
                                                    const long long* seeds, int episode, double temperature) {
    /* queue entries of n_games games from virtual row vrow0 on */
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
    for (int i = tid; i < n_games; i += nth) {
        const int row = (vrow0 + i) % a.n_rows;
        a.row_seed[row] = seeds[i];
        a.length[row] = 0;
        a.row_episode[row] = episode;
        a.row_temp[row] = temperature;
    }
}

/* after k_arena_rows: reset the queue to [row0, row0 + n_games) and start its first min(G, n_games) games */
__global__ void __launch_bounds__(256) k_arena_begin(const ArenaArrays a, int row0, int n_games, int episode,
                                                     float* obs_out) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
    if (tid == 0) {
        const int started = n_games < a.G ? n_games : a.G;
        a.ctl[0] = row0 + started;
        a.ctl[1] = row0 + n_games;
        a.ctl[2] = row0; /* nothing has finished yet */
        a.ctl[3] = 0;
    }
    for (int g = tid; g < a.G; g += nth) {
        if (g < n_games) {
            arena_start_game(a, g, row0 + g, obs_out);
        } else {
            a.alive[g] = 0;
            a.episode[g] = episode;
        }
    }
}

/* mzs_arena_enqueue: the rows were written by k_arena_rows; move the queue's end */
__global__ void k_arena_enqueue(const ArenaArrays a, int n_games) {
    if (blockIdx.x == 0 && threadIdx.x == 0) a.ctl[1] += n_games;
}

/* One move of every live game: store_search_statistics (game.py:154-168), policy_step
 * (game.py:176-225: temperature rescale, sample or argmax, env.step, append to the lists).
 * visits / root_value / root_prior are the search results of this move (mzs_results layout). */
__device__ __forceinline__ void arena_step_slot(const ArenaArrays& a, int g, const int* visits,
                                                const float* root_value, const double* root_prior,
                                                double temperature, int from_prior, float* obs_io) {
    const int A = a.A, row = a.row[g], t = a.length[row];
    const int* v = visits + (size_t)g * A;
    /* store_search_statistics: visit fractions, or the normalised priors below 3 visits */
    double vsum = 0.0;
    for (int i = 0; i < A; ++i) vsum += (double)v[i];
    float* pol = a.policy + ((size_t)row * a.T + t) * A;
    if (vsum >= 3.0) {
        for (int i = 0; i < A; ++i) pol[i] = (float)((double)v[i] / vsum);
    } else {
        double psum = 0.0;
        for (int i = 0; i < A; ++i) psum += root_prior[(size_t)g * A + i];
        for (int i = 0; i < A; ++i) pol[i] = (float)(root_prior[(size_t)g * A + i] / psum);
    }
    a.root_value[(size_t)row * a.T + t] = root_value[g];
    /* policy_step: softmax_stable(policy, temperature), then choice or argmax; the policy is what
     * MCTS.run returns: visit counts, or the root priors when num_simulations == 0
     * (monte_carlo_tree_search.py:218-221) */
    const double* pr = root_prior + (size_t)g * A;
    const double p0 = from_prior ? pr[0] : (double)v[0];
    double tsum = 0.0;
    bool all_equal = true;
    const double rt = a.row_temp[row];
    if (rt == rt) temperature = rt; /* the game's own temperature (mzs_arena_enqueue); NaN: this move's argument */
    for (int i = 0; i < A; ++i) {
        const double pi = from_prior ? pr[i] : (double)v[i];
        tsum += mzs_policy_power(pi, temperature);
        all_equal = all_equal && pi == p0;
    }
    int act = 0;
    if (temperature > 0.1 || all_equal) {
        /* np.random.choice(action, p): cdf = cumsum(p); cdf /= cdf[-1]; searchsorted(u, 'right') */
        const double u = mzs_uniform01((uint64_t)a.seed[g], (uint32_t)a.episode[g], (uint32_t)t, MZS_RNG_ACTION);
        double cl = 0.0;
        for (int i = 0; i < A; ++i)
            cl += mzs_policy_power(from_prior ? pr[i] : (double)v[i], temperature) / tsum;
        double c = 0.0;
        act = A - 1;
        for (int i = 0; i < A; ++i) {
            c += mzs_policy_power(from_prior ? pr[i] : (double)v[i], temperature) / tsum;
            if (u < c / cl) {
                act = i;
                break;
            }
        }
    } else {
        double best = mzs_policy_power(p0, temperature) / tsum;
        for (int i = 1; i < A; ++i) {
            const double p = mzs_policy_power(from_prior ? pr[i] : (double)v[i], temperature) / tsum;
            if (p > best) best = p, act = i; /* np.argmax: first maximum */
        }
    }
    /* env.step */
    double* s = a.state + (size_t)g * 4;
    mzs_cartpole st = {s[0], s[1], s[2], s[3]};
    const int terminated = mzs_cartpole_step(&st, act);
    s[0] = st.x, s[1] = st.x_dot, s[2] = st.theta, s[3] = st.theta_dot;
    float* o = obs_io + (size_t)g * a.obs_dim;
    o[0] = (float)st.x, o[1] = (float)st.x_dot, o[2] = (float)st.theta, o[3] = (float)st.theta_dot;
    float* ro = a.obs + ((size_t)row * a.T + t) * a.obs_dim;
    ro[0] = o[0], ro[1] = o[1], ro[2] = o[2], ro[3] = o[3];
    a.action[(size_t)row * a.T + t] = act;
    a.reward[(size_t)row * a.T + t] = 1.0f;
    a.length[row] = t + 1;
    /* play_game's loop condition (self_play.py:62): not terminal and counter < limit_of_game_play */
    a.alive[g] = !terminated && t + 1 < a.T;
    a.stream_seed[g] = arena_stream_seed(a.seed[g], a.episode[g], t + 1);
}

__global__ void __launch_bounds__(256) k_arena_step(const ArenaArrays a, const int* visits,
                                                    const float* root_value, const double* root_prior,
                                                    double temperature, int from_prior, float* obs_io) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.G || !a.alive[g]) return;
    arena_step_slot(a, g, visits, root_value, root_prior, temperature, from_prior, obs_io);
}

/* Free slots take the round's next unplayed games, in slot order (deterministic: a block-wide
 * scan, no atomics); *alive_count = slots with a running game afterwards.  One block of
 * kRefillThreads threads; thread t owns the consecutive slots [t * per, (t + 1) * per).  The move's
 * status entry (see status_ring) is the last thing written.  consumed (may be null): draws each
 * slot's search took (mzs_results), whose maximum the host sizes the next prefix from. */
constexpr int kRefillThreads = 1024;
constexpr int kRefillWaves = kRefillThreads / 64;

__device__ __forceinline__ void arena_refill_block(const ArenaArrays& a, float* obs_io, int* alive_count,
                                                   const long long* consumed, int* wsum /* [2 * kRefillWaves] */) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int per = (a.G + kRefillThreads - 1) / kRefillThreads;
    const int g0 = tid * per < a.G ? tid * per : a.G, g1 = g0 + per < a.G ? g0 + per : a.G;
    const int next = a.ctl[0], end = a.ctl[1];
    int dead = 0, used = 0;
    for (int g = g0; g < g1; ++g) {
        dead += !a.alive[g];
        if (consumed) used = consumed[g] > used ? (int)consumed[g] : used;
    }
    /* inclusive scan of `dead` inside the wave, then over the sixteen wave totals */
    int inc = dead;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(inc, off, 64);
        if (lane >= off) inc += v;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    int base = 0, total_dead = 0;
#pragma unroll
    for (int i = 0; i < kRefillWaves; ++i) {
        const int v = wsum[i];
        base += i < w ? v : 0;
        total_dead += v;
    }
    int k = next + base + inc - dead; /* virtual row of this thread's first free slot */
    int oldest = 0x7fffffff;          /* oldest game still (or newly) running among this thread's slots */
    for (int g = g0; g < g1; ++g) {
        if (!a.alive[g]) {
            if (k < end) arena_start_game(a, g, k, obs_io);
            ++k;
        }
        if (a.alive[g]) oldest = a.vrow[g] < oldest ? a.vrow[g] : oldest;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int v = __shfl_xor(oldest, off, 64), u = __shfl_xor(used, off, 64);
        oldest = v < oldest ? v : oldest;
        used = u > used ? u : used;
    }
    __syncthreads(); /* every read of wsum's first half is done */
    if (lane == 0) wsum[w] = oldest, wsum[kRefillWaves + w] = used;
    __syncthreads();
    if (tid == 0) {
        for (int i = 1; i < kRefillWaves; ++i) {
            oldest = wsum[i] < oldest ? wsum[i] : oldest;
            used = wsum[kRefillWaves + i] > used ? wsum[kRefillWaves + i] : used;
        }
        const int started = next + total_dead <= end ? total_dead : end - next;
        const int alive = a.G - total_dead + started;
        const int done = oldest < next + started ? oldest : next + started;
        const int move = a.ctl[3];
        a.ctl[0] = next + started;
        a.ctl[2] = done;
        a.ctl[3] = move + 1;
        *alive_count = alive;
        if (a.status_ring) {
            /* no fences: the reader's contract is to look only after an event behind this kernel
             * has completed, when all of it is visible; the move index in both halves (two 16-byte
             * stores) lets mzs_arena_status reject an entry it was asked for too early */
            int4* e = reinterpret_cast<int4*>(a.status_ring + (size_t)(move % kArenaRing) * 8);
            e[0] = make_int4(move, next + started, end, done);
            e[1] = make_int4(alive, used, move, 0);
        }
    }
}

__global__ void __launch_bounds__(kRefillThreads) k_arena_refill(const ArenaArrays a, float* obs_io,
                                                                 int* alive_count, const long long* consumed) {
    __shared__ int wsum[2 * kRefillWaves];
    arena_refill_block(a, obs_io, alive_count, consumed, wsum);
}

struct BatchParams {
    ArenaArrays a;
    int B, K, td;
    const int* row_pos;        /* [B][2] (trajectory row, position)                          */
    const double* disc_pow;    /* [td+1] discount ** i, evaluated by the host in Python      */
    float* obs;                /* [B][obs]                                                   */
    float* actions;            /* [K][B][A]                                                  */
    float* tv;                 /* [K][B]                                                     */
    float* tr;                 /* [K][B]                                                     */
    float* tp;                 /* [K][B][A]                                                  */
};

/* n-step value target of Game.make_target (game.py:242-250) with the reference's dtype behaviour
 * under NumPy 2: when a bootstrap value exists the sum runs in fp32 (np.float32 root value times
 * a Python float stays fp32, and so does every += of a Python float), otherwise in fp64. */
__device__ __forceinline__ double target_value(const ArenaArrays& a, int row, int len, int ci, int td,
                                              const double* disc_pow) {
    const int bi = ci + td;
    const float* rw = a.reward + (size_t)row * a.T;
    const int hi = bi < len ? bi : len;
    if (bi < len) {
        float v = a.root_value[(size_t)row * a.T + bi] * (float)disc_pow[td];
        for (int i = 0; ci + i < hi; ++i) v = v + (float)((double)rw[ci + i] * disc_pow[i]);
        return (double)v;
    }
    double v = 0.0;
    for (int i = 0; ci + i < hi; ++i) v = v + (double)rw[ci + i] * disc_pow[i];
    return v; /* make_target's caller rounds it to fp32 (reshape_batch), make_priority keeps fp64 */
}

/* Game.make_priority (game.py:262-284) for rows [row0, row0 + n): |root value - n-step target| in
 * fp64 per position (0 past the end of the game), and the game's maximum is left to the host
 * (np.max over a few dozen doubles).  One thread per (row, position). */
__global__ void __launch_bounds__(256) k_arena_priority(const ArenaArrays a, int row0, int n, int td,
                                                        const double* disc_pow, int out_stride, double* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * out_stride) return;
    const int r = i / out_stride, pos = i - r * out_stride, row = row0 + r;
    const int len = a.length[row];
    double v = 0.0;
    if (pos < len && pos < a.T)
        v = fabs((double)a.root_value[(size_t)row * a.T + pos] - target_value(a, row, len, pos, td, disc_pow));
    out[i] = v;
}

/* the same, packed: game r's len(r) priorities at out[offsets[r] ...] (offsets = exclusive prefix sum
 * of the rows' lengths, made by the caller) — what the host's ReplayStore keeps, without the zeros */
__global__ void __launch_bounds__(256) k_arena_priority_packed(const ArenaArrays a, int row0, int n, int td,
                                                               const double* disc_pow, int longest,
                                                               const long long* offsets, double* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * longest) return;
    const int r = i / longest, pos = i - r * longest, row = row0 + r;
    const int len = a.length[row];
    if (pos < len && pos < a.T)
        out[offsets[r] + pos] =
            fabs((double)a.root_value[(size_t)row * a.T + pos] - target_value(a, row, len, pos, td, disc_pow));
}

/* ReplayBuffer.sample_batch's per-sample work (replay_buffer.py:147-151): make_image,
 * fill_gap_empty_action, make_target, written in the layout reshape_batch produces
 * (muzero_model.py:440-472).  One thread per (sample, unroll step). */
__global__ void __launch_bounds__(256) k_arena_make_batch(const BatchParams p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.B * p.K) return;
    const ArenaArrays& a = p.a;
    const int b = i % p.B, k = i / p.B;
    int row = p.row_pos[2 * b], pos = p.row_pos[2 * b + 1];
    /* the host validates the table (muzero_amd/selfplay.py); clamp anyway so that a bad entry reads
     * a wrong row instead of faulting the GPU */
    row = row < 0 ? 0 : (row >= a.n_rows ? a.n_rows - 1 : row);
    pos = pos < 0 ? 0 : (pos >= a.T ? a.T - 1 : pos);
    const int len = a.length[row], A = a.A, ci = pos + k;
    if (k == 0) {
        const float* o = a.obs + ((size_t)row * a.T + pos) * a.obs_dim;
        for (int j = 0; j < a.obs_dim; ++j) p.obs[(size_t)b * a.obs_dim + j] = o[j];
    }
    /* action_history[pos:][k], zeros past the end of the game */
    const int act = ci < len ? a.action[(size_t)row * a.T + ci] : -1;
    for (int j = 0; j < A; ++j) p.actions[((size_t)k * p.B + b) * A + j] = j == act ? 1.0f : 0.0f;
    const float last_reward = (ci > 0 && ci <= len) ? a.reward[(size_t)row * a.T + ci - 1] : 0.0f;
    p.tr[(size_t)k * p.B + b] = last_reward;
    if (ci < len) {
        p.tv[(size_t)k * p.B + b] = (float)target_value(a, row, len, ci, p.td, p.disc_pow);
        for (int j = 0; j < A; ++j)
            p.tp[((size_t)k * p.B + b) * A + j] = a.policy[((size_t)row * a.T + ci) * A + j];
    } else { /* absorbing state */
        p.tv[(size_t)k * p.B + b] = 0.0f;
        for (int j = 0; j < A; ++j) p.tp[((size_t)k * p.B + b) * A + j] = 0.0f;
    }
}

}  // namespace mzs
#endif
