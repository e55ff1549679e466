/*
 * mz_oracle.h — CPU oracle for the MuZero self-play search path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in muzero_amd/ may include, link or call
 * this; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * It restates, in plain C, the algorithm of the reference
 * (/root/reference, DHDev0/Muzero):
 *   search : monte_carlo_tree_search.py:113-265
 *   model  : neural_network_mlp_model.py:7-143, muzero_model.py:490-517,:571-587,
 *            :828-905
 * under the arithmetic contract of include/mzs_numeric.h.
 *
 * Pinning (tests/test_oracle_golden.py, fixtures made by
 * tests/golden/make_fixtures.py importing the reference in the build container):
 *   L1  search with the reference's own model outputs injected: visit counts,
 *       expanded leaves, root value, draws consumed — bit-exact on every fixture;
 *   L2  model functions within 1e-5 relative of the reference wrappers;
 *   L3  end-to-end search: bit-exact visit counts on every fixture search whose
 *       recorded minimum decision margin is >= 1e-4 (near-ties below the model
 *       tolerance may legitimately flip), root values within 1e-5 relative.
 */
#ifndef MZ_ORACLE_H
#define MZ_ORACLE_H

#include <stddef.h>

#include "../include/mzs_layout.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mzo_search_cfg {
    mzs_dims dims;
    int32_t num_simulations;      /* S                                            */
    int32_t train;                /* add Dirichlet noise at the root (mcts.py:134) */
    int32_t n_threads;            /* games are sharded over OpenMP threads         */
    int32_t _pad;
    double discount;              /* gamma, rounded to fp32 where the reference does */
    double exploration_fraction;  /* root_exploration_fraction                     */
    const double* pbc_table;      /* [S+1]: sqrt(n)*(log((n+base+1)/base)+init), host numpy */
} mzo_search_cfg;

/* L1 mode: model outputs supplied per simulation instead of computed. */
typedef struct mzo_table_model {
    const float* root_policy; /* [G][A]    softmax prior of the root               */
    const float* sim_reward;  /* [G][S]    reward of the leaf expanded at sim s    */
    const float* sim_value;   /* [G][S]    value  of that leaf                     */
    const float* sim_prior;   /* [G][S][A] prior of its children                   */
} mzo_table_model;

/* Any pointer may be NULL. N = 1 + A*(S+1). */
typedef struct mzo_search_out {
    int32_t* visits;      /* [G][A] root-child visit counts (mcts.py:219)         */
    float* root_value;    /* [G]    root.value() (mcts.py:19-20)                  */
    double* root_prior;   /* [G][A] root-child priors after noise (mcts.py:144)   */
    int64_t* consumed;    /* [G]    tie-break draws consumed (mcts.py:163)        */
    int64_t* path_len;    /* [G]    sum over sims of selections made (mcts.py:173-176) */
    int32_t* sim_leaf;    /* [G][S] node expanded by simulation s                 */
    int32_t* node_visit;  /* [G][N] */
    float* node_vsum;     /* [G][N] */
    float* node_reward;   /* [G][N] */
    float* node_prior;    /* [G][N] */
    int32_t* child_base;  /* [G][N] first child index, 0 = not expanded           */
    float* minmax;        /* [G][2] MinMaxStats minimum, maximum (mcts.py:23-35)  */
    double* min_margin;   /* [G]    smallest gap between best and runner-up score */
    float* hidden;        /* [G][S+1][S_dim] hidden state of each expansion       */
} mzo_search_out;

void mzo_representation(const mzs_dims* d, const float* blob, const float* obs, int32_t B,
                        float* hidden);
void mzo_dynamics(const mzs_dims* d, const float* blob, const float* hidden,
                  const int32_t* action, int32_t B, float* reward, float* next_hidden);
void mzo_prediction(const mzs_dims* d, const float* blob, const float* hidden, int32_t B,
                    float* policy, float* value);

/* 'vision_model' (oracle/mz_vision_oracle.c; pinned by tests/golden/model_L2_vis*.npz): obs
 * [B][3][98][98], hidden states [B][3][7][7]; the three functions above dispatch here.  stages (may
 * be NULL): Down_sample's output of the first observation; reward_logits (may be NULL) [B][S_dim]. */
void mzo_vision_representation(const mzs_dims* d, const float* blob, const float* obs, int32_t B, float* hidden,
                               float* stages);
void mzo_vision_dynamics(const mzs_dims* d, const float* blob, const float* hidden, const int32_t* action, int32_t B,
                         float* reward, float* next_hidden, float* reward_logits);
void mzo_vision_prediction(const mzs_dims* d, const float* blob, const float* hidden, int32_t B, float* policy,
                           float* value);
void mzo_softmax(const float* q, int32_t n, float* p);

/* building blocks exposed for unit pinning */
void mzo_scale_to_bound(const float* x, int32_t n, float* y);
float mzo_support_to_scalar(const float* logits, int32_t n);
float mzo_expf(float x);
/* n groups of four bf16 pairs -> what one v_mfma_f32_4x4x4_16B_bf16 with a zero accumulator returns
 * (include/mzs_numeric.h mzs_dot4_bf16); round-to-nearest-even fp32 -> bf16 -> fp32 */
void mzo_dot4_bf16(const uint16_t* a, const uint16_t* b, int32_t n, float* out);
void mzo_round_bf16(const float* x, int32_t n, float* out);
float mzo_elu(float x);

/* returns 0, or -1 on a bad argument */
int mzo_search(const mzo_search_cfg* cfg, const float* blob, const mzo_table_model* tbl,
               const float* obs, const double* dirichlet, const double* noise,
               int64_t noise_stride, int32_t G, const mzo_search_out* out);

/* ---- training step (oracle/mz_train_oracle.c; pinned by tests/golden/train_*.npz) ---- */
typedef struct mzo_train_batch {
    int32_t B, K;               /* samples in this shard, unroll length (k_hypothetical_steps)  */
    const float* obs;           /* [B][obs_dim]                                                 */
    const float* actions;       /* [K][B][A]  one-hot, or zeros for absorbing positions         */
    const float* target_value;  /* [K][B]                                                       */
    const float* target_reward; /* [K][B]                                                       */
    const float* target_policy; /* [K][B][A]                                                    */
    const float* is_ratio;      /* [B] importance-sampling ratio, NULL = none (uniform sampling) */
    float inv_batch;            /* 1 / global batch size                                        */
} mzo_train_batch;

/* Muzero.train up to (not including) the optimizer step: grad [total blob floats] (blob order),
 * loss_terms [3K-1] batch means of the raw terms in store_loss order, *total_loss = loss_nn,
 * priority [K][B] = |h^-1(value_k) - target_value_k| (muzero_model.py:710-713). */
/* the criteria of the value and reward terms (loss_type, muzero_model.py:350-403: 'general' = CE / CE,
 * 'general_kkc' = KL / KL, 'game' = MSE / ZERO, 'game_mmc' = MSE / MSE; the policy term is always CE);
 * process-wide state of the oracle, default CE / CE */
#define MZO_LOSS_CE 0
#define MZO_LOSS_KL 1
#define MZO_LOSS_MSE 2
#define MZO_LOSS_ZERO 3
void mzo_train_set_loss(int value_mode, int reward_mode);
int mzo_train_grad(const mzs_dims* d, const float* blob, const mzo_train_batch* bt, float* grad,
                   float* loss_terms, double* total_loss, float* priority);
void mzo_set_momentum(double momentum); /* Adam beta1 / SGD momentum of the following steps (default 0.9) */
void mzo_adam_step(float* w, float* m, float* v, const float* g, size_t n, int step, double lr);
void mzo_sgd_step(float* w, float* buf, const float* g, size_t n, int step, double lr);

/* ---- self-play bookkeeping (oracle/mz_selfplay_oracle.c; pinned by tests/test_selfplay_oracle.py).
 * Trajectory arrays are [n_rows][T]...; slots take rows as mzs_arena_begin / mzs_arena_step do. */
void mzo_store_policy(int32_t A, const int32_t* visits, const double* prior, float* policy);
int32_t mzo_policy_action(int32_t A, const double* policy, double temperature, double u);
double mzo_cartpole_step(double* state4, int32_t action, int32_t* terminated); /* returns the reward */
typedef struct mzo_arena { /* arrays owned by the caller (oracle/mz_oracle.py) */
    /* next_row / row_end / done_upto count VIRTUAL rows (never wrap; trajectory row = v % n_rows):
     * [next_row, row_end) are the queued games no slot has started yet, every game below done_upto
     * has finished */
    int32_t G, n_rows, T, A, episode, next_row, row_end, done_upto;
    double* state;        /* [G][4] */
    int32_t* alive;       /* [G] */
    int32_t* row;         /* [G] */
    int64_t* stream_seed; /* [G] */
    float* obs;           /* [G][4] */
    int64_t* row_seed;    /* [n_rows] */
    int32_t* length;      /* [n_rows] */
    float* traj_obs;      /* [n_rows][T][4] */
    int32_t* traj_action; /* [n_rows][T] */
    float* traj_reward;   /* [n_rows][T] */
    float* traj_root;     /* [n_rows][T] */
    float* traj_policy;   /* [n_rows][T][A] */
    int32_t* row_episode; /* [n_rows] random-number key of the row's game (the round it was queued for) */
    double* row_temp;     /* [n_rows] action-selection temperature of the row's game; NaN: the step's argument */
    int32_t* vrow;        /* [G] virtual row of the slot's game */
} mzo_arena;
void mzo_arena_begin(mzo_arena* ar, int32_t row0, int32_t n_games, const int64_t* seeds, int32_t episode);
/* mzs_arena_enqueue: n_games more games behind the queue's end */
void mzo_arena_enqueue(mzo_arena* ar, int32_t n_games, const int64_t* seeds, int32_t episode, double temperature);
int32_t mzo_arena_step(mzo_arena* ar, const int32_t* visits, const float* root_value, const double* root_prior,
                       double temperature, int32_t from_prior);
void mzo_make_priority(int32_t n_rows, int32_t T, const int32_t* length, const float* traj_reward,
                       const float* traj_root, int32_t td, const double* disc_pow, double* out);
void mzo_make_batch(int32_t T, int32_t A, int32_t obs_dim, const int32_t* length, const float* traj_obs,
                    const int32_t* traj_action, const float* traj_reward, const float* traj_root,
                    const float* traj_policy, int32_t B, const int32_t* row_pos, int32_t K, int32_t td,
                    const double* disc_pow, float* obs, float* actions, float* tv, float* tr, float* tp);

#ifdef __cplusplus
}
#endif
#endif
