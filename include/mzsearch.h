/*
 * mzsearch.h — C-ABI of libmzsearch.so, the MI355X (gfx950) implementation of the
 * MuZero self-play search hot path of DHDev0/Muzero.
 *
 * The reference has no FFI for this path: the seam is three Python call surfaces
 * (SURVEY.md §8b).  Each entry point below names the reference code it replaces;
 * muzero_amd/search.py and muzero_amd/model.py bind them with ctypes and present
 * the reference's own classes on top.
 *
 * Conventions: every function returns 0 on success or a negative MZS_E_* code and
 * never throws; mzs_last_error() describes the last failure on the calling thread.
 * "_dev" pointers are device memory on the handle's GPU, owned by the caller and
 * borrowed for the duration of the call.  `stream` is a hipStream_t (pass
 * torch.cuda.current_stream().cuda_stream); calls only enqueue work, the caller
 * synchronises.  One handle per (GPU, stream); a handle is not thread-safe.
 * Arithmetic follows include/mzs_numeric.h; weight blob and node numbering follow
 * include/mzs_layout.h.
 */
#ifndef MZSEARCH_H
#define MZSEARCH_H

#include <stddef.h>
#include <stdint.h>

#include "mzs_layout.h"

#ifdef __cplusplus
extern "C" {
#endif

#define MZS_OK 0
#define MZS_E_ARG (-1)      /* bad argument / unsupported size */
#define MZS_E_HIP (-2)      /* a HIP runtime call failed       */
#define MZS_E_STATE (-3)    /* call out of order (e.g. search before set_weights) */

/* search engines */
#define MZS_ENGINE_FUSED 0    /* one launch per search: wave per game, tree in LDS; picks the
                                 register-resident specialisation when the shape has one   */
#define MZS_ENGINE_FUSED_GENERIC 2 /* fused, always the runtime-shape kernel (A/B testing)  */

typedef struct mzs_handle mzs_handle;

typedef struct mzs_config {
    mzs_dims dims;                  /* Muzero(...) ctor arguments, muzero_model.py:16-35        */
    int32_t max_games;              /* G: concurrent games (trees) this handle can hold         */
    int32_t max_simulations;        /* S: upper bound of num_simulations (mcts.py:224)          */
    int32_t device;                 /* HIP device ordinal                                       */
    int32_t engine;                 /* MZS_ENGINE_*                                             */
    double discount;                /* Monte_carlo_tree_search.discount (mcts.py:81)            */
    double root_exploration_fraction; /* mcts.py:83                                             */
    const double* pbc_table;        /* host, [max_simulations+1]: sqrt(n)*(log((n+base+1)/base)+init)
                                       for parent visit count n, evaluated by the caller in
                                       float64 as mcts.py:156-157 does (muzero_amd.rng.pbc_table) */
} mzs_config;

/* Monte_carlo_tree_search.__init__/reset (monte_carlo_tree_search.py:38-86) + Muzero model
 * shapes: allocates the SoA tree for max_games games and the weight store. */
int mzs_create(const mzs_config* cfg, mzs_handle** out);
void mzs_destroy(mzs_handle* h);

/* Number of floats mzs_set_weights expects (include/mzs_layout.h). */
size_t mzs_weight_count(const mzs_dims* dims);

/* Replaces holding the three nn.Modules (muzero_model.py:257-273): uploads the packed fp32
 * weights.  `blob` may be a host or a device pointer. */
int mzs_set_weights(mzs_handle* h, const float* blob, size_t n_floats);

/* Monte_carlo_tree_search.run (monte_carlo_tree_search.py:224-265) for n_games games at once.
 *   obs_dev       [n_games][obs_dim] f32   observation of each game's root
 *   dirichlet_dev [n_games][A] f64         np.random.dirichlet([alpha]*A) per game (mcts.py:140);
 *                                          may be NULL when train == 0 or num_simulations == 0
 *   noise_dev     [n_games][noise_stride] f64  the np.random.uniform(1e-7, 2e-7) tie-break draws
 *                                          in consumption order (mcts.py:163); stride >=
 *                                          mzs_max_draws(A, num_simulations)
 * Results stay in the handle (mzs_results / mzs_tree_array). */
int mzs_search(mzs_handle* h, int32_t n_games, int32_t num_simulations, int32_t train,
               const float* obs_dev, const double* dirichlet_dev, const double* noise_dev,
               int64_t noise_stride, void* stream);

/* Name of the kernel mzs_search launches on this handle for a search of num_simulations
 * simulations (the register-resident / weight-stream / generic variants, include the template
 * arguments), NUL-terminated into name_out: what a profile of the search shows, for benchmark
 * records.  No counterpart in the reference. */
int mzs_search_kernel_name(mzs_handle* h, int32_t num_simulations, char* name_out, size_t name_size);

/* The per-game random streams mzs_search consumes, generated on the device: for game g exactly
 * what numpy.random.RandomState(seeds[g]) returns for dirichlet([alpha] * A) (only when train and
 * num_simulations > 0) followed by uniform(lo, hi, size = mzs_max_draws(A, S)) — i.e. what the
 * reference draws from NumPy's global stream after np.random.seed(seed)
 * (monte_carlo_tree_search.py:140, :163 with lo = 1e-7, hi = 2e-7).  Uniform draws are bit-exact;
 * the Dirichlet goes through the device's fp64 log / pow (within an ulp or two of NumPy's).
 * seeds_dev [n_games] i64 in [0, 2^32); 0 < alpha <= 1. */
int mzs_fill_streams(mzs_handle* h, int32_t n_games, int32_t num_simulations, int32_t train,
                     const int64_t* seeds_dev, double alpha, double lo, double hi,
                     double* dirichlet_dev, double* noise_dev, int64_t noise_stride, void* stream);

/* mzs_fill_streams + mzs_search as the self-play loop needs them once per move, without writing
 * the whole worst-case stream first: a search of S simulations may read up to mzs_max_draws(A, S)
 * tie-break draws (one chain of S nodes) but typically reads a fifth of that.  This call
 *   1. writes the Dirichlet rows and the first prefix_draws uniform draws of every game
 *      (prefix_draws <= 0: 9 * A * S, about twice the typical consumption),
 *   2. searches; a game whose search wants a draw past the prefix is flagged (mzs_results status),
 *   3. for the flagged games only: writes the whole stream and searches again.
 * Steps 1-2 are the common case; the two launches of step 3 leave at once when no game is flagged.
 * Results (mzs_results, tree arrays) equal those of mzs_fill_streams + mzs_search bit for bit;
 * dirichlet_dev [n_games][A] and noise_dev [n_games][noise_stride >= mzs_max_draws] are the
 * caller's scratch.  Reference: the same np.random draws, monte_carlo_tree_search.py:140, :163. */
int mzs_search_streams(mzs_handle* h, int32_t n_games, int32_t num_simulations, int32_t train,
                       const float* obs_dev, const int64_t* seeds_dev, double alpha, double lo, double hi,
                       int32_t prefix_draws, double* dirichlet_dev, double* noise_dev, int64_t noise_stride,
                       void* stream);

/* Same search with the model outputs supplied instead of computed (L1 parity mode): isolates
 * ucb_score/select_child/expand/backup (mcts.py:155-209) from the network arithmetic.
 *   root_policy_dev [n_games][A], sim_reward_dev/sim_value_dev [n_games][S],
 *   sim_prior_dev [n_games][S][A]: what prediction/dynamics returned at each simulation. */
int mzs_search_injected(mzs_handle* h, int32_t n_games, int32_t num_simulations, int32_t train,
                        const float* root_policy_dev, const float* sim_reward_dev,
                        const float* sim_value_dev, const float* sim_prior_dev,
                        const double* dirichlet_dev, const double* noise_dev,
                        int64_t noise_stride, void* stream);

/* Self-test of one assumption of the numeric contract on the device in use: the search kernels
 * take 1 / visit_count in f64 as hardware estimate + two Newton steps, which must equal the
 * correctly rounded quotient (what NumPy / the oracle compute) for every count a tree can hold.
 * Evaluates d = 1 .. n_max on the device and counts the mismatches against the host's 1.0 / d.
 * No counterpart in the reference (pb_c / (n + 1) is Python float division: mcts.py:158). */
int mzs_selftest_reciprocals(int32_t device, int32_t n_max, int32_t* n_bad);

/* A second destination for the two arrays a caller on the host reads after every search: from
 * now on each search also stores its visit counts ([max_games][A] int32) and root values
 * ([max_games] float) there.  Meant for device-visible host memory (hipHostMalloc / a pinned
 * torch tensor): the results are on the host when the search kernel completes, with no copy
 * launch behind it (on MI355X a 48 KB device->host copy after a 0.22 ms search costs ~12 us of
 * stream bubbles).  NULL, NULL switches it off.  The handle's own arrays are written as before.
 * No counterpart in the reference (its MCTS returns Python lists: mcts.py:215-222). */
int mzs_set_result_sink(mzs_handle* h, int32_t* visits, float* root_value);

/* generate_action_and_policy_base_on_node_count (mcts.py:215-222) + what
 * Game.store_search_statistics reads (game.py:154-168).  Any pointer may be NULL.
 *   visits_dev     [n_games][A] i32  root-child visit counts
 *   root_value_dev [n_games] f32     root.value()
 *   root_prior_dev [n_games][A] f64  root-child priors after exploration noise
 *   consumed_dev   [n_games] i64     tie-break draws consumed (so the caller can advance its RNG)
 *   path_len_dev   [n_games] i64     total selections made (mean path length = /S) */
int mzs_results(mzs_handle* h, int32_t n_games, int32_t* visits_dev, float* root_value_dev,
                double* root_prior_dev, int64_t* consumed_dev, int64_t* path_len_dev,
                void* stream);

/* Device pointers into the handle's SoA tree (row stride = mzs_tree_stride(h) elements):
 * the arrays that replace the Node objects (mcts.py:6-20). */
#define MZS_TREE_VISIT 0       /* i32 [G][stride]  Node.visit_count                 */
#define MZS_TREE_VALUE_SUM 1   /* f32 [G][stride]  Node.value_sum                   */
#define MZS_TREE_PRIOR 2       /* f32 [G][stride]  Node.prior                       */
#define MZS_TREE_REWARD 3      /* f32 [G][stride]  Node.reward                      */
#define MZS_TREE_CHILD_BASE 4  /* i32 [G][stride]  first child index, 0 = leaf      */
#define MZS_TREE_SIM_LEAF 5    /* i32 [G][max_simulations] node expanded at sim s   */
#define MZS_TREE_MINMAX 6      /* f32 [G][2]       MinMaxStats minimum, maximum     */
#define MZS_TREE_HIDDEN 7      /* f32 [G][max_simulations+1][mzs_hidden_stride(h)]  */
#define MZS_TREE_STATUS 8      /* i32 [G]          0 ok, 1 noise stream exhausted   */
/* the arrays mzs_results copies from (zero-copy access for callers that read in place) */
#define MZS_TREE_ROOT_VISITS 9 /* i32 [G][A]       root-child visit counts          */
#define MZS_TREE_ROOT_VALUE 10 /* f32 [G]          root.value(); starts right behind the visit
                                  counts of max_games games (one copy reads both)   */
#define MZS_TREE_ROOT_PRIOR 11 /* f64 [G][A]       root-child priors after noise    */
#define MZS_TREE_CONSUMED 12   /* i64 [G]          tie-break draws consumed         */
#define MZS_TREE_PATH_LEN 13   /* i64 [G]          selections made                  */
void* mzs_tree_array(mzs_handle* h, int32_t which);
int32_t mzs_tree_stride(const mzs_handle* h);
int32_t mzs_hidden_stride(const mzs_handle* h);

/* Muzero.representation_function_inference + prediction_function_inference
 * (muzero_model.py:828-845, :883-905) on a batch: hidden [B][S_dim], policy [B][A] (softmax),
 * value [B] (after inverse_transform_with_support).  policy/value may be NULL. */
int mzs_initial_inference(mzs_handle* h, int32_t batch, const float* obs_dev, float* hidden_dev,
                          float* policy_dev, float* value_dev, void* stream);

/* Muzero.dynamics_function_inference + prediction_function_inference on the new state
 * (muzero_model.py:848-905): reward [B], next_hidden [B][S_dim], policy [B][A], value [B].
 * policy/value may be NULL (dynamics only). */
int mzs_recurrent_inference(mzs_handle* h, int32_t batch, const float* hidden_dev,
                            const int32_t* action_dev, float* reward_dev, float* next_hidden_dev,
                            float* policy_dev, float* value_dev, void* stream);

/* Muzero.prediction_function_inference alone (muzero_model.py:883-905). */
int mzs_prediction_inference(mzs_handle* h, int32_t batch, const float* hidden_dev,
                             float* policy_dev, float* value_dev, void* stream);

/* ---- training step (the reference's Muzero.train, muzero_model.py:791-811) ----------------
 * Optimizer state and the step counter live in the handle; weights are the ones of
 * mzs_set_weights and every mzs_train_apply refreshes what the search / inference kernels read. */
#define MZS_OPT_ADAM 0 /* torch.optim.Adam(lr, betas 0.9/0.999, eps 1e-8), muzero_model.py:419 */
#define MZS_OPT_SGD 1  /* torch.optim.SGD(lr, momentum 0.9), muzero_model.py:421               */

/* Allocates the per-sample tapes for batches of up to max_batch samples (this rank's shard) and
 * k_hypothetical_steps = unroll_steps (muzero_model.py:623), and zeroed optimizer state. */
int mzs_train_create(mzs_handle* h, int32_t max_batch, int32_t unroll_steps);

/* reshape_batch -> compute_forward -> evaluate_loss -> backward (muzero_model.py:440-472,
 * :609-647, :650-734, :757-759) on `batch` samples, up to the optimizer step:
 *   obs_dev [B][obs_dim]; actions_dev [K][B][A] one-hot (zeros = absorbing, replay_buffer.py:120);
 *   target_value_dev / target_reward_dev [K][B]; target_policy_dev [K][B][A];
 *   is_ratio_dev [B] importance-sampling ratio or NULL (uniform sampling, replay_buffer.py:165);
 *   inv_global_batch = 1 / (batch size over all data-parallel ranks)
 * outputs: grad_dev [mzs_weight_count] in blob order (this shard's share: ranks SUM them),
 *   loss_dev [3K] = the 3K-1 raw loss terms in store_loss order (shard's share of the batch mean)
 *   then loss_nn; priority_dev [K][B] = |h^-1(value_k) - target_value_k| (muzero_model.py:710). */
int mzs_train_grad(mzs_handle* h, int32_t batch, const float* obs_dev, const float* actions_dev,
                   const float* target_value_dev, const float* target_reward_dev,
                   const float* target_policy_dev, const float* is_ratio_dev, float inv_global_batch,
                   float* grad_dev, float* loss_dev, float* priority_dev, void* stream);

/* mzs_train_grad for callers that cannot allocate: code under HIP graph capture, tight serving /
 * training loops, harnesses that forbid hipMalloc.  NO counterpart in the reference (its training
 * step allocates autograd tensors freely).  Same arguments (plus unroll_steps, which mzs_train_grad
 * takes from mzs_train_create), same kernels, same results bit for bit — but it needs no
 * mzs_train_create and calls no hipMalloc / hipFree / allocating routine: tapes, per-sample loss
 * terms and gradient partial sums live in a file-scope static device array of
 * MZS_STATIC_SCRATCH_BYTES (16 MiB) that every GPU the library runs on holds until the process
 * ends.  Sizes beyond it (e.g. the CartPole MLP at unroll 5 fits batch <= 777) return MZS_E_ARG.
 * Every call on a GPU shares that GPU's array: calls that overlap on one GPU (two streams, two
 * handles, two threads) race — serialise them.  The handle still comes from mzs_create +
 * mzs_set_weights (which allocate, once, outside the loop). */
#define MZS_STATIC_SCRATCH_BYTES (16u << 20)
int mzs_train_grad_static(mzs_handle* h, int32_t batch, int32_t unroll_steps, const float* obs_dev,
                          const float* actions_dev, const float* target_value_dev,
                          const float* target_reward_dev, const float* target_policy_dev,
                          const float* is_ratio_dev, float inv_global_batch, float* grad_dev, float* loss_dev,
                          float* priority_dev, void* stream);

/* 'vision_model' (mzs_dims.model_kind = MZS_MODEL_VISION; neural_network_vision_model.py in training mode):
 * the same three calls — mzs_train_create, mzs_train_grad (obs_dev [B][3][98][98]; the action plane of step k is
 * (argmax(actions[k][b]) + 1) / A, muzero_model.py:505-516), mzs_train_apply — with two additions:
 *   BatchNorm2d runs on batch statistics (of this rank's shard, as torch.nn.DataParallel's replicas do,
 *   muzero_model.py:333-340) and every application moves the running statistics (momentum 0.1); the K-th
 *   dynamics / prediction calls, whose outputs evaluate_loss drops, still run for that.  The blob slots of the
 *   running statistics come back in grad_dev holding (new - old) x batch x inv_global_batch: ranks SUM them
 *   with the gradient and mzs_train_apply ADDS the sum to the blob instead of optimising it.
 *   Dropout2d(p = 0.5) zeroes whole (sample, channel) maps, three times per residual block.  torch draws the
 *   masks from its generator; here they come from a counter-based generator keyed by (seed, call number), or
 *   — for parity tests against recorded reference runs — from the caller: */

/* bytes of masks one mzs_train_grad on `batch` samples consumes: one per (Dropout2d call, sample, channel),
 * calls in the reference's forward order (representation, prediction, K x (dynamics, prediction)) */
int mzs_train_dropout_count(mzs_handle* h, int32_t batch, int64_t* n_out);

/* masks_dev != NULL: the NEXT mzs_train_grad reads its masks (1 = kept, scaled by 2; 0 = dropped) from
 * masks_dev[0 .. n_masks), which must equal mzs_train_dropout_count and stay valid until that call's kernels
 * have run; later calls draw their own again.  masks_dev == NULL: restart the generator at `seed`. */
int mzs_train_set_dropout(mzs_handle* h, const unsigned char* masks_dev, int64_t n_masks, uint64_t seed);

/* The criteria of the value and reward terms of evaluate_loss, i.e. the reference's loss_type
 * (init_criterion_and_optimizer, muzero_model.py:350-403; Loss_function, neural_network_mlp_model.py:
 * 236-262, always on softmax(logits) + 1e-9): 'general' = CE / CE (the default), 'general_kkc' =
 * KL / KL (log(target + 1e-9): NaN for negative targets, as in the reference), 'game' = MSE / ZERO,
 * 'game_mmc' = MSE / MSE.  The policy term is the cross-entropy in all of them; with a KL value criterion
 * ('general_kkc', whose policy criterion carries label = ["zero_clamp_transform"], muzero_model.py:373-375)
 * it is taken against target + 1e-9 like the reference's.  Applies to every later mzs_train_grad /
 * mzs_train_grad_static on the handle. */
#define MZS_LOSS_CE 0
#define MZS_LOSS_KL 1
#define MZS_LOSS_MSE 2
#define MZS_LOSS_ZERO 3
int mzs_train_set_loss(mzs_handle* h, int32_t value_criterion, int32_t reward_criterion);

/* Adam's beta1 / SGD's momentum for the following mzs_train_apply calls (default 0.9, what
 * muzero_model.py:418-421 constructs).  OneCycleLR (muzero_model.py:433) cycles it per step together
 * with the rate: the caller evaluates both on the host (muzero_amd/train.py) and passes them in. */
int mzs_train_set_momentum(mzs_handle* h, double momentum);

/* optimizer.step() (muzero_model.py:764) with the (all-reduced) gradient; lr is the scheduler's
 * current value (muzero_amd/train.py evaluates the reference's schedulers on the host). */
int mzs_train_apply(mzs_handle* h, const float* grad_dev, int32_t optimizer, double lr, void* stream);

/* current weights in blob order, to a host or device buffer (synchronises the device) */
int mzs_get_weights(mzs_handle* h, float* blob_out, size_t n_floats);

/* ---- self-play arena: what sits on either side of the search ---------
 * n_slots concurrent episodes of a built-in environment, their trajectories in HBM (n_rows
 * trajectory rows of max_moves moves: the Game objects' lists, game.py:46-55), and the gather of
 * training batches from them. */
#define MZS_ENV_CARTPOLE 0 /* Gymnasium CartPole-v1 restated from its equations (include/mzs_env.h) */

typedef struct mzs_arena mzs_arena;
typedef struct mzs_arena_config {
    int32_t device;
    int32_t env;        /* MZS_ENV_*                                                   */
    int32_t n_slots;    /* G: episodes played at once (= games per search batch)       */
    int32_t n_rows;     /* trajectory rows kept in HBM (the replay window), >= n_slots  */
    int32_t max_moves;  /* limit_of_game_play (gameplay section of the config)         */
} mzs_arena_config;

int mzs_arena_create(const mzs_arena_config* cfg, mzs_arena** out);
void mzs_arena_destroy(mzs_arena* arena);
int32_t mzs_arena_obs_dim(const mzs_arena* arena);
int32_t mzs_arena_n_actions(const mzs_arena* arena);

/* Start of a round of n_games episodes (learning_cycle's number_of_self_play_before_training
 * play_game calls, self_play.py:230-236) recorded in trajectory rows [row0, row0 + n_games):
 * Game() + Game.observation(iteration=0) (game.py:126-141).  Game i is reset from the counter-based
 * generator keyed by (seeds_dev[i], episode).  The first min(n_slots, n_games) games start now, in
 * slots 0, 1, ...; every later game starts in the slot freed by a finished one, so the n_slots
 * searches of a move stay full until the round runs out of games.  obs_dev [n_slots][obs_dim]
 * receives the first observations. */
int mzs_arena_begin(mzs_arena* arena, int32_t row0, int32_t n_games, const int64_t* seeds_dev, int32_t episode,
                    float* obs_dev, void* stream);

/* More games behind the end of the queue mzs_arena_begin started (learning_cycle's NEXT iteration
 * of play_game calls, self_play.py:217-236, queued while the current one still plays): game i is
 * recorded in trajectory row (v + i) % n_rows, v = the queue's end — rows are a ring, the caller
 * keeps n_rows large enough that rows still in its replay window are not overwritten.  The games
 * start in the slots freed by finished ones, in queue order, so the lock-step searches stay full
 * across iteration boundaries (a round that ends with its queue empty runs its last moves on
 * ever fewer live slots).  `episode` keys the games' random numbers, `temperature` is their
 * action-selection temperature (NaN: the temperature argument of each mzs_arena_step).
 * MZS_ARENA_QUEUE: i32 [3] virtual rows {next unplayed, queue end, done_upto}: every game below
 * done_upto has finished. */
int mzs_arena_enqueue(mzs_arena* arena, int32_t n_games, const int64_t* seeds_dev, int32_t episode,
                      double temperature, void* stream);

/* One move of every live slot, fed with this move's search results (mzs_results layout):
 * Game.store_search_statistics (game.py:154-168), Game.policy_step (game.py:176-225: temperature
 * rescale, np.random.choice above temperature 0.1 else argmax, env.step, append to the lists);
 * num_simulations = the search's simulation count: with 0 the policy handed to policy_step is the
 * root priors instead of the visit counts (monte_carlo_tree_search.py:218-221);
 * then free slots take the round's next unplayed games in slot order.  obs_dev is overwritten
 * with the next observations (the reset observation for a newly started game);
 * *alive_count_dev = slots with a running game afterwards (0: the round is complete — play_game's
 * loop condition, self_play.py:62, has ended every game).  MZS_ARENA_STREAM_SEED holds, per slot,
 * the seed of the RandomState stream its next search draws from (mzs_fill_streams). */
int mzs_arena_step(mzs_arena* arena, const int32_t* visits_dev, const float* root_value_dev,
                   const double* root_prior_dev, double temperature, int32_t num_simulations, float* obs_dev,
                   int32_t* alive_count_dev, void* stream);

/* n_moves lock-step moves in one call: per move mzs_search_streams on the slots' observations
 * (obs_dev) and stream seeds, then mzs_arena_step on that search's results.  The loop of
 * self_play.py:62-73 for every slot at once, enqueued on `stream` without a host synchronisation
 * (the caller reads *alive_count_dev when it wants to know whether games are still running).
 * dirichlet_dev / noise_dev / noise_stride: scratch as for mzs_search_streams. */
int mzs_arena_play(mzs_arena* arena, mzs_handle* h, int32_t n_moves, int32_t num_simulations, int32_t train,
                   double alpha, double lo, double hi, int32_t prefix_draws, double temperature, float* obs_dev,
                   double* dirichlet_dev, double* noise_dev, int64_t noise_stride, int32_t* alive_count_dev,
                   void* stream);

/* What lock-step move `move` (0 = the first mzs_arena_step / move of mzs_arena_play after
 * mzs_arena_begin) left behind: out6 = {move, next unplayed, queue end, done_upto (MZS_ARENA_QUEUE),
 * slots with a running game, most draws any slot's search consumed (0 from mzs_arena_step, which is
 * not told)}.  The move's last kernel stores the entry into pinned host memory, so this is a plain
 * host read: no device call, no synchronisation, the stream keeps running.  It is the caller's
 * business to know that the move has finished (an event recorded behind it): MZS_E_STATE if it has
 * not, or if more than 256 later moves have overwritten the entry.  This is how the self-play loop
 * of self_play.py:62 (`while not terminal and counter < limit`) is watched for thousands of games
 * without draining the GPU's queue at every look. */
int mzs_arena_status(mzs_arena* arena, int32_t move, int32_t* out6);

/* ReplayBuffer.sample_batch's per-sample work (replay_buffer.py:147-151: make_image,
 * fill_gap_empty_action, Game.make_target game.py:237-260) for `batch` (row, position) pairs
 * chosen by the caller, written in the layout mzs_train_grad reads.  discount_pow_dev [td+1] =
 * discount ** i evaluated on the host as Python does. */
int mzs_arena_make_batch(mzs_arena* arena, int32_t batch, const int32_t* row_pos_dev, int32_t unroll_steps,
                         int32_t td_steps, const double* discount_pow_dev, float* obs_dev, float* actions_dev,
                         float* target_value_dev, float* target_reward_dev, float* target_policy_dev,
                         void* stream);

/* Game.make_priority (game.py:262-284) for trajectory rows [row0, row0 + n_rows): priority_dev
 * [n_rows][out_stride] f64 = |root_value - n-step value target| per position (0 from the end of
 * the game on; out_stride may be smaller than max_moves when the caller knows the longest game).
 * ReplayBuffer.save_game (replay_buffer.py:70-79) takes the per-game maximum of it. */
int mzs_arena_priority(mzs_arena* arena, int32_t row0, int32_t n_rows, int32_t td_steps,
                       const double* discount_pow_dev, int32_t out_stride, double* priority_dev, void* stream);

/* The same priorities without the zeros (Game.make_priority returns one array of game_length
 * entries per game, game.py:262-284): game r's entries start at priority_dev[offsets_dev[r]],
 * offsets_dev [n_rows] i64 = exclusive prefix sum of the rows' lengths (made by the caller, who
 * sizes priority_dev from it), longest = the longest of those games. */
int mzs_arena_priority_packed(mzs_arena* arena, int32_t row0, int32_t n_rows, int32_t td_steps,
                              const double* discount_pow_dev, int32_t longest, const int64_t* offsets_dev,
                              double* priority_dev, void* stream);

#define MZS_ARENA_LENGTH 0     /* i32 [n_rows]                  game_length            */
#define MZS_ARENA_OBS 1        /* f32 [n_rows][T][obs_dim]      observations            */
#define MZS_ARENA_ACTION 2     /* i32 [n_rows][T]               action_history (index)  */
#define MZS_ARENA_REWARD 3     /* f32 [n_rows][T]               rewards                 */
#define MZS_ARENA_ROOT_VALUE 4 /* f32 [n_rows][T]               root_values             */
#define MZS_ARENA_POLICY 5     /* f32 [n_rows][T][A]            child_visits            */
#define MZS_ARENA_ALIVE 6      /* i32 [n_slots]                                         */
#define MZS_ARENA_STATE 7      /* f64 [n_slots][4]              environment state       */
#define MZS_ARENA_STREAM_SEED 8 /* i64 [n_slots]                seed of the next search streams */
#define MZS_ARENA_SLOT_ROW 9   /* i32 [n_slots]                 row the slot records into */
#define MZS_ARENA_QUEUE 10     /* i32 [3]                       virtual rows: next unplayed, queue end, done_upto */
void* mzs_arena_array(mzs_arena* arena, int32_t which);

/* ---- replay store, host side (CPU only; no GPU work, callable without a device).
 * ReplayBuffer.sample_position + the mouve_prio lookup of sample_batch (replay_buffer.py:100-118,
 * :155) for a batch: rows[b] / row_len[b] = prio_position of sample b's game; u[b] = the uniform
 * number np.random.choice would draw (NULL: positions in pos[] are given, uniform position
 * sampling); pos[b] <- the position, mouve[b] <- soft_prio_position[pos].  NumPy's float64
 * arithmetic (pairwise sum, sequential cumsum, cdf / cdf[-1], searchsorted right) restated. */
int mzs_replay_sample_positions(int32_t batch, const double* const* rows, const int32_t* row_len, const double* u,
                                int64_t* pos, double* mouve);
/* ReplayBuffer.update_value (replay_buffer.py:169-177): new_value [K][batch]; game_len[b] = the
 * sampled game's game_length, game_idx[b] its buffer index into prio_game. */
int mzs_replay_update_value(int32_t batch, int32_t num_unroll, double* const* rows, const int32_t* row_len,
                            const int32_t* game_len, const int64_t* game_idx, const int64_t* pos,
                            const double* new_value, double* prio_game);

const char* mzs_last_error(void);
const char* mzs_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MZSEARCH_H */
