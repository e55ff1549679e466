/*
 * mzs_layout.h — packed weight-blob layout and node numbering shared by the
 * C-ABI library (muzero_amd/csrc) and the CPU oracle (oracle/).
 *
 * Weight blob: fp32, every matrix row-major [out][in] exactly as torch stores
 * nn.Linear.weight, every vector [out].  Order (L = n_mid_layers; the mid layer
 * is ONE tied tensor per net, applied L times — the reference repeats the same
 * linear_mid object, neural_network_mlp_model.py:33-39, :75-84, :116-122):
 *
 *   representation (neural_network_mlp_model.py:7-44)
 *     r_in_w [H][obs]   r_in_b [H]   (r_mid_w [H][H]  r_mid_b [H] if L>0)
 *     r_out_w [Sd][H]   r_out_b [Sd]
 *   dynamics (neural_network_mlp_model.py:49-88; trunk shared by both heads)
 *     d_in_w [H][Sd+A]  d_in_b [H]   (d_mid_w [H][H]  d_mid_b [H] if L>0)
 *     d_rew_w [Sd][H]   d_rew_b [Sd]   d_st_w [Sd][H]   d_st_b [Sd]
 *   prediction (neural_network_mlp_model.py:92-128; trunk shared by both heads)
 *     p_in_w [H][Sd]    p_in_b [H]   (p_mid_w [H][H]  p_mid_b [H] if L>0)
 *     p_pol_w [A][H]    p_pol_b [A]    p_val_w [Sd][H]  p_val_b [Sd]
 *
 * Node numbering of one game's tree (replaces the Node objects of
 * monte_carlo_tree_search.py:6-20): node 0 is the root; the A children created
 * by the e-th expansion live at 1 + e*A + a.  Expansion 0 is the root expansion
 * (monte_carlo_tree_search.py:113-123), expansion s+1 belongs to simulation s
 * (:195-200).  Nodes per game: 1 + A*(S+1).
 */
#ifndef MZS_LAYOUT_H
#define MZS_LAYOUT_H

#include <stddef.h>
#include <stdint.h>

/* the vision layout is also computed inside kernels */
#if defined(__HIPCC__)
#define MZS_LFN __host__ __device__ static inline
#else
#define MZS_LFN static inline
#endif

typedef struct mzs_dims {
    int32_t obs_dim;      /* flattened observation length                    */
    int32_t n_actions;    /* A                                               */
    int32_t state_dim;    /* S_dim: hidden-state width == support size       */
    int32_t hidden_dim;   /* H                                               */
    int32_t n_mid_layers; /* L: how many times the tied mid layer is applied (MLP);
                             num_layers of every nn.LSTM (LSTM model, 1..MZS_LSTM_MAX_LAYERS) */
    int32_t model_kind;   /* MZS_MODEL_MLP (0, 'mlp_model'), MZS_MODEL_LSTM ('lstm_model') or
                             MZS_MODEL_VISION ('vision_model')                                    */
    int32_t precision;    /* MZS_PRECISION_F32 (0), MZS_PRECISION_BF16 or MZS_PRECISION_BF16X3:
                             Muzero(type_format=...), muzero_model.py:194-204.  bf16 = weights and layer
                             inputs rounded to bfloat16, fp32 accumulation and epilogues; bf16x3 = both
                             kept as a hi + lo PAIR of bfloat16s, three matrix products per layer
                             (include/mzs_numeric.h "dot_bf16" / "dot_bf16x3")                    */
} mzs_dims;

#define MZS_PRECISION_F32 0
#define MZS_PRECISION_BF16 1   /* fast: within 1e-2 of the reference on the CartPole checkpoints 123 / 133 */
#define MZS_PRECISION_BF16X3 2 /* compensated: within 1e-2 (measured 2e-3) on every fixture weight set   */

#define MZS_MODEL_MLP 0
#define MZS_MODEL_LSTM 1
#define MZS_LSTM_MAX_LAYERS 4

typedef struct mzs_blob_offsets {
    size_t r_in_w, r_in_b, r_mid_w, r_mid_b, r_out_w, r_out_b;
    size_t d_in_w, d_in_b, d_mid_w, d_mid_b, d_rew_w, d_rew_b, d_st_w, d_st_b;
    size_t p_in_w, p_in_b, p_mid_w, p_mid_b, p_pol_w, p_pol_b, p_val_w, p_val_b;
    size_t total;
} mzs_blob_offsets;

static inline mzs_blob_offsets mzs_blob_layout(const mzs_dims* d) {
    mzs_blob_offsets o;
    size_t H = (size_t)d->hidden_dim, Sd = (size_t)d->state_dim, A = (size_t)d->n_actions;
    size_t ob = (size_t)d->obs_dim, mid = d->n_mid_layers > 0 ? 1 : 0, p = 0;
    o.r_in_w = p;  p += H * ob;
    o.r_in_b = p;  p += H;
    o.r_mid_w = p; p += mid * H * H;
    o.r_mid_b = p; p += mid * H;
    o.r_out_w = p; p += Sd * H;
    o.r_out_b = p; p += Sd;
    o.d_in_w = p;  p += H * (Sd + A);
    o.d_in_b = p;  p += H;
    o.d_mid_w = p; p += mid * H * H;
    o.d_mid_b = p; p += mid * H;
    o.d_rew_w = p; p += Sd * H;
    o.d_rew_b = p; p += Sd;
    o.d_st_w = p;  p += Sd * H;
    o.d_st_b = p;  p += Sd;
    o.p_in_w = p;  p += H * Sd;
    o.p_in_b = p;  p += H;
    o.p_mid_w = p; p += mid * H * H;
    o.p_mid_b = p; p += mid * H;
    o.p_pol_w = p; p += A * H;
    o.p_pol_b = p; p += A;
    o.p_val_w = p; p += Sd * H;
    o.p_val_b = p; p += Sd;
    o.total = p;
    return o;
}

/* 'lstm_model' (neural_network_lstm_model.py:10-80), blob in state_dict order, torch layouts:
 *   representation  state_norm.weight [Sd][obs], .bias [Sd]            (one Linear, :19-21)
 *   dynamics        reward:               0.weight [H][Sd+A], 0.bias [H], then per LSTM layer l
 *                     weight_ih_l [4*out][in_l], weight_hh_l [4*out][out], bias_ih_l, bias_hh_l [4*out]
 *                     (out = Sd; in_0 = H, in_l = out; gate rows in torch order i, f, g, o)
 *                   next_state_normalized: the same again (its own Linear, :33-47)
 *   prediction      policy (out = A), value (out = Sd), each Linear [H][Sd] + LSTM (:63-77)
 * Every call of the search feeds the LSTM a length-1 sequence with zero initial state, so
 * weight_hh and the forget gate never contribute; they are carried in the blob for layout
 * fidelity only. */
typedef struct mzs_lstm_head_offsets {
    size_t lin_w, lin_b;
    size_t w_ih[MZS_LSTM_MAX_LAYERS], w_hh[MZS_LSTM_MAX_LAYERS];
    size_t b_ih[MZS_LSTM_MAX_LAYERS], b_hh[MZS_LSTM_MAX_LAYERS];
    int32_t in_dim, out_dim;
} mzs_lstm_head_offsets;

typedef struct mzs_lstm_offsets {
    size_t r_w, r_b;
    mzs_lstm_head_offsets rew, st, pol, val;
    size_t total;
} mzs_lstm_offsets;

static inline size_t mzs_lstm_head_layout(mzs_lstm_head_offsets* h, size_t p, int in_dim, int H, int out,
                                          int n_layers) {
    h->in_dim = in_dim;
    h->out_dim = out;
    h->lin_w = p; p += (size_t)H * (size_t)in_dim;
    h->lin_b = p; p += (size_t)H;
    for (int l = 0; l < n_layers && l < MZS_LSTM_MAX_LAYERS; ++l) {
        const size_t in_l = l == 0 ? (size_t)H : (size_t)out;
        h->w_ih[l] = p; p += 4 * (size_t)out * in_l;
        h->w_hh[l] = p; p += 4 * (size_t)out * (size_t)out;
        h->b_ih[l] = p; p += 4 * (size_t)out;
        h->b_hh[l] = p; p += 4 * (size_t)out;
    }
    return p;
}

static inline mzs_lstm_offsets mzs_lstm_layout(const mzs_dims* d) {
    mzs_lstm_offsets o;
    const int H = d->hidden_dim, Sd = d->state_dim, A = d->n_actions, L = d->n_mid_layers;
    size_t p = 0;
    o.r_w = p; p += (size_t)Sd * (size_t)d->obs_dim;
    o.r_b = p; p += (size_t)Sd;
    p = mzs_lstm_head_layout(&o.rew, p, Sd + A, H, Sd, L);
    p = mzs_lstm_head_layout(&o.st, p, Sd + A, H, Sd, L);
    p = mzs_lstm_head_layout(&o.pol, p, Sd, H, A, L);
    p = mzs_lstm_head_layout(&o.val, p, Sd, H, Sd, L);
    o.total = p;
    return o;
}

/* 'vision_model' (neural_network_vision_model.py:41-300): ResNet-v2 blocks on a 98 x 98 x 3 image
 * (muzero_model.py:312-314), hidden state = 3 x 7 x 7 feature map, MLP heads on its 147 values.
 * mzs_dims: obs_dim = 98 (image side), state_dim = support size of the value / reward heads,
 * hidden_dim = H and n_mid_layers = L of the head MLPs (L >= 1 is also the number of times the
 * dynamics / prediction residual block is applied, :214, :284).
 * Blob: every DISTINCT tensor once (the reference's Sequentials repeat one Residual_block, one
 * BatchNorm2d and one Linear object in several places), conv weights [out][in][ky][kx]:
 *   residual block (:41-82): conv_a = `convolution_1` (first and last conv of the block), conv_b =
 *     `convolution_3` (the middle one), then the one BatchNorm2d: weight, bias, running_mean,
 *     running_var;   y = x + conv_a(R(conv_b(R(conv_a(R(x))))))  with R = relu(batchnorm(.))
 *   representation: Down_sample (:84-122): conv_in 3->1 stride 2, res_in x2, conv_out 1->3 stride 2,
 *     res_out x2, avgpool 3/2/1, res_out x3, avgpool; then res (:147-149).  The tensors of the
 *     unused down_sampling=False path (:141-142) follow, for checkpoint fidelity.
 *   dynamics (:167-230): conv 4->3, batchnorm, relu, res x L, relu -> next state; reward: 1x1 conv
 *     4->3 (+bias), flatten, Linear 147->H, relu, (Linear H->H, relu) x L [one tensor], Linear H->Sd.
 *     Input channel 3 is the action plane, constant (a + 1) / A (muzero_model.py:505-516).
 *   prediction (:234-300): res x L, then value and policy: 1x1 conv 3->3 (+bias), flatten, MLP. */
#define MZS_MODEL_VISION 2
#define MZS_VIS_C 3
#define MZS_VIS_SIDE 98
#define MZS_VIS_S1 49
#define MZS_VIS_S2 25
#define MZS_VIS_S3 13
#define MZS_VIS_S4 7
#define MZS_VIS_HIDDEN (MZS_VIS_C * MZS_VIS_S4 * MZS_VIS_S4)

typedef struct mzs_vis_bn {
    size_t w, b, mean, var;
} mzs_vis_bn;
typedef struct mzs_vis_res {
    size_t conv_a, conv_b;
    mzs_vis_bn bn;
    int32_t C;
} mzs_vis_res;
typedef struct mzs_vis_mlp {
    size_t in_w, in_b, mid_w, mid_b, out_w, out_b;
    int32_t n_out;
} mzs_vis_mlp;
typedef struct mzs_vision_offsets {
    size_t r_conv_in, r_conv_out, r_conv;
    mzs_vis_res r_res_in, r_res_out, r_res;
    mzs_vis_bn r_bn;
    size_t d_conv, d_conv_rew, d_conv_rew_b;
    mzs_vis_bn d_bn;
    mzs_vis_res d_res;
    mzs_vis_mlp d_fc;
    mzs_vis_res p_res;
    size_t p_conv_val, p_conv_val_b, p_conv_pol, p_conv_pol_b;
    mzs_vis_mlp p_v, p_p;
    size_t total;
} mzs_vision_offsets;

MZS_LFN size_t mzs_vis_bn_layout(mzs_vis_bn* b, size_t p, int C) {
    b->w = p; p += (size_t)C;
    b->b = p; p += (size_t)C;
    b->mean = p; p += (size_t)C;
    b->var = p; p += (size_t)C;
    return p;
}
MZS_LFN size_t mzs_vis_res_layout(mzs_vis_res* r, size_t p, int C) {
    r->C = C;
    r->conv_a = p; p += (size_t)C * C * 9;
    r->conv_b = p; p += (size_t)C * C * 9;
    return mzs_vis_bn_layout(&r->bn, p, C);
}
MZS_LFN size_t mzs_vis_mlp_layout(mzs_vis_mlp* m, size_t p, int H, int L, int n_out) {
    m->n_out = n_out;
    m->in_w = p; p += (size_t)H * MZS_VIS_HIDDEN;
    m->in_b = p; p += (size_t)H;
    m->mid_w = p; p += L > 0 ? (size_t)H * H : 0;
    m->mid_b = p; p += L > 0 ? (size_t)H : 0;
    m->out_w = p; p += (size_t)n_out * H;
    m->out_b = p; p += (size_t)n_out;
    return p;
}
MZS_LFN mzs_vision_offsets mzs_vision_layout(const mzs_dims* d) {
    mzs_vision_offsets o;
    const int C = MZS_VIS_C, H = d->hidden_dim, L = d->n_mid_layers, Sd = d->state_dim, A = d->n_actions;
    size_t p = 0;
    o.r_conv_in = p; p += (size_t)(C / 2) * C * 9;
    p = mzs_vis_res_layout(&o.r_res_in, p, C / 2);
    o.r_conv_out = p; p += (size_t)C * (C / 2) * 9;
    p = mzs_vis_res_layout(&o.r_res_out, p, C);
    p = mzs_vis_res_layout(&o.r_res, p, C);
    o.r_conv = p; p += (size_t)C * C * 9;
    p = mzs_vis_bn_layout(&o.r_bn, p, C);
    o.d_conv = p; p += (size_t)C * (C + 1) * 9;
    p = mzs_vis_bn_layout(&o.d_bn, p, C);
    p = mzs_vis_res_layout(&o.d_res, p, C);
    o.d_conv_rew = p; p += (size_t)C * (C + 1);
    o.d_conv_rew_b = p; p += (size_t)C;
    p = mzs_vis_mlp_layout(&o.d_fc, p, H, L, Sd);
    p = mzs_vis_res_layout(&o.p_res, p, C);
    o.p_conv_val = p; p += (size_t)C * C;
    o.p_conv_val_b = p; p += (size_t)C;
    o.p_conv_pol = p; p += (size_t)C * C;
    o.p_conv_pol_b = p; p += (size_t)C;
    p = mzs_vis_mlp_layout(&o.p_v, p, H, L, Sd);
    p = mzs_vis_mlp_layout(&o.p_p, p, H, L, A);
    o.total = p;
    return o;
}

/* floats in the weight blob of a model kind */
static inline size_t mzs_blob_total(const mzs_dims* d) {
    return d->model_kind == MZS_MODEL_VISION ? mzs_vision_layout(d).total
           : d->model_kind == MZS_MODEL_LSTM ? mzs_lstm_layout(d).total : mzs_blob_layout(d).total;
}

/* floats of one hidden state (the search's per-node state) */
MZS_LFN int32_t mzs_hidden_floats(const mzs_dims* d) {
    return d->model_kind == MZS_MODEL_VISION ? MZS_VIS_HIDDEN : d->state_dim;
}

static inline int32_t mzs_nodes_per_game(int32_t n_actions, int32_t num_simulations) {
    return 1 + n_actions * (num_simulations + 1);
}

/* Upper bound on tie-break draws one search can consume: A per level, and the
 * descent of simulation s visits at most s+1 levels (monte_carlo_tree_search.py:163,172). */
static inline int64_t mzs_max_draws(int32_t n_actions, int32_t num_simulations) {
    return (int64_t)n_actions * num_simulations * (num_simulations + 1) / 2;
}

#endif /* MZS_LAYOUT_H */
