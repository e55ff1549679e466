/*
 * mz_oracle.c — CPU oracle (test infrastructure, see mz_oracle.h).
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -mfma -fopenmp).
 */
#include "mz_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "../include/mzs_numeric.h"

/* ------------------------------------------------------------------ model */

/* contract "dot_bf16" (include/mzs_numeric.h): weights and inputs rounded to bfloat16, groups of
 * four consecutive k through mzs_dot4_bf16, even groups chained from the bias, odd groups from the
 * first odd group, one add */
static void dense_bf16(const float* W, const float* b, const float* x, int nout, int nin, int ldw, float* y) {
    const int ng = (nin + 3) / 4;
    unsigned short* xb = (unsigned short*)malloc(sizeof(unsigned short) * (size_t)ng * 4 * 2);
    unsigned short* wb = xb + (size_t)ng * 4;
    for (int k = 0; k < ng * 4; ++k) xb[k] = k < nin ? mzs_f2bf16(x[k]) : 0;
    for (int j = 0; j < nout; ++j) {
        const float* w = W + (size_t)j * ldw;
        for (int k = 0; k < ng * 4; ++k) wb[k] = k < nin ? mzs_f2bf16(w[k]) : 0;
        float e = b[j], o = 0.0f;
        for (int g = 0; g < ng; g += 2) e = e + mzs_dot4_bf16(wb + 4 * g, xb + 4 * g);
        if (ng > 1) o = mzs_dot4_bf16(wb + 4, xb + 4);
        for (int g = 3; g < ng; g += 2) o = o + mzs_dot4_bf16(wb + 4 * g, xb + 4 * g);
        y[j] = e + o;
    }
    free(xb);
}

/* contract "dot_bf16x3" (include/mzs_numeric.h): weights and inputs as hi + lo bfloat16 pairs, three
 * mzs_dot4_bf16 per group of four k (hi hi, hi lo, lo hi), four chains of fp32 additions */
static void dense_bf16x3(const float* W, const float* b, const float* x, int nout, int nin, int ldw, float* y) {
    const int ng = (nin + 3) / 4;
    unsigned short* xh = (unsigned short*)calloc((size_t)ng * 4 * 4, sizeof(unsigned short));
    unsigned short *xl = xh + (size_t)ng * 4, *wh = xl + (size_t)ng * 4, *wl = wh + (size_t)ng * 4;
    for (int k = 0; k < nin; ++k) mzs_bf16_split(x[k], xh + k, xl + k);
    for (int j = 0; j < nout; ++j) {
        const float* w = W + (size_t)j * ldw;
        for (int k = 0; k < nin; ++k) mzs_bf16_split(w[k], wh + k, wl + k);
        float e = b[j], o = 0.0f, s = 0.0f, t = 0.0f;
        for (int g = 0; g < ng; g += 2) e = e + mzs_dot4_bf16(wh + 4 * g, xh + 4 * g);
        if (ng > 1) o = mzs_dot4_bf16(wh + 4, xh + 4);
        for (int g = 3; g < ng; g += 2) o = o + mzs_dot4_bf16(wh + 4 * g, xh + 4 * g);
        s = mzs_dot4_bf16(wh, xl);
        t = mzs_dot4_bf16(wl, xh);
        for (int g = 1; g < ng; ++g) {
            s = s + mzs_dot4_bf16(wh + 4 * g, xl + 4 * g);
            t = t + mzs_dot4_bf16(wl + 4 * g, xh + 4 * g);
        }
        y[j] = (e + o) + (s + t);
    }
    free(xh);
}

/* contract "dot": even-k chain from the bias, odd-k chain from zero, one add.
 * W row-major [nout][ldw]. */
static void dense_f32(const float* W, const float* b, const float* x, int nout, int nin, int ldw,
                      float* y) {
    for (int j = 0; j < nout; ++j) {
        float e = b[j], o = 0.0f;
        const float* w = W + (size_t)j * ldw;
        for (int k = 0; k < nin; k += 2) e = mzs_fma(w[k], x[k], e);
        for (int k = 1; k < nin; k += 2) o = mzs_fma(w[k], x[k], o);
        y[j] = e + o;
    }
}

/* the model's precision picks the contract; every layer of every net goes through here */
static int g_bf16 = 0;
#pragma omp threadprivate(g_bf16)
static void dense(const float* W, const float* b, const float* x, int nout, int nin, int ldw, float* y) {
    if (g_bf16 == MZS_PRECISION_BF16X3)
        dense_bf16x3(W, b, x, nout, nin, ldw, y);
    else if (g_bf16)
        dense_bf16(W, b, x, nout, nin, ldw, y);
    else
        dense_f32(W, b, x, nout, nin, ldw, y);
}
/* a weight used outside a dot (the one-hot action column): rounded like every other weight under
 * "dot_bf16", exact under "dot" and "dot_bf16x3" */
static float weight_value(float w) { return g_bf16 == MZS_PRECISION_BF16 ? mzs_bf16_round(w) : w; }

static void elu_inplace(float* x, int n) {
    for (int i = 0; i < n; ++i) x[i] = mzs_elu(x[i]);
}

/* contract "sum": adjacent-pair tree over the zero-padded array (clobbers v[0..P)) */
static float tree_sum(float* v, int n) {
    int P = mzs_pow2_ceil(n);
    for (int i = n; i < P; ++i) v[i] = 0.0f;
    for (int len = P; len > 1; len >>= 1)
        for (int i = 0; i < len / 2; ++i) v[i] = v[2 * i] + v[2 * i + 1];
    return v[0];
}

/* softmax over n logits into p (contract exp + sum) */
static void softmax(const float* q, int n, float* p, float* scratch) {
    float m = q[0];
    for (int i = 1; i < n; ++i) m = q[i] > m ? q[i] : m;
    for (int i = 0; i < n; ++i) p[i] = mzs_expf(q[i] - m);
    memcpy(scratch, p, sizeof(float) * (size_t)n);
    float Z = tree_sum(scratch, n), rz = 1.0f / Z;
    for (int i = 0; i < n; ++i) p[i] = mzs_div_r2(p[i], Z, rz); /* contract quotient (mzs_numeric.h) */
}

void mzo_softmax(const float* q, int32_t n, float* p) {
    int P = mzs_pow2_ceil(n);
    float* s = (float*)malloc(sizeof(float) * (size_t)P);
    softmax(q, n, p, s);
    free(s);
}

/* inverse_transform_with_support (muzero_model.py:571-587): softmax, expectation
 * over the support i - n/2, inverse value transform. */
float mzo_support_to_scalar(const float* logits, int32_t n) {
    int P = mzs_pow2_ceil(n);
    float* p = (float*)malloc(sizeof(float) * (size_t)P * 2);
    float* s = p + P;
    softmax(logits, n, p, s);
    int half = n / 2;
    for (int i = 0; i < n; ++i) s[i] = (float)(i - half) * p[i];
    float y = tree_sum(s, n);
    free(p);
    return mzs_inverse_value_transform(y);
}

/* scale_to_bound_action (neural_network_mlp_model.py:135-143), one row */
void mzo_scale_to_bound(const float* x, int32_t n, float* y) {
    float mn = x[0], mx = x[0];
    for (int i = 1; i < n; ++i) {
        mn = x[i] < mn ? x[i] : mn;
        mx = x[i] > mx ? x[i] : mx;
    }
    float sc = mzs_bound_scale(mn, mx), rs = 1.0f / sc;
    for (int i = 0; i < n; ++i) y[i] = mzs_div_r2(x[i] - mn, sc, rs);
}

float mzo_expf(float x) { return mzs_expf(x); }
/* contract "dot_bf16" pieces, exported so that tests can pin them to recorded device outputs */
void mzo_dot4_bf16(const uint16_t* a, const uint16_t* b, int32_t n, float* out) {
    for (int i = 0; i < n; ++i) out[i] = mzs_dot4_bf16(a + 4 * (size_t)i, b + 4 * (size_t)i);
}
void mzo_round_bf16(const float* x, int32_t n, float* out) {
    for (int i = 0; i < n; ++i) out[i] = mzs_bf16_round(x[i]);
}
float mzo_elu(float x) { return mzs_elu(x); }

/* tied mid layer applied L times (neural_network_mlp_model.py:38-39) */
static void mid_layers(const mzs_dims* d, const float* W, const float* b, float* t, float* t2) {
    for (int l = 0; l < d->n_mid_layers; ++l) {
        dense(W, b, t, d->hidden_dim, d->hidden_dim, d->hidden_dim, t2);
        elu_inplace(t2, d->hidden_dim);
        memcpy(t, t2, sizeof(float) * (size_t)d->hidden_dim);
    }
}

typedef struct scratch {
    float *t, *t2, *q, *g; /* g: 3 gate pre-activation rows of the LSTM heads */
} scratch;

static scratch scratch_new(const mzs_dims* d) {
    scratch s;
    int n = d->hidden_dim > d->state_dim ? d->hidden_dim : d->state_dim;
    n = n > d->n_actions ? n : d->n_actions;
    n = mzs_pow2_ceil(n) * 2;
    s.t = (float*)malloc(sizeof(float) * (size_t)n * 6);
    s.t2 = s.t + n;
    s.q = s.t2 + n;
    s.g = s.q + n;
    return s;
}

/* ---- 'lstm_model' (neural_network_lstm_model.py): Linear, then an L-layer nn.LSTM stepped once
 * from zero state (the search calls it with one row = a length-1 unbatched sequence), h of the
 * last layer taken by extract_tensor (:5-8).  Gate pre-activation = (W_ih x + b_ih) + b_hh
 * (W_hh h0 = 0); rows of W_ih in torch order i, f, g, o. */
static void lstm_head(const mzs_dims* d, const float* blob, const mzs_lstm_head_offsets* h, const float* x,
                      int x_len, int extra_col, float* out, scratch* s) {
    const int H = d->hidden_dim, n = h->out_dim;
    dense(blob + h->lin_w, blob + h->lin_b, x, H, x_len, h->in_dim, s->t);
    if (extra_col >= 0) /* one-hot action column, as in dyn_one */
        for (int j = 0; j < H; ++j) s->t[j] = s->t[j] + weight_value(blob[h->lin_w + (size_t)j * h->in_dim + extra_col]);
    const float* cur = s->t;
    int in_l = H;
    for (int l = 0; l < d->n_mid_layers; ++l) {
        const float* W = blob + h->w_ih[l];
        const float *bi = blob + h->b_ih[l], *bh = blob + h->b_hh[l];
        dense(W, bi, cur, n, in_l, in_l, s->g);                                        /* i */
        dense(W + (size_t)2 * n * in_l, bi + 2 * n, cur, n, in_l, in_l, s->g + n);      /* g */
        dense(W + (size_t)3 * n * in_l, bi + 3 * n, cur, n, in_l, in_l, s->g + 2 * n);  /* o */
        for (int j = 0; j < n; ++j)
            s->t2[j] = mzs_lstm_cell0(s->g[j] + bh[j], s->g[n + j] + bh[2 * n + j], s->g[2 * n + j] + bh[3 * n + j]);
        memcpy(s->t, s->t2, sizeof(float) * (size_t)n);
        cur = s->t;
        in_l = n;
    }
    memcpy(out, cur, sizeof(float) * (size_t)n);
}

/* Representation_function.forward (neural_network_mlp_model.py:43-44) */
static void repr_one(const mzs_dims* d, const float* blob, const mzs_blob_offsets* o,
                     const float* obs, float* hidden, scratch* s) {
    if (d->model_kind == MZS_MODEL_VISION) { /* oracle/mz_vision_oracle.c */
        mzo_vision_representation(d, blob, obs, 1, hidden, NULL);
        return;
    }
    if (d->model_kind == MZS_MODEL_LSTM) { /* one Linear, no activation, no rescaling (lstm:19-21) */
        const mzs_lstm_offsets lo = mzs_lstm_layout(d);
        dense(blob + lo.r_w, blob + lo.r_b, obs, d->state_dim, d->obs_dim, d->obs_dim, hidden);
        return;
    }
    dense(blob + o->r_in_w, blob + o->r_in_b, obs, d->hidden_dim, d->obs_dim, d->obs_dim, s->t);
    elu_inplace(s->t, d->hidden_dim);
    mid_layers(d, blob + o->r_mid_w, blob + o->r_mid_b, s->t, s->t2);
    dense(blob + o->r_out_w, blob + o->r_out_b, s->t, d->state_dim, d->hidden_dim, d->hidden_dim,
          s->q);
    mzo_scale_to_bound(s->q, d->state_dim, hidden);
}

/* Dynamics_function.forward (neural_network_mlp_model.py:86-88) on cat(hidden, onehot(a))
 * + reward head through inverse_transform_with_support (muzero_model.py:877-879). */
static void dyn_one(const mzs_dims* d, const float* blob, const mzs_blob_offsets* o,
                    const float* hidden, int action, float* reward, float* next_hidden,
                    scratch* s) {
    int H = d->hidden_dim, Sd = d->state_dim, ld = Sd + d->n_actions;
    if (d->model_kind == MZS_MODEL_VISION) {
        const int32_t a32 = action;
        mzo_vision_dynamics(d, blob, hidden, &a32, 1, reward, next_hidden, NULL);
        return;
    }
    if (d->model_kind == MZS_MODEL_LSTM) {
        const mzs_lstm_offsets lo = mzs_lstm_layout(d);
        lstm_head(d, blob, &lo.rew, hidden, Sd, Sd + action, s->q, s);
        *reward = mzo_support_to_scalar(s->q, Sd);
        lstm_head(d, blob, &lo.st, hidden, Sd, Sd + action, s->q, s);
        memcpy(next_hidden, s->q, sizeof(float) * (size_t)Sd);
        return;
    }
    const float* W = blob + o->d_in_w;
    dense(W, blob + o->d_in_b, hidden, H, Sd, ld, s->t);
    /* one-hot column: fma(1, w, acc) == acc + w; the zero columns add nothing */
    for (int j = 0; j < H; ++j) s->t[j] = s->t[j] + weight_value(W[(size_t)j * ld + Sd + action]);
    elu_inplace(s->t, H);
    mid_layers(d, blob + o->d_mid_w, blob + o->d_mid_b, s->t, s->t2);
    dense(blob + o->d_rew_w, blob + o->d_rew_b, s->t, Sd, H, H, s->q);
    *reward = mzo_support_to_scalar(s->q, Sd);
    dense(blob + o->d_st_w, blob + o->d_st_b, s->t, Sd, H, H, s->q);
    mzo_scale_to_bound(s->q, Sd, next_hidden);
}

/* Prediction_function.forward (neural_network_mlp_model.py:127-128) + softmax policy
 * (muzero_model.py:898) + value through inverse_transform_with_support (:903). */
static void pred_one(const mzs_dims* d, const float* blob, const mzs_blob_offsets* o,
                     const float* hidden, float* policy, float* value, scratch* s) {
    int H = d->hidden_dim, Sd = d->state_dim, A = d->n_actions;
    if (d->model_kind == MZS_MODEL_VISION) {
        mzo_vision_prediction(d, blob, hidden, 1, policy, value);
        return;
    }
    if (d->model_kind == MZS_MODEL_LSTM) {
        const mzs_lstm_offsets lo = mzs_lstm_layout(d);
        lstm_head(d, blob, &lo.pol, hidden, Sd, -1, s->q, s);
        softmax(s->q, A, policy, s->t2);
        lstm_head(d, blob, &lo.val, hidden, Sd, -1, s->q, s);
        *value = mzo_support_to_scalar(s->q, Sd);
        return;
    }
    dense(blob + o->p_in_w, blob + o->p_in_b, hidden, H, Sd, Sd, s->t);
    elu_inplace(s->t, H);
    mid_layers(d, blob + o->p_mid_w, blob + o->p_mid_b, s->t, s->t2);
    dense(blob + o->p_pol_w, blob + o->p_pol_b, s->t, A, H, H, s->q);
    softmax(s->q, A, policy, s->t2);
    dense(blob + o->p_val_w, blob + o->p_val_b, s->t, Sd, H, H, s->q);
    *value = mzo_support_to_scalar(s->q, Sd);
}

void mzo_representation(const mzs_dims* d, const float* blob, const float* obs, int32_t B,
                        float* hidden) {
    if (d->model_kind == MZS_MODEL_VISION) {
        mzo_vision_representation(d, blob, obs, B, hidden, NULL);
        return;
    }
    mzs_blob_offsets o = mzs_blob_layout(d);
    scratch s = scratch_new(d);
    g_bf16 = d->precision; /* 0, MZS_PRECISION_BF16, MZS_PRECISION_BF16X3 */
    for (int i = 0; i < B; ++i)
        repr_one(d, blob, &o, obs + (size_t)i * d->obs_dim, hidden + (size_t)i * d->state_dim, &s);
    free(s.t);
}

void mzo_dynamics(const mzs_dims* d, const float* blob, const float* hidden,
                  const int32_t* action, int32_t B, float* reward, float* next_hidden) {
    if (d->model_kind == MZS_MODEL_VISION) {
        mzo_vision_dynamics(d, blob, hidden, action, B, reward, next_hidden, NULL);
        return;
    }
    mzs_blob_offsets o = mzs_blob_layout(d);
    scratch s = scratch_new(d);
    g_bf16 = d->precision; /* 0, MZS_PRECISION_BF16, MZS_PRECISION_BF16X3 */
    for (int i = 0; i < B; ++i)
        dyn_one(d, blob, &o, hidden + (size_t)i * d->state_dim, action[i], reward + i,
                next_hidden + (size_t)i * d->state_dim, &s);
    free(s.t);
}

void mzo_prediction(const mzs_dims* d, const float* blob, const float* hidden, int32_t B,
                    float* policy, float* value) {
    if (d->model_kind == MZS_MODEL_VISION) {
        mzo_vision_prediction(d, blob, hidden, B, policy, value);
        return;
    }
    mzs_blob_offsets o = mzs_blob_layout(d);
    scratch s = scratch_new(d);
    g_bf16 = d->precision; /* 0, MZS_PRECISION_BF16, MZS_PRECISION_BF16X3 */
    for (int i = 0; i < B; ++i)
        pred_one(d, blob, &o, hidden + (size_t)i * d->state_dim, policy + (size_t)i * d->n_actions,
                 value + i, &s);
    free(s.t);
}

/* ----------------------------------------------------------------- search */

typedef struct tree {
    int32_t* visit;      /* Node.visit_count                                 */
    float* vsum;         /* Node.value_sum                                   */
    float* prior;        /* Node.prior (np.float32 softmax output)           */
    float* reward;       /* Node.reward                                      */
    int32_t* child_base; /* replaces Node.children: first child, 0 = leaf    */
    double* root_prior;  /* root children priors after noise (np.float64)    */
    float* hidden;       /* [S+1][Sd] Node.hidden_state per expansion        */
    int32_t* path;       /* search_path                                      */
    float mn, mx;        /* MinMaxStats                                      */
} tree;

static int search_one(const mzo_search_cfg* cfg, const float* blob, const mzs_blob_offsets* bo,
                      const mzo_table_model* tbl, const float* obs, const double* dirichlet,
                      const double* noise, int64_t noise_cap, int g, tree* T, scratch* sc,
                      const mzo_search_out* out) {
    const mzs_dims* d = &cfg->dims;
    const int A = d->n_actions, S = cfg->num_simulations, Sd = mzs_hidden_floats(d); /* floats per hidden state */
    const size_t obs_floats = d->model_kind == MZS_MODEL_VISION ? (size_t)MZS_VIS_C * d->obs_dim * d->obs_dim
                                                                 : (size_t)d->obs_dim;
    const int N = mzs_nodes_per_game(A, S);
    const float g32 = (float)cfg->discount;               /* python float * np.float32 -> f32 */
    const float omf32 = (float)(1.0 - cfg->exploration_fraction);
    int64_t consumed = 0, path_total = 0;
    double min_margin = INFINITY;
    float policy[64], value = 0.0f, reward = 0.0f;
    float* pol = A <= 64 ? policy : (float*)malloc(sizeof(float) * (size_t)A);

    memset(T->visit, 0, sizeof(int32_t) * (size_t)N);
    memset(T->vsum, 0, sizeof(float) * (size_t)N);
    memset(T->prior, 0, sizeof(float) * (size_t)N);
    memset(T->reward, 0, sizeof(float) * (size_t)N);
    memset(T->child_base, 0, sizeof(int32_t) * (size_t)N);
    T->mn = INFINITY;                                      /* MinMaxStats (mcts.py:24-26) */
    T->mx = -INFINITY;

    /* root: representation + prediction (mcts.py:88-110), expand (:113-123) */
    if (tbl) {
        memcpy(pol, tbl->root_policy + (size_t)g * A, sizeof(float) * (size_t)A);
    } else {
        repr_one(d, blob, bo, obs + (size_t)g * obs_floats, T->hidden, sc);
        pred_one(d, blob, bo, T->hidden, pol, &value, sc);
    }
    T->child_base[0] = 1;
    for (int a = 0; a < A; ++a) {
        T->prior[1 + a] = pol[a];
        /* noise (mcts.py:134-145): f32 product widened, f64 sum */
        if (cfg->train && S > 0)
            T->root_prior[a] =
                (double)(pol[a] * omf32) + dirichlet[(size_t)g * A + a] * cfg->exploration_fraction;
        else
            T->root_prior[a] = (double)pol[a];
    }

    for (int s = 0; s < S; ++s) {
        /* descent (mcts.py:172-177) */
        int node = 0, depth = 0, action = 0, base = 0;
        T->path[depth++] = 0;
        while (T->child_base[node] != 0) {
            base = T->child_base[node];
            const double tn = cfg->pbc_table[T->visit[node]];
            int best = -1;
            double best_score = 0.0, second = -INFINITY;
            if (consumed + A > noise_cap) return -2;
            for (int a = 0; a < A; ++a) {
                /* ucb_score (mcts.py:155-163) */
                int c = base + a, n = T->visit[c];
                double pd = node == 0 ? T->root_prior[a] : (double)T->prior[c];
                double ps = (tn * pd) / (double)(n + 1);
                double vs = 0.0;
                if (n > 0) {
                    float q = T->vsum[c] / (float)n;
                    float x = g32 * q;
                    x = T->reward[c] + x;
                    if (T->mx > T->mn) x = (x - T->mn) / (T->mx - T->mn);
                    vs = (double)x;
                }
                double score = (ps + vs) + noise[consumed + a];
                /* max over (score, action) tuples: ties -> larger action (mcts.py:167) */
                if (best < 0 || score >= best_score) {
                    if (best >= 0) second = best_score > second ? best_score : second;
                    best = a;
                    best_score = score;
                } else if (score > second) {
                    second = score;
                }
            }
            if (A > 1 && T->visit[node] > 0 && best_score - second < min_margin)
                min_margin = best_score - second;
            consumed += A;
            action = best;
            node = base + best;
            T->path[depth++] = node;
        }
        path_total += depth - 1;
        /* dynamics on the parent's hidden state, prediction on the new one (mcts.py:180-192) */
        const int e_parent = (base - 1) / A, e_new = s + 1;
        if (tbl) {
            reward = tbl->sim_reward[(size_t)g * S + s];
            value = tbl->sim_value[(size_t)g * S + s];
            memcpy(pol, tbl->sim_prior + ((size_t)g * S + s) * A, sizeof(float) * (size_t)A);
        } else {
            dyn_one(d, blob, bo, T->hidden + (size_t)e_parent * Sd, action, &reward,
                    T->hidden + (size_t)e_new * Sd, sc);
            pred_one(d, blob, bo, T->hidden + (size_t)e_new * Sd, pol, &value, sc);
        }
        /* expand (mcts.py:186-187, :195-200) */
        T->reward[node] = reward;
        T->child_base[node] = 1 + e_new * A;
        for (int a = 0; a < A; ++a) T->prior[1 + e_new * A + a] = pol[a];
        if (out->sim_leaf) out->sim_leaf[(size_t)g * S + s] = node;
        /* backup (mcts.py:203-209): fp32, product and sum rounded separately */
        float v = value;
        for (int i = depth - 1; i >= 0; --i) {
            int b = T->path[i];
            T->vsum[b] = T->vsum[b] + v;
            T->visit[b] += 1;
            float q = T->vsum[b] / (float)T->visit[b];
            T->mx = q > T->mx ? q : T->mx;
            T->mn = q < T->mn ? q : T->mn;
            float gv = g32 * v;
            v = T->reward[b] + gv;
        }
    }

    if (out->visits)
        for (int a = 0; a < A; ++a) out->visits[(size_t)g * A + a] = T->visit[1 + a];
    if (out->root_value)
        out->root_value[g] = T->visit[0] == 0 ? 0.0f : T->vsum[0] / (float)T->visit[0];
    if (out->root_prior) memcpy(out->root_prior + (size_t)g * A, T->root_prior, sizeof(double) * (size_t)A);
    if (out->consumed) out->consumed[g] = consumed;
    if (out->path_len) out->path_len[g] = path_total;
    if (out->node_visit) memcpy(out->node_visit + (size_t)g * N, T->visit, sizeof(int32_t) * (size_t)N);
    if (out->node_vsum) memcpy(out->node_vsum + (size_t)g * N, T->vsum, sizeof(float) * (size_t)N);
    if (out->node_reward) memcpy(out->node_reward + (size_t)g * N, T->reward, sizeof(float) * (size_t)N);
    if (out->node_prior) memcpy(out->node_prior + (size_t)g * N, T->prior, sizeof(float) * (size_t)N);
    if (out->child_base) memcpy(out->child_base + (size_t)g * N, T->child_base, sizeof(int32_t) * (size_t)N);
    if (out->minmax) {
        out->minmax[(size_t)g * 2] = T->mn;
        out->minmax[(size_t)g * 2 + 1] = T->mx;
    }
    if (out->min_margin) out->min_margin[g] = min_margin;
    if (out->hidden && !tbl)
        memcpy(out->hidden + (size_t)g * (S + 1) * Sd, T->hidden, sizeof(float) * (size_t)(S + 1) * Sd);
    if (pol != policy) free(pol);
    return 0;
}

int mzo_search(const mzo_search_cfg* cfg, const float* blob, const mzo_table_model* tbl,
               const float* obs, const double* dirichlet, const double* noise,
               int64_t noise_stride, int32_t G, const mzo_search_out* out) {
    const mzs_dims* d = &cfg->dims;
    if (!cfg || !out || G < 0 || d->n_actions < 1 || cfg->num_simulations < 0) return -1;
    if (!tbl && (!blob || !obs)) return -1;
    if (cfg->train && cfg->num_simulations > 0 && !dirichlet) return -1;
    if (cfg->num_simulations > 0 && (!noise || !cfg->pbc_table)) return -1;
    const mzs_blob_offsets bo = mzs_blob_layout(d);
    const int N = mzs_nodes_per_game(d->n_actions, cfg->num_simulations);
    const int S = cfg->num_simulations;
    int rc = 0;
    int nt = cfg->n_threads > 0 ? cfg->n_threads : 1;
#pragma omp parallel num_threads(nt)
    {
        tree T;
        T.visit = (int32_t*)malloc(sizeof(int32_t) * (size_t)N * 3);
        T.child_base = T.visit + N;
        T.path = T.child_base + N;
        T.vsum = (float*)malloc(sizeof(float) * (size_t)N * 3);
        T.prior = T.vsum + N;
        T.reward = T.prior + N;
        T.root_prior = (double*)malloc(sizeof(double) * (size_t)d->n_actions);
        T.hidden = (float*)malloc(sizeof(float) * (size_t)(S + 1) * mzs_hidden_floats(d));
        scratch sc = scratch_new(d);
        g_bf16 = d->precision; /* 0, MZS_PRECISION_BF16, MZS_PRECISION_BF16X3 */ /* threadprivate */
#pragma omp for schedule(static)
        for (int g = 0; g < G; ++g) {
            int r = search_one(cfg, blob, &bo, tbl, obs, dirichlet,
                               noise ? noise + (size_t)g * noise_stride : NULL, noise_stride, g, &T,
                               &sc, out);
            if (r != 0) {
#pragma omp critical
                rc = r;
            }
        }
        free(sc.t);
        free(T.hidden);
        free(T.root_prior);
        free(T.vsum);
        free(T.visit);
    }
    return rc;
}
