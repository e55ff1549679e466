/*
 * mz_train_oracle.c — CPU oracle of the unroll-K training step (test infrastructure, see
 * mz_oracle.h).  Restates, for the MLP model family first:
 *   Muzero.compute_forward            muzero_model.py:609-647   (hidden-state gradient x0.5, :640)
 *   Muzero.evaluate_loss              muzero_model.py:650-734   (zip() drops the (K+1)-th prediction,
 *                                                                1/K gradient scale for k > 0 :657,:601)
 *   Muzero.transform_with_support     muzero_model.py:536-563   (single-bin encoding, kept as is)
 *   Loss_function.cross_entropy       neural_network_mlp_model.py:240-243 (softmax, + 1e-9)
 *   l2                                neural_network_mlp_model.py:280-285 (zip() truncates to the
 *                                                                shortest parameter list)
 *   torch.optim.Adam / SGD(momentum)  muzero_model.py:418-421
 * and the same step for the LSTM model family (neural_network_lstm_model.py; second half of this file).
 * Pinned by tests/golden/train_*.npz (made from the reference by make_train_fixtures.py and
 * make_lstm_train_fixtures.py).
 * Storage is fp32 like the reference; dot products accumulate in double, so torch-CPU and the
 * GPU path both sit within fp32 rounding of it (tolerances in tests/test_train_*.py).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "mz_oracle.h"
#include "../include/mzs_numeric.h"

typedef struct net {
    int nin, ld, H, L;
    const float *Win, *bin, *Wmid, *bmid;
    float *gWin, *gbin, *gWmid, *gbmid;
} net;

static float elu_f(float z) { return z > 0.0f ? z : (float)expm1((double)z); }

static void lin(const float* W, const float* b, const float* x, int nout, int nin, int ld, float* y) {
    for (int j = 0; j < nout; ++j) {
        double acc = b[j];
        for (int k = 0; k < nin; ++k) acc += (double)W[(size_t)j * ld + k] * x[k];
        y[j] = (float)acc;
    }
}

/* T[(L+1)][H]: activations after every ELU */
static void net_forward(const net* n, const float* x, float* T) {
    lin(n->Win, n->bin, x, n->H, n->nin, n->ld, T);
    for (int j = 0; j < n->H; ++j) T[j] = elu_f(T[j]);
    for (int l = 0; l < n->L; ++l) {
        float* t1 = T + (size_t)(l + 1) * n->H;
        lin(n->Wmid, n->bmid, T + (size_t)l * n->H, n->H, n->H, n->H, t1);
        for (int j = 0; j < n->H; ++j) t1[j] = elu_f(t1[j]);
    }
}

/* y += W^T d */
static void lin_back_input(const float* W, const float* d, int nout, int nin, int ld, float* gx) {
    for (int k = 0; k < nin; ++k) {
        double acc = 0.0;
        for (int j = 0; j < nout; ++j) acc += (double)W[(size_t)j * ld + k] * d[j];
        gx[k] = (float)acc;
    }
}

static void lin_back_weight(float* gW, float* gb, const float* d, const float* x, int nout, int nin,
                            int ld) {
    for (int j = 0; j < nout; ++j) {
        for (int k = 0; k < nin; ++k) gW[(size_t)j * ld + k] += d[j] * x[k];
        gb[j] += d[j];
    }
}

/* gT: gradient w.r.t. the last trunk activation (clobbered); gx: gradient w.r.t. the input */
static void net_backward(const net* n, const float* x, const float* T, float* gT, float* gx,
                         float* tmp) {
    for (int l = n->L; l >= 1; --l) {
        const float* t = T + (size_t)l * n->H;
        for (int j = 0; j < n->H; ++j) gT[j] *= t[j] > 0.0f ? 1.0f : t[j] + 1.0f;
        lin_back_weight(n->gWmid, n->gbmid, gT, T + (size_t)(l - 1) * n->H, n->H, n->H, n->H);
        lin_back_input(n->Wmid, gT, n->H, n->H, n->H, tmp);
        memcpy(gT, tmp, sizeof(float) * (size_t)n->H);
    }
    for (int j = 0; j < n->H; ++j) gT[j] *= T[j] > 0.0f ? 1.0f : T[j] + 1.0f;
    lin_back_weight(n->gWin, n->gbin, gT, x, n->H, n->nin, n->ld);
    if (gx) lin_back_input(n->Win, gT, n->H, n->nin, n->ld, gx);
}

typedef struct scale_rec {
    int imin, imax;
    float sc;
} scale_rec;

/* scale_to_bound_action forward (neural_network_mlp_model.py:135-143); first index on ties */
static void scale_forward(const float* z, int n, float* h, scale_rec* r) {
    r->imin = r->imax = 0;
    for (int i = 1; i < n; ++i) {
        if (z[i] < z[r->imin]) r->imin = i;
        if (z[i] > z[r->imax]) r->imax = i;
    }
    float sc = z[r->imax] - z[r->imin];
    if (sc < 1e-5f) sc = sc + 1e-5f;
    r->sc = sc;
    for (int i = 0; i < n; ++i) h[i] = (z[i] - z[r->imin]) / sc;
}

/* g: dL/dh  ->  gz: dL/dz */
static void scale_backward(const float* g, const float* h, int n, const scale_rec* r, float* gz) {
    double gsum = 0.0, ghs = 0.0;
    for (int i = 0; i < n; ++i) {
        gsum += g[i];
        ghs += (double)g[i] * h[i];
    }
    const double dsc = -ghs / r->sc;            /* dL/d(scale)            */
    const double dmn = -gsum / r->sc - dsc;     /* dL/d(min)  (scale = max - min) */
    for (int i = 0; i < n; ++i) gz[i] = g[i] / r->sc;
    gz[r->imin] += (float)dmn;
    gz[r->imax] += (float)dsc;
}

/* transform_with_support (muzero_model.py:536-563): one (index, weight) pair */
static void support_target(float x, int S, int* idx, float* weight) {
    const int sd = S / 2;
    const int rem = (int)(2.0 * ((double)S / 2.0 - (double)sd));
    float ax = fabsf(x);
    float sg0 = x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f);
    float t = sg0 * (sqrtf(ax + 1.0f) - 1.0f) + 0.001f * x;
    if (t < (float)-sd) t = (float)-sd;
    if (t > (float)(sd + rem)) t = (float)(sd + rem);
    float sg = t > 0.0f ? 1.0f : (t < 0.0f ? -1.0f : 0.0f);
    float fl = floorf(t), w = (ceilf(t) - t) * sg;
    long i = (long)(fl + (float)sd + 1.0f);
    if (i < -(S - 1)) i = -(S - 1);
    if (i > S - 1) i = S - 1;
    *idx = (int)i;
    *weight = w;
}

/* One loss term of evaluate_loss (muzero_model.py:650-700) on q = softmax(z) + 1e-9 (the prediction
 * transforms of every loss_type, muzero_model.py:350-403) against the target row p; gz += upstream *
 * dloss/dz; returns the loss.  mode = the criterion (neural_network_mlp_model.py:236-262):
 *   MZO_LOSS_CE    cross_entropy  -sum p log q                                  c = -p / q
 *   MZO_LOSS_KL    kldiv          sum p' (log p' - log q), p' = p + 1e-9        c = -p' / q
 *   MZO_LOSS_MSE   mse            mean (p - q)^2                                c = -2 (p - q) / n
 *   MZO_LOSS_ZERO  zero_loss      0                                             c = 0
 * with dloss/dz = s (c - sum c s), s = softmax(z). */
static int g_loss_value = MZO_LOSS_CE, g_loss_reward = MZO_LOSS_CE;
void mzo_train_set_loss(int value_mode, int reward_mode) { g_loss_value = value_mode, g_loss_reward = reward_mode; }

static float loss_term(int mode, const float* z, int n, const float* p, float upstream, float* gz, float* s) {
    float m = z[0];
    for (int i = 1; i < n; ++i) m = z[i] > m ? z[i] : m;
    double Z = 0.0;
    for (int i = 0; i < n; ++i) {
        s[i] = expf(z[i] - m);
        Z += s[i];
    }
    double loss = 0.0, dot = 0.0;
    for (int i = 0; i < n; ++i) {
        s[i] = (float)(s[i] / Z);
        const float q = s[i] + 1e-9f;
        double c = 0.0;
        if (mode == MZO_LOSS_CE) {
            loss += -(double)p[i] * log((double)q);
            c = (double)(-p[i] / q);
        } else if (mode == MZO_LOSS_KL) {
            const float pp = p[i] + 1e-9f;
            loss += (double)pp * (log((double)pp) - log((double)q));
            c = (double)(-pp / q);
        } else if (mode == MZO_LOSS_MSE) {
            const double d = (double)p[i] - (double)q;
            loss += d * d / n;
            c = -2.0 * d / n;
        }
        dot += c * s[i];
    }
    if (gz)
        for (int i = 0; i < n; ++i) {
            const float q = s[i] + 1e-9f;
            double c = 0.0;
            if (mode == MZO_LOSS_CE) c = (double)(-p[i] / q);
            else if (mode == MZO_LOSS_KL) c = (double)(-(p[i] + 1e-9f) / q);
            else if (mode == MZO_LOSS_MSE) c = -2.0 * ((double)p[i] - (double)q) / n;
            gz[i] += upstream * s[i] * (float)(c - dot);
        }
    return (float)loss;
}
/* the policy term: cross-entropy, against p + 1e-9 under 'general_kkc' (label = ["zero_clamp_transform"],
 * muzero_model.py:373-375; the one loss type whose value criterion is the KL divergence) */
static float ce_loss(const float* z, int n, const float* p, float upstream, float* gz, float* s) {
    if (g_loss_value != MZO_LOSS_KL) return loss_term(MZO_LOSS_CE, z, n, p, upstream, gz, s);
    float pc[64];
    for (int i = 0; i < n; ++i) pc[i] = p[i] + 1e-9f;
    return loss_term(MZO_LOSS_CE, z, n, pc, upstream, gz, s);
}

/* ================================================================== 'lstm_model' ==========
 * neural_network_lstm_model.py: representation = one Linear (:19-21, no activation, no
 * scale_to_bound); each of the four heads (dynamics reward / next state :33-47, prediction
 * policy / value :63-77) = Linear(in -> H), then nn.LSTM(H -> out, num_layers = L), h of the last
 * layer.  compute_forward hands each head a 2-D (B, in) tensor, which nn.LSTM takes as ONE
 * unbatched sequence of length B with a zero initial state: row b of a head's output depends on
 * rows 0..b of its input, and autograd runs back through the batch axis.  Kept as is.
 * torch's cell (gate rows i, f, g, o):  z = W_ih x_t + b_ih + W_hh h_{t-1} + b_hh,
 *   c_t = sig(f) c_{t-1} + sig(i) tanh(g),  h_t = sig(o) tanh(c_t). */
typedef struct head_rec { /* what one call of one head leaves behind for its backward pass */
    float* x;                         /* [B][in]                                   */
    float* u;                         /* [B][H]  the Linear's output               */
    float* gate[MZS_LSTM_MAX_LAYERS]; /* [B][4n] activated gates i, f, g, o        */
    float* c[MZS_LSTM_MAX_LAYERS];    /* [B][n]                                    */
    float* h[MZS_LSTM_MAX_LAYERS];    /* [B][n]                                    */
} head_rec;

static head_rec head_rec_new(const mzs_lstm_head_offsets* ho, int H, int L, int B) {
    head_rec r;
    const size_t n = (size_t)ho->out_dim;
    memset(&r, 0, sizeof(r));
    r.x = (float*)calloc((size_t)B * ho->in_dim, sizeof(float));
    r.u = (float*)calloc((size_t)B * H, sizeof(float));
    for (int l = 0; l < L; ++l) {
        r.gate[l] = (float*)calloc((size_t)B * 4 * n, sizeof(float));
        r.c[l] = (float*)calloc((size_t)B * n, sizeof(float));
        r.h[l] = (float*)calloc((size_t)B * n, sizeof(float));
    }
    return r;
}

static void head_rec_free(head_rec* r) {
    free(r->x), free(r->u);
    for (int l = 0; l < MZS_LSTM_MAX_LAYERS; ++l) free(r->gate[l]), free(r->c[l]), free(r->h[l]);
}

static float sigmoid_f(float z) { return (float)(1.0 / (1.0 + exp(-(double)z))); }

/* X [B][in] -> the head's output = r->h[L-1] ([B][n]) */
static void head_forward(const float* blob, const mzs_lstm_head_offsets* ho, int H, int L, int B,
                         const float* X, head_rec* r) {
    const int n = ho->out_dim, in = ho->in_dim;
    memcpy(r->x, X, sizeof(float) * (size_t)B * in);
    for (int t = 0; t < B; ++t)
        lin(blob + ho->lin_w, blob + ho->lin_b, X + (size_t)t * in, H, in, in, r->u + (size_t)t * H);
    for (int l = 0; l < L; ++l) {
        const int in_l = l == 0 ? H : n;
        const float* xs = l == 0 ? r->u : r->h[l - 1];
        const float *Wi = blob + ho->w_ih[l], *Wh = blob + ho->w_hh[l];
        const float *bi = blob + ho->b_ih[l], *bh = blob + ho->b_hh[l];
        for (int t = 0; t < B; ++t) {
            const float* x = xs + (size_t)t * in_l;
            const float* hp = t > 0 ? r->h[l] + (size_t)(t - 1) * n : NULL;
            const float* cp = t > 0 ? r->c[l] + (size_t)(t - 1) * n : NULL;
            float* g = r->gate[l] + (size_t)t * 4 * n;
            for (int j = 0; j < 4 * n; ++j) {
                double acc = (double)bi[j] + (double)bh[j];
                for (int k = 0; k < in_l; ++k) acc += (double)Wi[(size_t)j * in_l + k] * x[k];
                if (hp)
                    for (int k = 0; k < n; ++k) acc += (double)Wh[(size_t)j * n + k] * hp[k];
                const float z = (float)acc;
                g[j] = (j >= 2 * n && j < 3 * n) ? (float)tanh((double)z) : sigmoid_f(z);
            }
            for (int j = 0; j < n; ++j) {
                const float cv = g[n + j] * (cp ? cp[j] : 0.0f) + g[j] * g[2 * n + j];
                r->c[l][(size_t)t * n + j] = cv;
                r->h[l][(size_t)t * n + j] = g[3 * n + j] * (float)tanh((double)cv);
            }
        }
    }
}

/* dOut [B][n]: dL/d(output rows) (read only).  Accumulates the head's parameter gradients into
 * grad (blob order); dX [B][in] = dL/d(input rows), may be NULL. */
static void head_backward(const float* blob, float* grad, const mzs_lstm_head_offsets* ho, int H, int L,
                          int B, const head_rec* r, const float* dOut, float* dX) {
    const int n = ho->out_dim, in = ho->in_dim;
    const int wmax = H > n ? H : n;
    float* dh_up = (float*)malloc(sizeof(float) * (size_t)B * wmax);   /* dL/d(layer output rows) */
    float* dx_dn = (float*)malloc(sizeof(float) * (size_t)B * wmax);   /* dL/d(layer input rows)  */
    float* dz = (float*)malloc(sizeof(float) * (size_t)4 * n);
    float* dc_next = (float*)malloc(sizeof(float) * (size_t)n * 2);
    float* dhrec = dc_next + n;
    memcpy(dh_up, dOut, sizeof(float) * (size_t)B * n);
    for (int l = L - 1; l >= 0; --l) {
        const int in_l = l == 0 ? H : n;
        const float* xs = l == 0 ? r->u : r->h[l - 1];
        const float *Wi = blob + ho->w_ih[l], *Wh = blob + ho->w_hh[l];
        float *gWi = grad + ho->w_ih[l], *gWh = grad + ho->w_hh[l];
        float *gbi = grad + ho->b_ih[l], *gbh = grad + ho->b_hh[l];
        for (int j = 0; j < n; ++j) dc_next[j] = 0.0f, dhrec[j] = 0.0f;
        for (int t = B - 1; t >= 0; --t) {
            const float* g = r->gate[l] + (size_t)t * 4 * n;
            const float* c = r->c[l] + (size_t)t * n;
            const float* cp = t > 0 ? r->c[l] + (size_t)(t - 1) * n : NULL;
            const float* hp = t > 0 ? r->h[l] + (size_t)(t - 1) * n : NULL;
            const float* x = xs + (size_t)t * in_l;
            for (int j = 0; j < n; ++j) {
                const float dh = dh_up[(size_t)t * n + j] + dhrec[j];
                const float tc = (float)tanh((double)c[j]);
                const float ig = g[j], fg = g[n + j], gg = g[2 * n + j], og = g[3 * n + j];
                const float dc = dh * og * (1.0f - tc * tc) + dc_next[j];
                dz[j] = dc * gg * ig * (1.0f - ig);
                dz[n + j] = dc * (cp ? cp[j] : 0.0f) * fg * (1.0f - fg);
                dz[2 * n + j] = dc * ig * (1.0f - gg * gg);
                dz[3 * n + j] = dh * tc * og * (1.0f - og);
                dc_next[j] = dc * fg;
            }
            lin_back_weight(gWi, gbi, dz, x, 4 * n, in_l, in_l);
            for (int j = 0; j < 4 * n; ++j) gbh[j] += dz[j];
            if (hp) {
                for (int j = 0; j < 4 * n; ++j)
                    for (int k = 0; k < n; ++k) gWh[(size_t)j * n + k] += dz[j] * hp[k];
                lin_back_input(Wh, dz, 4 * n, n, n, dhrec);
            }
            lin_back_input(Wi, dz, 4 * n, in_l, in_l, dx_dn + (size_t)t * in_l);
        }
        memcpy(dh_up, dx_dn, sizeof(float) * (size_t)B * in_l);
    }
    /* dh_up now holds dL/du [B][H] */
    for (int t = 0; t < B; ++t) {
        lin_back_weight(grad + ho->lin_w, grad + ho->lin_b, dh_up + (size_t)t * H, r->x + (size_t)t * in, H, in,
                        in);
        if (dX) lin_back_input(blob + ho->lin_w, dh_up + (size_t)t * H, H, in, in, dX + (size_t)t * in);
    }
    free(dh_up), free(dx_dn), free(dz), free(dc_next);
}

static int lstm_train_grad(const mzs_dims* d, const float* blob, const mzo_train_batch* bt, float* grad,
                           float* loss_terms, double* total_loss, float* priority) {
    const int obs = d->obs_dim, A = d->n_actions, S = d->state_dim, H = d->hidden_dim,
              L = d->n_mid_layers, K = bt->K, B = bt->B;
    if (L < 1 || L > MZS_LSTM_MAX_LAYERS) return -1;
    const mzs_lstm_offsets o = mzs_lstm_layout(d);
    memset(grad, 0, sizeof(float) * o.total);
    const int n_terms = 3 * K - 1;
    head_rec* rp = (head_rec*)malloc(sizeof(head_rec) * (size_t)K * 4);
    head_rec *rv = rp + K, *rr = rp + 2 * K, *rs = rp + 3 * K; /* rr[k], rs[k]: dynamics call k (k < K-1) */
    float* h0 = (float*)malloc(sizeof(float) * (size_t)B * S);
    float* X = (float*)malloc(sizeof(float) * (size_t)B * (S + A));
    for (int k = 0; k < K; ++k) {
        rp[k] = head_rec_new(&o.pol, H, L, B);
        rv[k] = head_rec_new(&o.val, H, L, B);
        rr[k] = head_rec_new(&o.rew, H, L, B);
        rs[k] = head_rec_new(&o.st, H, L, B);
    }
    /* ---------------- forward (muzero_model.py:609-647) */
    for (int b = 0; b < B; ++b)
        lin(blob + o.r_w, blob + o.r_b, bt->obs + (size_t)b * obs, S, obs, obs, h0 + (size_t)b * S);
    const float* hk = h0;
    for (int k = 0; k < K; ++k) {
        head_forward(blob, &o.pol, H, L, B, hk, &rp[k]);
        head_forward(blob, &o.val, H, L, B, hk, &rv[k]);
        if (k + 1 < K) {
            for (int b = 0; b < B; ++b) {
                memcpy(X + (size_t)b * (S + A), hk + (size_t)b * S, sizeof(float) * (size_t)S);
                memcpy(X + (size_t)b * (S + A) + S, bt->actions + ((size_t)k * B + b) * A,
                       sizeof(float) * (size_t)A);
            }
            head_forward(blob, &o.rew, H, L, B, X, &rr[k]);
            head_forward(blob, &o.st, H, L, B, X, &rs[k]);
            hk = rs[k].h[L - 1];
        }
    }
    /* ---------------- losses (muzero_model.py:650-734) + backward through the unroll and, inside
     * every head, through the batch axis */
    float* dV = (float*)calloc((size_t)B * S * 3, sizeof(float));
    float *dR = dV + (size_t)B * S, *gh = dV + (size_t)2 * B * S;
    float* dP = (float*)calloc((size_t)B * A, sizeof(float));
    float* ghn = (float*)calloc((size_t)B * S, sizeof(float));           /* from dynamics call k      */
    float* gx = (float*)calloc((size_t)B * (S + A) * 2, sizeof(float));
    float* gx2 = gx + (size_t)B * (S + A);
    float* sm = (float*)malloc(sizeof(float) * (size_t)(S > A ? S : A) * 2);
    float* ptar = sm + (S > A ? S : A);
    double* sample_loss = (double*)calloc((size_t)B, sizeof(double));
    double wsum = 0.0;
    for (int i = 0; i < n_terms; ++i) loss_terms[i] = 0.0f;
    for (int b = 0; b < B; ++b) wsum += (bt->is_ratio ? bt->is_ratio[b] : 1.0f) * bt->inv_batch;
    for (int k = K - 1; k >= 0; --k) {
        const int t0 = k == 0 ? 0 : 3 * k - 1;
        const float* val = rv[k].h[L - 1];
        const float* pol = rp[k].h[L - 1];
        const float* rew = k > 0 ? rr[k - 1].h[L - 1] : NULL;
        memset(dV, 0, sizeof(float) * (size_t)B * S * 2);
        memset(dP, 0, sizeof(float) * (size_t)B * A);
        for (int b = 0; b < B; ++b) {
            const float w = (bt->is_ratio ? bt->is_ratio[b] : 1.0f) * bt->inv_batch;
            const float gs = (k > 0 ? 1.0f / (float)K : 1.0f) * w;
            int idx;
            float wt;
            support_target(bt->target_value[(size_t)k * B + b], S, &idx, &wt);
            for (int i = 0; i < S; ++i) ptar[i] = 0.0f;
            ptar[idx] = wt;
            const float lv = loss_term(g_loss_value, val + (size_t)b * S, S, ptar, gs, dV + (size_t)b * S, sm);
            const float lp = ce_loss(pol + (size_t)b * A, A, bt->target_policy + ((size_t)k * B + b) * A, gs,
                                     dP + (size_t)b * A, sm);
            loss_terms[t0] += lv * bt->inv_batch;
            loss_terms[t0 + 1] += lp * bt->inv_batch;
            sample_loss[b] += (double)lv + lp;
            priority[(size_t)k * B + b] =
                fabsf(mzo_support_to_scalar(val + (size_t)b * S, S) - bt->target_value[(size_t)k * B + b]);
            if (k > 0) {
                support_target(bt->target_reward[(size_t)k * B + b], S, &idx, &wt);
                for (int i = 0; i < S; ++i) ptar[i] = 0.0f;
                ptar[idx] = wt;
                const float lr = loss_term(g_loss_reward, rew + (size_t)b * S, S, ptar, gs, dR + (size_t)b * S, sm);
                loss_terms[t0 + 2] += lr * bt->inv_batch;
                sample_loss[b] += lr;
            }
        }
        head_backward(blob, grad, &o.val, H, L, B, &rv[k], dV, gx);
        head_backward(blob, grad, &o.pol, H, L, B, &rp[k], dP, gx2);
        for (size_t i = 0; i < (size_t)B * S; ++i) gh[i] = gx[i] + gx2[i] + ghn[i];
        if (k > 0) {
            for (size_t i = 0; i < (size_t)B * S; ++i) gh[i] *= 0.5f; /* register_hook, :640 */
            head_backward(blob, grad, &o.st, H, L, B, &rs[k - 1], gh, gx);
            head_backward(blob, grad, &o.rew, H, L, B, &rr[k - 1], dR, gx2);
            for (int b = 0; b < B; ++b)
                for (int i = 0; i < S; ++i)
                    ghn[(size_t)b * S + i] = gx[(size_t)b * (S + A) + i] + gx2[(size_t)b * (S + A) + i];
        } else {
            for (int b = 0; b < B; ++b)
                lin_back_weight(grad + o.r_w, grad + o.r_b, gh + (size_t)b * S, bt->obs + (size_t)b * obs, S, obs,
                                obs);
        }
    }
    double total = 0.0;
    for (int b = 0; b < B; ++b)
        total += (double)((bt->is_ratio ? bt->is_ratio[b] : 1.0f) * bt->inv_batch) * sample_loss[b];
    /* l2 (neural_network_mlp_model.py:280-285): the zip over the three parameter lists stops after the
     * representation's two tensors, so only (state_norm.weight, reward.0.weight, policy.0.weight)
     * and (state_norm.bias, reward.0.bias, policy.0.bias) are regularised — no LSTM tensor is */
    {
        const size_t segs[][2] = {{o.r_w, (size_t)S * obs},          {o.r_b, (size_t)S},
                                  {o.rew.lin_w, (size_t)H * (S + A)}, {o.rew.lin_b, (size_t)H},
                                  {o.pol.lin_w, (size_t)H * S},       {o.pol.lin_b, (size_t)H}};
        double sq = 0.0;
        for (int s = 0; s < 6; ++s)
            for (size_t i = 0; i < segs[s][1]; ++i) {
                const float wv = blob[segs[s][0] + i];
                sq += (double)wv * wv;
                grad[segs[s][0] + i] += (float)(2.0 * 1e-4 * wsum) * wv;
            }
        total += wsum * 1e-4 * sq;
    }
    *total_loss = total;
    for (int k = 0; k < K; ++k) head_rec_free(&rp[k]), head_rec_free(&rv[k]), head_rec_free(&rr[k]), head_rec_free(&rs[k]);
    free(rp), free(h0), free(X), free(dV), free(dP), free(ghn), free(gx), free(sm), free(sample_loss);
    return 0;
}

int mzo_train_grad(const mzs_dims* d, const float* blob, const mzo_train_batch* bt, float* grad,
                   float* loss_terms, double* total_loss, float* priority) {
    const int obs = d->obs_dim, A = d->n_actions, S = d->state_dim, H = d->hidden_dim,
              L = d->n_mid_layers, K = bt->K, B = bt->B;
    if (K < 1 || B < 1) return -1;
    if (d->model_kind == MZS_MODEL_LSTM) return lstm_train_grad(d, blob, bt, grad, loss_terms, total_loss, priority);
    const mzs_blob_offsets o = mzs_blob_layout(d);
    memset(grad, 0, sizeof(float) * o.total);
    const int n_terms = 3 * K - 1;
    for (int i = 0; i < n_terms; ++i) loss_terms[i] = 0.0f;
    net R = {obs, obs, H, L, blob + o.r_in_w, blob + o.r_in_b, blob + o.r_mid_w, blob + o.r_mid_b,
             grad + o.r_in_w, grad + o.r_in_b, grad + o.r_mid_w, grad + o.r_mid_b};
    net D = {S + A, S + A, H, L, blob + o.d_in_w, blob + o.d_in_b, blob + o.d_mid_w, blob + o.d_mid_b,
             grad + o.d_in_w, grad + o.d_in_b, grad + o.d_mid_w, grad + o.d_mid_b};
    net P = {S, S, H, L, blob + o.p_in_w, blob + o.p_in_b, blob + o.p_mid_w, blob + o.p_mid_b,
             grad + o.p_in_w, grad + o.p_in_b, grad + o.p_mid_w, grad + o.p_mid_b};
    const size_t TH = (size_t)(L + 1) * H;
    float* h = (float*)malloc(sizeof(float) * (size_t)K * S);          /* hidden states h_0..h_{K-1} */
    float* zs = (float*)malloc(sizeof(float) * (size_t)K * S);         /* pre-scale rows             */
    scale_rec* sr = (scale_rec*)malloc(sizeof(scale_rec) * (size_t)K);
    float* TR = (float*)malloc(sizeof(float) * TH);
    float* TP = (float*)malloc(sizeof(float) * TH * K);
    float* TD = (float*)malloc(sizeof(float) * TH * K);
    float* xD = (float*)malloc(sizeof(float) * (size_t)K * (S + A));
    float* pol = (float*)malloc(sizeof(float) * (size_t)K * A);
    float* val = (float*)malloc(sizeof(float) * (size_t)K * S);
    float* rew = (float*)malloc(sizeof(float) * (size_t)K * S);        /* rew[k] = reward logits of step k (k >= 1) */
    float* gT = (float*)malloc(sizeof(float) * (size_t)H * 3);
    float *tmp = gT + H, *gT2 = gT + 2 * H;
    int nmax = S + A > H ? S + A : H;
    float* gz = (float*)malloc(sizeof(float) * (size_t)nmax * 6);
    float *gz2 = gz + nmax, *gh = gz + 2 * nmax, *ghn = gz + 3 * nmax, *sm = gz + 4 * nmax,
          *ptar = gz + 5 * nmax;
    double total = 0.0, wsum = 0.0;

    for (int b = 0; b < B; ++b) {
        const float w = (bt->is_ratio ? bt->is_ratio[b] : 1.0f) * bt->inv_batch;
        wsum += w;
        const float* ob = bt->obs + (size_t)b * obs;
        /* ---------------- forward (muzero_model.py:609-647) */
        net_forward(&R, ob, TR);
        lin(blob + o.r_out_w, blob + o.r_out_b, TR + (size_t)L * H, S, H, H, zs);
        scale_forward(zs, S, h, &sr[0]);
        for (int k = 0; k < K; ++k) {
            float* tp = TP + TH * k;
            net_forward(&P, h + (size_t)k * S, tp);
            lin(blob + o.p_pol_w, blob + o.p_pol_b, tp + (size_t)L * H, A, H, H, pol + (size_t)k * A);
            lin(blob + o.p_val_w, blob + o.p_val_b, tp + (size_t)L * H, S, H, H, val + (size_t)k * S);
            if (k + 1 < K) {
                float* x = xD + (size_t)k * (S + A);
                memcpy(x, h + (size_t)k * S, sizeof(float) * (size_t)S);
                memcpy(x + S, bt->actions + ((size_t)k * B + b) * A, sizeof(float) * (size_t)A);
                float* td = TD + TH * k;
                net_forward(&D, x, td);
                lin(blob + o.d_rew_w, blob + o.d_rew_b, td + (size_t)L * H, S, H, H,
                    rew + (size_t)(k + 1) * S);
                lin(blob + o.d_st_w, blob + o.d_st_b, td + (size_t)L * H, S, H, H,
                    zs + (size_t)(k + 1) * S);
                scale_forward(zs + (size_t)(k + 1) * S, S, h + (size_t)(k + 1) * S, &sr[k + 1]);
            }
        }
        /* ---------------- losses + backward (muzero_model.py:650-734) */
        double sample_loss = 0.0;
        for (int i = 0; i < S; ++i) ghn[i] = 0.0f; /* dL/dh_{k} arriving from dynamics step k */
        for (int k = K - 1; k >= 0; --k) {
            const float gs = (k > 0 ? 1.0f / (float)K : 1.0f) * w;
            const int t0 = k == 0 ? 0 : 3 * k - 1; /* term order: v0 p0 | v1 p1 r1 | ... */
            const float* tp = TP + TH * k;
            /* value */
            int idx;
            float wt;
            support_target(bt->target_value[(size_t)k * B + b], S, &idx, &wt);
            for (int i = 0; i < S; ++i) ptar[i] = 0.0f;
            ptar[idx] = wt;
            for (int i = 0; i < S; ++i) gz[i] = 0.0f;
            float lv = loss_term(g_loss_value, val + (size_t)k * S, S, ptar, gs, gz, sm);
            /* policy */
            for (int i = 0; i < A; ++i) gz2[i] = 0.0f;
            float lp = ce_loss(pol + (size_t)k * A, A, bt->target_policy + ((size_t)k * B + b) * A, gs,
                               gz2, sm);
            loss_terms[t0] += lv * bt->inv_batch;
            loss_terms[t0 + 1] += lp * bt->inv_batch;
            sample_loss += (double)lv + lp;
            priority[(size_t)k * B + b] =
                fabsf(mzo_support_to_scalar(val + (size_t)k * S, S) - bt->target_value[(size_t)k * B + b]);
            /* prediction heads -> trunk */
            lin_back_weight(grad + o.p_val_w, grad + o.p_val_b, gz, tp + (size_t)L * H, S, H, H);
            lin_back_weight(grad + o.p_pol_w, grad + o.p_pol_b, gz2, tp + (size_t)L * H, A, H, H);
            lin_back_input(blob + o.p_val_w, gz, S, H, H, gT);
            lin_back_input(blob + o.p_pol_w, gz2, A, H, H, gT2);
            for (int j = 0; j < H; ++j) gT[j] += gT2[j];
            net_backward(&P, h + (size_t)k * S, tp, gT, gh, tmp);
            for (int i = 0; i < S; ++i) gh[i] += ghn[i];
            if (k > 0) {
                for (int i = 0; i < S; ++i) gh[i] *= 0.5f; /* register_hook, muzero_model.py:640 */
                scale_backward(gh, h + (size_t)k * S, S, &sr[k], gz);
                /* reward of step k (from dynamics step k-1) */
                support_target(bt->target_reward[(size_t)k * B + b], S, &idx, &wt);
                for (int i = 0; i < S; ++i) ptar[i] = 0.0f;
                ptar[idx] = wt;
                for (int i = 0; i < S; ++i) gz2[i] = 0.0f;
                float lr = loss_term(g_loss_reward, rew + (size_t)k * S, S, ptar, gs, gz2, sm);
                loss_terms[t0 + 2] += lr * bt->inv_batch;
                sample_loss += lr;
                const float* td = TD + TH * (k - 1);
                lin_back_weight(grad + o.d_st_w, grad + o.d_st_b, gz, td + (size_t)L * H, S, H, H);
                lin_back_weight(grad + o.d_rew_w, grad + o.d_rew_b, gz2, td + (size_t)L * H, S, H, H);
                lin_back_input(blob + o.d_st_w, gz, S, H, H, gT);
                lin_back_input(blob + o.d_rew_w, gz2, S, H, H, gT2);
                for (int j = 0; j < H; ++j) gT[j] += gT2[j];
                net_backward(&D, xD + (size_t)(k - 1) * (S + A), td, gT, gz, tmp);
                for (int i = 0; i < S; ++i) ghn[i] = gz[i]; /* the action columns get no gradient use */
            } else {
                scale_backward(gh, h, S, &sr[0], gz);
                lin_back_weight(grad + o.r_out_w, grad + o.r_out_b, gz, TR + (size_t)L * H, S, H, H);
                lin_back_input(blob + o.r_out_w, gz, S, H, H, gT);
                net_backward(&R, ob, TR, gT, NULL, tmp);
            }
        }
        total += (double)w * sample_loss;
    }
    /* l2 (neural_network_mlp_model.py:280-285): zip over the three parameter lists stops at the
     * shortest (the representation's), so the dynamics' next-state head and the prediction's value
     * head are left out */
    {
        const size_t segs[][2] = {
            {o.r_in_w, (size_t)H * obs}, {o.r_in_b, (size_t)H}, {o.r_out_w, (size_t)S * H}, {o.r_out_b, (size_t)S},
            {o.d_in_w, (size_t)H * (S + A)}, {o.d_in_b, (size_t)H}, {o.d_rew_w, (size_t)S * H}, {o.d_rew_b, (size_t)S},
            {o.p_in_w, (size_t)H * S}, {o.p_in_b, (size_t)H}, {o.p_pol_w, (size_t)A * H}, {o.p_pol_b, (size_t)A},
            {o.r_mid_w, (size_t)H * H}, {o.r_mid_b, (size_t)H}, {o.d_mid_w, (size_t)H * H}, {o.d_mid_b, (size_t)H},
            {o.p_mid_w, (size_t)H * H}, {o.p_mid_b, (size_t)H}};
        const int nseg = L > 0 ? 18 : 12;
        double sq = 0.0;
        for (int s = 0; s < nseg; ++s)
            for (size_t i = 0; i < segs[s][1]; ++i) {
                const float wv = blob[segs[s][0] + i];
                sq += (double)wv * wv;
                grad[segs[s][0] + i] += (float)(2.0 * 1e-4 * wsum) * wv;
            }
        total += wsum * 1e-4 * sq;
    }
    *total_loss = total;
    free(h), free(zs), free(sr), free(TR), free(TP), free(TD), free(xD), free(pol), free(val), free(rew);
    free(gT), free(gz);
    return 0;
}

/* torch.optim.Adam (betas 0.9 / 0.999, eps 1e-8, no weight decay) — step t >= 1 */
/* Adam's beta1 / SGD's momentum for the following steps: 0.9 unless OneCycleLR cycles it
 * (torch.optim.lr_scheduler.OneCycleLR, cycle_momentum: muzero_model.py:433); process-wide state */
static double g_momentum = 0.9;
void mzo_set_momentum(double momentum) { g_momentum = momentum; }

void mzo_adam_step(float* w, float* m, float* v, const float* g, size_t n, int step, double lr) {
    const double b1 = g_momentum, b2 = 0.999, eps = 1e-8;
    const double bc1 = 1.0 - pow(b1, step), bc2 = 1.0 - pow(b2, step);
    const double step_size = lr / bc1, bc2s = sqrt(bc2);
    for (size_t i = 0; i < n; ++i) {
        m[i] = m[i] + (g[i] - m[i]) * (float)(1.0 - b1);
        v[i] = v[i] * (float)b2 + (float)(1.0 - b2) * g[i] * g[i];
        const float denom = (float)(sqrtf(v[i]) / bc2s) + (float)eps;
        w[i] = w[i] - (float)step_size * (m[i] / denom);
    }
}

/* torch.optim.SGD(momentum = 0.9): buf = g on the first step */
void mzo_sgd_step(float* w, float* buf, const float* g, size_t n, int step, double lr) {
    for (size_t i = 0; i < n; ++i) {
        buf[i] = step == 1 ? g[i] : (float)g_momentum * buf[i] + g[i];
        w[i] = w[i] - (float)lr * buf[i];
    }
}
