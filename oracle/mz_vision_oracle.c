/*
 * mz_vision_oracle.c — CPU oracle of the 'vision_model' inference functions (test infrastructure,
 * see mz_oracle.h).  Restates, in eval mode (BatchNorm2d on its running statistics, Dropout2d off):
 *   Residual_block.forward               neural_network_vision_model.py:41-82
 *   Down_sample.forward                  :84-122
 *   Representation_function.forward      :125-163  (down_sampling=True, the constructor default)
 *   Dynamics_function.forward            :167-230  + Muzero.one_hot_encode's action plane,
 *                                                    muzero_model.py:505-516
 *   Prediction_function.forward          :234-300
 *   inference wrappers                   muzero_model.py:828-905 (softmax policy, support -> scalar)
 * Arithmetic contract (what the HIP kernels reproduce bit for bit):
 *   3x3 convolution   one fp32 fma chain per output, taps in (input channel, ky, kx) order,
 *                     taps that fall into the zero padding skipped
 *   batchnorm + relu  max(0, fma(x - mean, gamma / sqrt(var + 1e-5), beta)), correctly rounded / and sqrt
 *   avgpool 3/2/1     taps added in (ky, kx) order, padding skipped, sum / 9 (count_include_pad)
 *   1x1 convolution   chain from the bias over the input channels
 *   Linear            contract "dot" of include/mzs_numeric.h (even / odd fma chains), relu
 *   softmax, support  the functions of mz_oracle.c (contract exp / sum / quotient)
 * Pinned by tests/golden/model_L2_vis*.npz (made from the reference by make_vision_fixtures.py).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "mz_oracle.h"
#include "../include/mzs_numeric.h"

#define VC MZS_VIS_C

/* out [Co][So][So] = conv3x3(in [Ci][Si][Si]), stride s, padding 1, no bias */
static void conv3(const float* w, const float* in, int Ci, int Si, int Co, int s, float* out) {
    const int So = (Si + 2 - 3) / s + 1;
    for (int co = 0; co < Co; ++co)
        for (int oy = 0; oy < So; ++oy)
            for (int ox = 0; ox < So; ++ox) {
                float acc = 0.0f;
                for (int ci = 0; ci < Ci; ++ci)
                    for (int ky = 0; ky < 3; ++ky) {
                        const int iy = oy * s + ky - 1;
                        if (iy < 0 || iy >= Si) continue;
                        for (int kx = 0; kx < 3; ++kx) {
                            const int ix = ox * s + kx - 1;
                            if (ix < 0 || ix >= Si) continue;
                            acc = mzs_fma(w[((co * Ci + ci) * 3 + ky) * 3 + kx], in[(ci * Si + iy) * Si + ix], acc);
                        }
                    }
                out[(co * So + oy) * So + ox] = acc;
            }
}

/* relu(batchnorm(x)) in eval mode, [C][n] in place or to out */
static void bn_relu(const float* blob, const mzs_vis_bn* bn, const float* in, int C, int n, float* out) {
    for (int c = 0; c < C; ++c) {
        const float inv = blob[bn->w + c] / sqrtf(blob[bn->var + c] + 1e-5f);
        const float mean = blob[bn->mean + c], beta = blob[bn->b + c];
        for (int i = 0; i < n; ++i) {
            const float y = mzs_fma(in[c * n + i] - mean, inv, beta);
            out[c * n + i] = y > 0.0f ? y : 0.0f;
        }
    }
}

/* x += conv_a(R(conv_b(R(conv_a(R(x)))))), R = relu(batchnorm(.));  x [C][S][S]; t1, t2 scratch */
static void res_block(const float* blob, const mzs_vis_res* r, float* x, int S, float* t1, float* t2) {
    const int C = r->C, n = S * S;
    bn_relu(blob, &r->bn, x, C, n, t1);
    conv3(blob + r->conv_a, t1, C, S, C, 1, t2);
    bn_relu(blob, &r->bn, t2, C, n, t1);
    conv3(blob + r->conv_b, t1, C, S, C, 1, t2);
    bn_relu(blob, &r->bn, t2, C, n, t1);
    conv3(blob + r->conv_a, t1, C, S, C, 1, t2);
    for (int i = 0; i < C * n; ++i) x[i] = t2[i] + x[i];
}

/* AvgPool2d(kernel 3, stride 2, padding 1), count_include_pad */
static void avgpool(const float* in, int C, int Si, float* out) {
    const int So = (Si + 2 - 3) / 2 + 1;
    for (int c = 0; c < C; ++c)
        for (int oy = 0; oy < So; ++oy)
            for (int ox = 0; ox < So; ++ox) {
                float acc = 0.0f;
                for (int ky = 0; ky < 3; ++ky) {
                    const int iy = oy * 2 + ky - 1;
                    if (iy < 0 || iy >= Si) continue;
                    for (int kx = 0; kx < 3; ++kx) {
                        const int ix = ox * 2 + kx - 1;
                        if (ix < 0 || ix >= Si) continue;
                        acc = acc + in[(c * Si + iy) * Si + ix];
                    }
                }
                out[(c * So + oy) * So + ox] = acc / 9.0f;
            }
}

/* contract "dot": even / odd fma chains from the bias */
static void linear(const float* W, const float* b, const float* x, int nout, int nin, int relu, float* y) {
    for (int j = 0; j < nout; ++j) {
        float e = b[j], o = 0.0f;
        const float* w = W + (size_t)j * nin;
        for (int k = 0; k < nin; k += 2) e = mzs_fma(w[k], x[k], e);
        for (int k = 1; k < nin; k += 2) o = mzs_fma(w[k], x[k], o);
        const float v = e + o;
        y[j] = relu && !(v > 0.0f) ? 0.0f : v;
    }
}

/* 1x1 conv (+bias) over [Ci][n] -> flatten -> MLP; logits [n_out] */
static void head_mlp(const mzs_dims* d, const float* blob, size_t conv_w, size_t conv_b, int Ci, const mzs_vis_mlp* m,
                     const float* x, float* logits, float* t1, float* t2) {
    const int n = MZS_VIS_S4 * MZS_VIS_S4, H = d->hidden_dim;
    for (int co = 0; co < VC; ++co)
        for (int i = 0; i < n; ++i) {
            float acc = blob[conv_b + co];
            for (int ci = 0; ci < Ci; ++ci) acc = mzs_fma(blob[conv_w + co * Ci + ci], x[ci * n + i], acc);
            t1[co * n + i] = acc;
        }
    linear(blob + m->in_w, blob + m->in_b, t1, H, MZS_VIS_HIDDEN, 1, t2);
    for (int l = 0; l < d->n_mid_layers; ++l) {
        linear(blob + m->mid_w, blob + m->mid_b, t2, H, H, 1, t1);
        memcpy(t2, t1, sizeof(float) * (size_t)H);
    }
    linear(blob + m->out_w, blob + m->out_b, t2, m->n_out, H, 0, logits);
}

static float* scratch3(const mzs_dims* d) {
    int n = MZS_VIS_S1 * MZS_VIS_S1 * VC;
    n = n > d->hidden_dim ? n : d->hidden_dim;
    n = n > 4 * MZS_VIS_S4 * MZS_VIS_S4 ? n : 4 * MZS_VIS_S4 * MZS_VIS_S4;
    return (float*)malloc(sizeof(float) * (size_t)n * 3);
}

void mzo_vision_representation(const mzs_dims* d, const float* blob, const float* obs, int32_t B, float* hidden,
                               float* stages) {
    const mzs_vision_offsets o = mzs_vision_layout(d);
    const int n = MZS_VIS_S1 * MZS_VIS_S1 * VC;
    float* x = scratch3(d);
    float *t1 = x + (d->hidden_dim > n ? d->hidden_dim : n), *t2 = t1 + (d->hidden_dim > n ? d->hidden_dim : n);
    for (int b = 0; b < B; ++b) {
        const float* img = obs + (size_t)b * VC * MZS_VIS_SIDE * MZS_VIS_SIDE;
        conv3(blob + o.r_conv_in, img, VC, MZS_VIS_SIDE, VC / 2, 2, x); /* 1 x 49 x 49 */
        res_block(blob, &o.r_res_in, x, MZS_VIS_S1, t1, t2);
        res_block(blob, &o.r_res_in, x, MZS_VIS_S1, t1, t2);
        conv3(blob + o.r_conv_out, x, VC / 2, MZS_VIS_S1, VC, 2, t1); /* 3 x 25 x 25 */
        memcpy(x, t1, sizeof(float) * VC * MZS_VIS_S2 * MZS_VIS_S2);
        res_block(blob, &o.r_res_out, x, MZS_VIS_S2, t1, t2);
        res_block(blob, &o.r_res_out, x, MZS_VIS_S2, t1, t2);
        avgpool(x, VC, MZS_VIS_S2, t1); /* 3 x 13 x 13 */
        memcpy(x, t1, sizeof(float) * VC * MZS_VIS_S3 * MZS_VIS_S3);
        for (int i = 0; i < 3; ++i) res_block(blob, &o.r_res_out, x, MZS_VIS_S3, t1, t2);
        avgpool(x, VC, MZS_VIS_S3, t1); /* 3 x 7 x 7 */
        if (stages && b == 0) memcpy(stages, t1, sizeof(float) * MZS_VIS_HIDDEN); /* Down_sample's output */
        memcpy(x, t1, sizeof(float) * MZS_VIS_HIDDEN);
        res_block(blob, &o.r_res, x, MZS_VIS_S4, t1, t2);
        memcpy(hidden + (size_t)b * MZS_VIS_HIDDEN, x, sizeof(float) * MZS_VIS_HIDDEN);
    }
    free(x);
}

void mzo_vision_dynamics(const mzs_dims* d, const float* blob, const float* hidden, const int32_t* action, int32_t B,
                         float* reward, float* next_hidden, float* reward_logits) {
    const mzs_vision_offsets o = mzs_vision_layout(d);
    const int n = MZS_VIS_S4 * MZS_VIS_S4, Sd = d->state_dim;
    float* x = scratch3(d);
    const int w = MZS_VIS_S1 * MZS_VIS_S1 * VC > d->hidden_dim ? MZS_VIS_S1 * MZS_VIS_S1 * VC : d->hidden_dim;
    float *t1 = x + w, *t2 = t1 + w;
    float* logits = (float*)malloc(sizeof(float) * (size_t)Sd);
    float xin[4 * MZS_VIS_S4 * MZS_VIS_S4];
    for (int b = 0; b < B; ++b) {
        /* cat(state, action plane): plane value (a + 1) / A (muzero_model.py:515) */
        memcpy(xin, hidden + (size_t)b * MZS_VIS_HIDDEN, sizeof(float) * MZS_VIS_HIDDEN);
        const float pv = (float)(action[b] + 1) / (float)d->n_actions;
        for (int i = 0; i < n; ++i) xin[VC * n + i] = pv;
        conv3(blob + o.d_conv, xin, VC + 1, MZS_VIS_S4, VC, 1, t1);
        bn_relu(blob, &o.d_bn, t1, VC, n, x);
        for (int l = 0; l < d->n_mid_layers; ++l) res_block(blob, &o.d_res, x, MZS_VIS_S4, t1, t2);
        for (int i = 0; i < MZS_VIS_HIDDEN; ++i) next_hidden[(size_t)b * MZS_VIS_HIDDEN + i] = x[i] > 0.0f ? x[i] : 0.0f;
        head_mlp(d, blob, o.d_conv_rew, o.d_conv_rew_b, VC + 1, &o.d_fc, xin, logits, t1, t2);
        if (reward_logits) memcpy(reward_logits + (size_t)b * Sd, logits, sizeof(float) * (size_t)Sd);
        reward[b] = mzo_support_to_scalar(logits, Sd);
    }
    free(x), free(logits);
}

/* softmax over n logits (contract exp, adjacent-pair tree sum, corrected quotient): mz_oracle.c */
void mzo_softmax(const float* q, int32_t n, float* p);

void mzo_vision_prediction(const mzs_dims* d, const float* blob, const float* hidden, int32_t B, float* policy,
                           float* value) {
    const mzs_vision_offsets o = mzs_vision_layout(d);
    const int Sd = d->state_dim, A = d->n_actions;
    float* x = scratch3(d);
    const int w = MZS_VIS_S1 * MZS_VIS_S1 * VC > d->hidden_dim ? MZS_VIS_S1 * MZS_VIS_S1 * VC : d->hidden_dim;
    float *t1 = x + w, *t2 = t1 + w;
    float* logits = (float*)malloc(sizeof(float) * (size_t)(Sd > A ? Sd : A));
    for (int b = 0; b < B; ++b) {
        memcpy(x, hidden + (size_t)b * MZS_VIS_HIDDEN, sizeof(float) * MZS_VIS_HIDDEN);
        for (int l = 0; l < d->n_mid_layers; ++l) res_block(blob, &o.p_res, x, MZS_VIS_S4, t1, t2);
        head_mlp(d, blob, o.p_conv_val, o.p_conv_val_b, VC, &o.p_v, x, logits, t1, t2);
        value[b] = mzo_support_to_scalar(logits, Sd);
        head_mlp(d, blob, o.p_conv_pol, o.p_conv_pol_b, VC, &o.p_p, x, logits, t1, t2);
        mzo_softmax(logits, A, policy + (size_t)b * A);
    }
    free(x), free(logits);
}
