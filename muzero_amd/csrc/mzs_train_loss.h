/*
 * mzs_train_loss.h — the loss criteria of evaluate_loss (muzero_model.py:650-713) as wave-level device
 * functions: one wavefront holds one logit vector (n <= 64), lane = bin.  Shared by the MLP / LSTM training
 * kernels (mzs_train.h, mzs_train_lstm.h) and the vision model's (mzs_vision_train.hip).
 */
#ifndef MZS_TRAIN_LOSS_H
#define MZS_TRAIN_LOSS_H

#include "mzs_device.h"

namespace mzs {

/* order-free sum over the wave of one value per lane */
__device__ __forceinline__ float train_sum(float v) { return wave_tree_sum(v); }

/* transform_with_support (muzero_model.py:536-563): the one (index, weight) pair it scatters */
__device__ __forceinline__ void train_support_target(float x, int S, int* idx, float* weight) {
    const int sd = S / 2, rem = S & 1;
    const float ax = fabsf(x);
    const float sg0 = x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f);
    float t = sg0 * (__builtin_sqrtf(ax + 1.0f) - 1.0f) + 0.001f * x;
    t = fmaxf(t, (float)-sd);
    t = fminf(t, (float)(sd + rem));
    const float sg = t > 0.0f ? 1.0f : (t < 0.0f ? -1.0f : 0.0f);
    int i = (int)(floorf(t) + (float)sd + 1.0f);
    i = i < -(S - 1) ? -(S - 1) : (i > S - 1 ? S - 1 : i);
    *idx = i;
    *weight = (ceilf(t) - t) * sg;
}

/* One loss term of evaluate_loss (muzero_model.py:650-700): criterion `mode` (MZS_LOSS_*, the
 * Loss_function methods of neural_network_mlp_model.py:236-262) of q = softmax(z) + 1e-9 against the
 * target p (this lane's entry); delta = upstream * dloss/dz to dz[0..n).  With c = dloss/dq:
 *   CE  -sum p log q, c = -p / q          KL  sum p' (log p' - log q), p' = p + 1e-9, c = -p' / q
 *   MSE mean (p - q)^2, c = -2 (p - q) / n          ZERO 0, c = 0
 * and dloss/dz = s (c - sum c s).  Returns the loss (uniform); *softmax_out = this lane's softmax
 * probability. */
/* criterion of the policy term: always the cross-entropy, against p + 1e-9 when the loss type is
 * 'general_kkc' (label = ["zero_clamp_transform"], muzero_model.py:373-375) — the one loss type whose
 * value criterion is the KL divergence, which is how the kernels recognise it */
constexpr int kLossCeClamped = 100;
__device__ __forceinline__ int train_policy_mode(int loss_value) {
    return loss_value == MZS_LOSS_KL ? kLossCeClamped : MZS_LOSS_CE;
}

__device__ __forceinline__ float train_ce(const float* z, int n, float p, float upstream, float* dz,
                                          float* softmax_out, int lane, int mode = MZS_LOSS_CE) {
    const bool valid = lane < n;
    const float zv = valid ? z[lane] : 0.0f;
    const float m = wave_max(valid ? zv : -__builtin_inff());
    const float e = valid ? expf(zv - m) : 0.0f;
    const float Z = train_sum(e);
    const float s = e / Z;
    const float q = s + 1e-9f;
    const float pv = valid ? p : 0.0f;
    float c, term;
    if (mode == MZS_LOSS_CE || mode == kLossCeClamped) {
        const float pc = mode == kLossCeClamped ? pv + 1e-9f : pv;
        c = -pc / q;
        term = -pc * logf(q);
    } else if (mode == MZS_LOSS_KL) {
        const float pp = pv + 1e-9f;
        c = -pp / q;
        term = pp * (logf(pp) - logf(q));
    } else if (mode == MZS_LOSS_MSE) {
        const float d = pv - q;
        c = -2.0f * d / (float)n;
        term = d * d / (float)n;
    } else {
        c = 0.0f;
        term = 0.0f;
    }
    const float loss = train_sum(valid ? term : 0.0f);
    const float dot = train_sum(valid ? c * s : 0.0f);
    if (valid) dz[lane] = upstream * s * (c - dot);
    *softmax_out = s;
    wave_sync();
    return loss;
}

}  // namespace mzs
#endif
