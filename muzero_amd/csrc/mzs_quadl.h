/*
 * mzs_quadl.h — the 'lstm_model' (neural_network_lstm_model.py) on the four-games-per-wavefront
 * search: lane roles, tree code and arithmetic of mzs_quad.h, weights streamed from the device
 * store as in mzs_quadw.h (the four heads do not share trunks, so a simulation touches 892 weight
 * rows: more than the register file holds).
 *
 * One head = Linear (K inputs -> H) and one LSTM layer stepped from zero state: the i, g, o gate
 * projections are three 64-neuron output groups of one fused layer (columns 0, 64, 128 of
 * LstmHead::gate[0]); a lane holds the same four units of each gate, so the cell step
 * h = sigmoid(o) * tanh(sigmoid(i) * tanh(g)) is lane-local.  Every unit of the weight stream is a
 * single chunk (fan-in <= 64); four units per head, so every head starts in stream phase 0.
 * Compile-time S_dim (<= 32... 64), H (<= 64) and A; one LSTM layer (number_of_hidden_layer = 1).
 *
 * Reference code restated (file:line in /root/reference): neural_network_lstm_model.py:10-80,
 * muzero_model.py:828-905 (as mzs_device.h lstm_head / model_*).
 */
#ifndef MZS_QUADL_H
#define MZS_QUADL_H

#include "mzs_quadw.h"

namespace mzs {

/* the small vectors of one head in the simulation loop (the block's LDS copy: WideTabs) */
struct LstmHeadTabs {
    const float *lin_b, *gate_b, *bhh; /* Linear bias [64], bias_ih [192], bias_hh [192] */
    const float* extra;                /* [A][64]: the action rows of the Linear (dynamics heads), else null */
};
struct LstmTabs {
    LstmHeadTabs rew, st, pol, val;
};
constexpr int kLstmHeadTabFloats = 64 + 192 + 192; /* + A * 64 for the two dynamics heads */

/* mzs_sigmoid / mzs_tanh / mzs_lstm_cell0 (include/mzs_numeric.h) on four values at once: the same
 * operations per element, on vectors (mzs_quad.h expf4) */
__device__ __forceinline__ f32x4 sigmoid4(const f32x4 x) { return splat4(1.0f) / (splat4(1.0f) + expf4(-x)); }
__device__ __forceinline__ f32x4 tanh4(const f32x4 x) {
    const f32x4 a = __builtin_elementwise_abs(x);
    const f32x4 t = expm1_neg4(splat4(-2.0f) * a);
    const f32x4 y = -t / (t + splat4(2.0f));
    return f32x4{x[0] < 0.0f ? -y[0] : y[0], x[1] < 0.0f ? -y[1] : y[1], x[2] < 0.0f ? -y[2] : y[2],
                 x[3] < 0.0f ? -y[3] : y[3]};
}
__device__ __forceinline__ f32x4 lstm_cell0_4(const f32x4 gi, const f32x4 gg, const f32x4 go) {
    const f32x4 c = sigmoid4(gi) * tanh4(gg);
    return sigmoid4(go) * tanh4(c);
}

/* one head on the input row xrow (LDS, KIN values): h of the LSTM layer, units n0..n0+3 of this
 * lane's game.  Four single-chunk units of the weight stream (Linear, then the i, g, o gate
 * projections): phases 0..3, so every head starts in phase 0; the o unit's request is the next
 * head's Linear (KN rows).  trow: the game's 64-float row for the Linear's output (the B operand
 * of the gates). */
template <int KIN, int H, int KN, class ST>
__device__ __forceinline__ f32x4 lstm_head_quad(ST& st, const LstmHeadTabs& ht, const float* xrow, int action,
                                                float* trow, const QuadLane& q) {
    static_assert(H <= 64 && KIN <= 64, "single-chunk units");
    const auto ld4 = [&](const float* p) {
        const float4 v = *reinterpret_cast<const float4*>(p + q.n0);
        return f32x4{v.x, v.y, v.z, v.w};
    };
    const f32x4 bi = ld4(ht.bhh), bg = ld4(ht.bhh + 64), bo = ld4(ht.bhh + 128);
    f32x4 c = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    if (action >= 0) c = ld4(ht.extra + action * 64); /* one-hot action column of the dynamics heads */
    f32x4 lin = wide_unit<KIN, 0>(st, ht.lin_b + q.n0, xrow, q);
    if (action >= 0) lin = lin + c;
    wave_sync();
    store4(trow + q.n0, lin);
    wave_sync();
    const float* gb = ht.gate_b + q.n0; /* bias_ih; bias_hh after the chain */
    const f32x4 gi = wide_unit<H, 1>(st, gb, trow, q);
    const f32x4 gg = wide_unit<H, 2>(st, gb + 64, trow, q);
    const f32x4 go = wide_unit<H, 3, KN>(st, gb + 128, trow, q);
    return lstm_cell0_4(gi + bi, gg + bg, go + bo);
}

/* Representation_function: one Linear, no activation (lstm:19-21); observation in the game's xin
 * row -> hidden state there and in the global row hdst.  Root only: runtime fan-in. */
template <int SD>
__device__ __forceinline__ void lstm_representation_quad(const Model& m, const WideBufs& Bf, const QuadLane& q,
                                                         float* hdst, int hid_stride) {
    float* xrow = Bf.xin + q.c * kWideXStride;
    const int n = quad_neuron_of_lane(q.lane);
    const float4 bv = *reinterpret_cast<const float4*>(m.w + m.r_in.b_off + q.n0);
    f32x4 e = f32x4{bv.x, bv.y, bv.z, bv.w}, o = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    for (int k = 0; k < m.obs_dim; ++k) {
        const float wk = m.w[m.r_in.w_off + (size_t)k * 64 + n], xk = xrow[k];
        if (k & 1)
            o = __builtin_amdgcn_mfma_f32_4x4x1f32(wk, xk, o, 0, 0, 0);
        else
            e = __builtin_amdgcn_mfma_f32_4x4x1f32(wk, xk, e, 0, 0, 0);
    }
    f32x4 y = e + o;
    const WideValid vd = wide_valid(q, SD);
#pragma unroll
    for (int r = 0; r < 4; ++r) y[r] = vd.v[r] ? y[r] : 0.0f;
    wave_sync();
    store4(xrow + q.n0, y);
    if (q.n0 < hid_stride) store4(hdst + q.n0, y);
    wave_sync();
}

/* Dynamics_function: reward head then next-state head (its raw LSTM output is the new hidden
 * state, lstm:49-51) */
template <int SD, int H, class ST>
__device__ __forceinline__ float lstm_dynamics_quad(ST& st, const LstmTabs& T, const WideBufs& Bf,
                                                    const QuadLane& q, int action, float* hdst, int hid_stride) {
    float* xrow = Bf.xin + q.c * kWideXStride;
    float* trow = Bf.trunk + q.c * kWideTStride;
    const WideValid vd = wide_valid(q, SD);
    const f32x4 hr = lstm_head_quad<SD, H, SD>(st, T.rew, xrow, action, trow, q);
    const float reward = wide_support_mean(wide_softmax<true>(hr, vd), vd, q, SD);
    f32x4 hs = lstm_head_quad<SD, H, SD>(st, T.st, xrow, action, trow, q);
#pragma unroll
    for (int r = 0; r < 4; ++r) hs[r] = vd.v[r] ? hs[r] : 0.0f;
    wave_sync();
    store4(xrow + q.n0, hs);
    if (q.n0 < hid_stride) store4(hdst + q.n0, hs);
    wave_sync();
    return reward;
}

/* Prediction_function: policy head, value head */
template <int SD, int H, class ST>
__device__ __forceinline__ float lstm_prediction_quad(ST& st, const LstmTabs& T, const WideBufs& Bf,
                                                      const QuadLane& q, int n_actions, f32x4* prior) {
    float* xrow = Bf.xin + q.c * kWideXStride;
    float* trow = Bf.trunk + q.c * kWideTStride;
    const f32x4 hp = lstm_head_quad<SD, H, SD>(st, T.pol, xrow, -1, trow, q);
    *prior = wide_softmax<true>(hp, wide_valid(q, n_actions));
    const f32x4 hv = lstm_head_quad<SD, H, SD>(st, T.val, xrow, -1, trow, q);
    const WideValid vd = wide_valid(q, SD);
    return wide_support_mean(wide_softmax<true>(hv, vd), vd, q, SD);
}

}  // namespace mzs
#endif
