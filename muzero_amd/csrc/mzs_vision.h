/*
 * mzs_vision.h — the 'vision_model' (ResNet-v2 on a 98 x 98 x 3 image) for one wavefront: the three
 * model functions the search and inference kernels call, in the shape of mzs_device.h's model_*.
 * Included by mzs_device.h.
 *
 * Reference code restated (file:line in /root/reference), eval mode (BatchNorm2d on its running
 * statistics, Dropout2d off):
 *   vis_res_block        neural_network_vision_model.py:41-82   Residual_block (v2, pre-activation)
 *   vis_representation   :84-122 Down_sample, :125-163 Representation_function (down_sampling=True)
 *   vis_dynamics         :167-230 Dynamics_function + the action plane of muzero_model.py:505-516
 *   vis_prediction       :234-300 Prediction_function; softmax / support -> scalar :898-903
 * Arithmetic contract: the one stated in oracle/mz_vision_oracle.c (fma chains in (ci, ky, kx) order
 * with the padding taps skipped, batchnorm as fma(x - mean, gamma / sqrt(var + eps), beta), average
 * pool as ordered sum / 9, Linear as the even / odd "dot" chains) — GPU == oracle bit for bit.
 *
 * Channel counts are 1..4 and the maps 49 x 49 down to 7 x 7: HBM- and latency-bound work with K =
 * 9..36 per output, nothing for MFMA.  Lanes own output pixels; every map lives in the wave's LDS
 * slice (three buffers of 49 x 49 floats for the representation, 7 x 7 x 4 afterwards), the weights
 * (a few hundred floats per convolution) are read from the blob through the scalar/L1 caches.
 */
#ifndef MZS_VISION_H
#define MZS_VISION_H

namespace mzs {

constexpr int kVisC = MZS_VIS_C, kVisHidden = MZS_VIS_HIDDEN, kVisPix = MZS_VIS_S4 * MZS_VIS_S4;
constexpr int kVisBuf = 2432;  /* floats of one scratch map: >= 49 * 49, >= 3 * 25 * 25 */
constexpr int kVisXh = 208;    /* hidden state (147) + action plane (49), rounded */

/* out [Co][So][So] = conv3x3(in [Ci][Si][Si]), stride s, padding 1, no bias; in: LDS or global.
 * One output per lane and pass; its Ci x 9 taps are all fetched (inputs and weights; a padding tap
 * reads a clamped address and is marked) BEFORE the fma chain that consumes them in (ci, ky, kx)
 * order, padding taps skipped by a select — the wave is alone on its SIMD, so a chain that waits for
 * a load per tap (as the plain loop did) is bound by 27-36 load latencies per output. */
template <int CI>
__device__ __forceinline__ void vis_conv3_t(const float* __restrict__ w, const float* in, int Si, int Co, int s,
                                            float* out, int lane) {
    const int So = (Si - 1) / s + 1, n = So * So;
    for (int idx0 = 0; idx0 < Co * n; idx0 += kWave) {
        const int idx = idx0 + lane;
        const bool live = idx < Co * n;
        const int id = live ? idx : 0;
        const int co = id / n, r = id - co * n, oy = r / So, ox = r - oy * So;
        float xv[CI * 9], wv[CI * 9];
        bool ok[9];
        int off[9];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int iy = oy * s + ky - 1, ix = ox * s + kx - 1;
                const bool v = iy >= 0 && iy < Si && ix >= 0 && ix < Si;
                ok[ky * 3 + kx] = v;
                off[ky * 3 + kx] = v ? iy * Si + ix : 0;
            }
        }
#pragma unroll
        for (int ci = 0; ci < CI; ++ci) {
            const float* __restrict__ wk = w + (co * CI + ci) * 9;
            const float* ip = in + (size_t)ci * Si * Si;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                wv[ci * 9 + t] = wk[t];
                xv[ci * 9 + t] = ip[off[t]];
            }
        }
        float acc = 0.0f;
#pragma unroll
        for (int ci = 0; ci < CI; ++ci) {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float a2 = mzs_fma(wv[ci * 9 + t], xv[ci * 9 + t], acc);
                acc = ok[t] ? a2 : acc;
            }
        }
        if (live) out[idx] = acc;
    }
    wave_sync();
}

__device__ __forceinline__ void vis_conv3(const float* __restrict__ w, const float* in, int Ci, int Si, int Co, int s,
                                          float* out, int lane) {
    switch (Ci) { /* the model's channel counts: C / 2, C, C + 1 with C = MZS_VIS_C */
        case 1: vis_conv3_t<1>(w, in, Si, Co, s, out, lane); return;
        case 2: vis_conv3_t<2>(w, in, Si, Co, s, out, lane); return;
        case 3: vis_conv3_t<3>(w, in, Si, Co, s, out, lane); return;
        case 4: vis_conv3_t<4>(w, in, Si, Co, s, out, lane); return;
        default: break;
    }
    const int So = (Si - 1) / s + 1, n = So * So;
    for (int idx = lane; idx < Co * n; idx += kWave) {
        const int co = idx / n, r = idx - co * n, oy = r / So, ox = r - oy * So;
        float acc = 0.0f;
        for (int ci = 0; ci < Ci; ++ci) {
            const float* __restrict__ wk = w + (co * Ci + ci) * 9;
            const float* ip = in + (size_t)ci * Si * Si;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int iy = oy * s + ky - 1;
                if (iy < 0 || iy >= Si) continue;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int ix = ox * s + kx - 1;
                    if (ix < 0 || ix >= Si) continue;
                    acc = mzs_fma(wk[ky * 3 + kx], ip[iy * Si + ix], acc);
                }
            }
        }
        out[idx] = acc;
    }
    wave_sync();
}

/* out = relu(batchnorm(in)) on [C][n], eval mode */
__device__ __forceinline__ void vis_bn_relu(const float* __restrict__ blob, const mzs_vis_bn& bn, const float* in, int C,
                                            int n, float* out, int lane) {
    /* the per-channel constants once per call (C <= kVisC + 1 channels, their loads in flight together),
     * not once per element: gamma / sqrt(var + eps) is the same value either way */
    constexpr int CM = kVisC + 1;
    float inv[CM], mu[CM], be[CM];
#pragma unroll
    for (int c = 0; c < CM; ++c) {
        const int cc = c < C ? c : 0;
        inv[c] = blob[bn.w + cc], mu[c] = blob[bn.var + cc], be[c] = blob[bn.b + cc];
    }
#pragma unroll
    for (int c = 0; c < CM; ++c) {
        const int cc = c < C ? c : 0;
        inv[c] = inv[c] / __builtin_sqrtf(mu[c] + 1e-5f);
        mu[c] = blob[bn.mean + cc];
    }
    for (int idx = lane; idx < C * n; idx += kWave) {
        const int c = idx / n;
        float iv = inv[0], m0 = mu[0], b0 = be[0];
#pragma unroll
        for (int k = 1; k < CM; ++k) {
            iv = c == k ? inv[k] : iv;
            m0 = c == k ? mu[k] : m0;
            b0 = c == k ? be[k] : b0;
        }
        const float y = mzs_fma(in[idx] - m0, iv, b0);
        out[idx] = y > 0.0f ? y : 0.0f;
    }
    wave_sync();
}

/* x += conv_a(R(conv_b(R(conv_a(R(x)))))), R = relu(batchnorm(.)); x [C][S][S]; t1, t2 scratch */
__device__ __forceinline__ void vis_res_block(const float* __restrict__ blob, const mzs_vis_res& r, float* x, int S,
                                              float* t1, float* t2, int lane) {
    const int C = r.C, n = S * S;
    vis_bn_relu(blob, r.bn, x, C, n, t1, lane);
    vis_conv3(blob + r.conv_a, t1, C, S, C, 1, t2, lane);
    vis_bn_relu(blob, r.bn, t2, C, n, t1, lane);
    vis_conv3(blob + r.conv_b, t1, C, S, C, 1, t2, lane);
    vis_bn_relu(blob, r.bn, t2, C, n, t1, lane);
    vis_conv3(blob + r.conv_a, t1, C, S, C, 1, t2, lane);
    for (int i = lane; i < C * n; i += kWave) x[i] = t2[i] + x[i];
    wave_sync();
}

/* AvgPool2d(3, stride 2, padding 1), count_include_pad */
__device__ __forceinline__ void vis_avgpool(const float* in, int C, int Si, float* out, int lane) {
    const int So = (Si - 1) / 2 + 1, n = So * So;
    for (int idx = lane; idx < C * n; idx += kWave) {
        const int c = idx / n, r = idx - c * n, oy = r / So, ox = r - oy * So;
        float acc = 0.0f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * 2 + ky - 1;
            if (iy < 0 || iy >= Si) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * 2 + kx - 1;
                if (ix < 0 || ix >= Si) continue;
                acc = acc + in[(c * Si + iy) * Si + ix];
            }
        }
        out[idx] = acc / 9.0f;
    }
    wave_sync();
}

/* y = [relu](b + W x), contract "dot" (even / odd chains); Wt = W^T [nin][nout] (the transposed copy
 * mzs_set_weights makes: lanes read neighbouring addresses), sixteen rows in flight ahead of the chains */
__device__ __forceinline__ void vis_linear(const float* __restrict__ Wt, const float* __restrict__ b, const float* x,
                                           int nout, int nin, bool relu, float* y, int lane) {
    for (int j = lane; j < nout; j += kWave) {
        const float* __restrict__ w = Wt + j;
        float e = b[j], o = 0.0f;
        int k = 0;
        for (; k + 15 < nin; k += 16) {
            float wv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) wv[u] = w[(size_t)(k + u) * nout];
#pragma unroll
            for (int u = 0; u < 16; u += 2) {
                e = mzs_fma(wv[u], x[k + u], e);
                o = mzs_fma(wv[u + 1], x[k + u + 1], o);
            }
        }
        for (; k + 1 < nin; k += 2) {
            e = mzs_fma(w[(size_t)k * nout], x[k], e);
            o = mzs_fma(w[(size_t)(k + 1) * nout], x[k + 1], o);
        }
        if (k < nin) e = mzs_fma(w[(size_t)k * nout], x[k], e);
        const float v = e + o;
        y[j] = relu && !(v > 0.0f) ? 0.0f : v;
    }
    wave_sync();
}

/* 1x1 convolution (+bias) of x [Ci][49] -> flatten -> Linear, relu, (Linear, relu) x L, Linear: logits */
__device__ __forceinline__ void vis_head_mlp(const Model& m, const float* __restrict__ blob, size_t conv_w, size_t conv_b,
                                             int Ci, const mzs_vis_mlp& fc, const float* x, float* logits, float* t1,
                                             float* t2, int lane) {
    for (int idx = lane; idx < kVisC * kVisPix; idx += kWave) {
        const int co = idx / kVisPix, i = idx - co * kVisPix;
        constexpr int CM = kVisC + 1;
        float wv[CM], xv[CM]; /* the taps' loads in flight together, then the chain */
#pragma unroll
        for (int ci = 0; ci < CM; ++ci) {
            const int cc = ci < Ci ? ci : 0;
            wv[ci] = blob[conv_w + co * Ci + cc], xv[ci] = x[cc * kVisPix + i];
        }
        float acc = blob[conv_b + co];
#pragma unroll
        for (int ci = 0; ci < CM; ++ci) {
            const float a2 = mzs_fma(wv[ci], xv[ci], acc);
            acc = ci < Ci ? a2 : acc;
        }
        t1[idx] = acc;
    }
    wave_sync();
    vis_linear(m.w + fc.in_w, blob + fc.in_b, t1, m.H, kVisHidden, true, t2, lane);
    float *a = t2, *bb = t1;
    for (int l = 0; l < m.L; ++l) {
        vis_linear(m.w + fc.mid_w, blob + fc.mid_b, a, m.H, m.H, true, bb, lane);
        float* t = a;
        a = bb;
        bb = t;
    }
    vis_linear(m.w + fc.out_w, blob + fc.out_b, a, fc.n_out, m.H, false, logits, lane);
}

__device__ __forceinline__ mzs_vision_offsets vis_offsets(const Model& m) {
    mzs_dims d;
    d.obs_dim = m.obs_dim, d.n_actions = m.A, d.state_dim = m.Sd, d.hidden_dim = m.H, d.n_mid_layers = m.L;
    d.model_kind = MZS_MODEL_VISION, d.precision = 0;
    return mzs_vision_layout(&d);
}

/* obs_g [3][98][98] (global) -> hidden state [3][7][7] in L.xh and gdst */
__device__ __forceinline__ void vis_representation(const Model& m, const WaveLds& L, const float* obs_g, float* gdst,
                                                   int lane) {
    const mzs_vision_offsets o = vis_offsets(m);
    const float* __restrict__ blob = m.blob;
    float *x = L.ob, *t1 = L.xa, *t2 = L.xb;
    vis_conv3(blob + o.r_conv_in, obs_g, kVisC, MZS_VIS_SIDE, kVisC / 2, 2, x, lane); /* 1 x 49 x 49 */
    vis_res_block(blob, o.r_res_in, x, MZS_VIS_S1, t1, t2, lane);
    vis_res_block(blob, o.r_res_in, x, MZS_VIS_S1, t1, t2, lane);
    vis_conv3(blob + o.r_conv_out, x, kVisC / 2, MZS_VIS_S1, kVisC, 2, t1, lane); /* 3 x 25 x 25 */
    {
        float* t = x;
        x = t1, t1 = t;
    }
    vis_res_block(blob, o.r_res_out, x, MZS_VIS_S2, t1, t2, lane);
    vis_res_block(blob, o.r_res_out, x, MZS_VIS_S2, t1, t2, lane);
    vis_avgpool(x, kVisC, MZS_VIS_S2, t1, lane); /* 3 x 13 x 13 */
    {
        float* t = x;
        x = t1, t1 = t;
    }
    for (int i = 0; i < 3; ++i) vis_res_block(blob, o.r_res_out, x, MZS_VIS_S3, t1, t2, lane);
    vis_avgpool(x, kVisC, MZS_VIS_S3, t1, lane); /* 3 x 7 x 7 */
    {
        float* t = x;
        x = t1, t1 = t;
    }
    vis_res_block(blob, o.r_res, x, MZS_VIS_S4, t1, t2, lane);
    for (int i = lane; i < kVisHidden; i += kWave) {
        const float v = x[i];
        L.xh[i] = v;
        if (gdst) gdst[i] = v;
    }
    wave_sync();
}

/* hidden state in L.xh + action -> reward (returned, uniform); next hidden state to L.xh and gdst */
__device__ __forceinline__ float vis_dynamics(const Model& m, const WaveLds& L, int action, float* gdst, int lane) {
    const mzs_vision_offsets o = vis_offsets(m);
    const float* __restrict__ blob = m.blob;
    float *x = L.ob, *t1 = L.xa, *t2 = L.xb, *logits = L.ob + 256;
    /* cat(state, action plane): the plane holds (a + 1) / A (muzero_model.py:515) */
    const float pv = (float)(action + 1) / (float)m.A;
    for (int i = lane; i < kVisPix; i += kWave) L.xh[kVisHidden + i] = pv;
    wave_sync();
    vis_conv3(blob + o.d_conv, L.xh, kVisC + 1, MZS_VIS_S4, kVisC, 1, t1, lane);
    vis_bn_relu(blob, o.d_bn, t1, kVisC, kVisPix, x, lane);
    for (int l = 0; l < m.L; ++l) vis_res_block(blob, o.d_res, x, MZS_VIS_S4, t1, t2, lane);
    vis_head_mlp(m, blob, o.d_conv_rew, o.d_conv_rew_b, kVisC + 1, o.d_fc, L.xh, logits, t1, t2, lane);
    const float reward = seg_support_scalar(logits, m.Sd, lane);
    for (int i = lane; i < kVisHidden; i += kWave) {
        const float v = x[i] > 0.0f ? x[i] : 0.0f; /* the Sequential's closing ReLU (:215) */
        L.xh[i] = v;
        if (gdst) gdst[i] = v;
    }
    wave_sync();
    return reward;
}

/* hidden state in L.xh (kept) -> softmax policy as lane pairs, value (uniform) */
__device__ __forceinline__ float vis_prediction(const Model& m, const WaveLds& L, Pair* policy, int lane) {
    const mzs_vision_offsets o = vis_offsets(m);
    const float* __restrict__ blob = m.blob;
    float *x = L.ob, *t1 = L.xa, *t2 = L.xb, *logits = L.ob + 256;
    for (int i = lane; i < kVisHidden; i += kWave) x[i] = L.xh[i];
    wave_sync();
    for (int l = 0; l < m.L; ++l) vis_res_block(blob, o.p_res, x, MZS_VIS_S4, t1, t2, lane);
    vis_head_mlp(m, blob, o.p_conv_val, o.p_conv_val_b, kVisC, o.p_v, x, logits, t1, t2, lane);
    const float value = seg_support_scalar(logits, m.Sd, lane);
    wave_sync();
    vis_head_mlp(m, blob, o.p_conv_pol, o.p_conv_pol_b, kVisC, o.p_p, x, logits, t1, t2, lane);
    *policy = seg_softmax(seg_load(logits, m.A, lane));
    wave_sync();
    return value;
}

}  // namespace mzs
#endif
