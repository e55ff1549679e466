/*
 * mzs_vision_mm.h — the three head MLPs of the 'vision_model' (147 -> H -> (H ->) n_out: reward, value,
 * policy; neural_network_vision_model.py:199-206, :262-281) on the matrix core, for the FOUR games of a
 * workgroup at once.
 *
 * mzs_vision.h runs a simulation one game per wavefront with the Linear layers as lane-per-output fma sweeps:
 * every wave streams the heads' 183 KB of weights (H = 64) from L2 by itself, and with sixteen waves on a CU
 * that stream — 2.9 MB per round of simulations through an L1 that delivers 64 bytes a clock — is what a
 * simulation waits for.  Here the convolutions stay one game per wave (a lane per pixel), but the block's four
 * waves meet behind a barrier and wave h evaluates head h for all four games with v_mfma_f32_4x4x1_16B_f32:
 * a weight row is fetched once per block instead of once per game, and the three heads run side by side.
 *
 *   A operand   lane l = neuron 64 g + l of the layer: four consecutive k per 16-byte load from a store laid out
 *               [Kpad / 4][NPAD][4] (vis_mm_offsets; K padded to 16 and fan-out to 64 with zeros)
 *   B operand   lane (kb = l >> 4, c = l & 3) holds x[game c][64 chunk + 16 kb .. + 15]; BLGP = 4 + kb broadcasts
 *   D           lane 4 b + c, register i = neuron 64 g + 4 b + i of game c
 *   contract    "dot" (include/mzs_numeric.h): an f32 MFMA is a k-ordered fmaf chain; even k into the accumulator
 *               that starts at the bias, odd k into the one that starts at +0, one add — the chains of vis_linear
 *               (zero-padded steps add +0: no effect on any bit unless a bias is -0)
 */
#ifndef MZS_VISION_MM_H
#define MZS_VISION_MM_H

namespace mzs {

__host__ __device__ inline size_t vis_mm_r(size_t n, size_t m) { return (n + m - 1) / m * m; }
__host__ __device__ inline size_t vis_mm_layer(int nin, int nout) { return vis_mm_r(nin, 16) * vis_mm_r(nout, 64); }
__host__ __device__ inline size_t vis_mm_head_floats(int H, int L, int n_out) {
    return vis_mm_layer(MZS_VIS_HIDDEN, H) + (L > 0 ? vis_mm_layer(H, H) : 0) + vis_mm_layer(H, n_out);
}
/* offsets into Model::w, behind the quad store of mzs_vision.h; heads 0 (reward), 1 (value): n_out = Sd; 2: A */
__host__ __device__ inline VisLin vis_mm_offsets(int head, int H, int L, int Sd, int A) {
    VisLin v;
    v.in_w = vis_lin_total(H, L, Sd, A) + (size_t)head * vis_mm_head_floats(H, L, Sd);
    v.mid_w = v.in_w + vis_mm_layer(MZS_VIS_HIDDEN, H);
    v.out_w = v.mid_w + (L > 0 ? vis_mm_layer(H, H) : 0);
    return v;
}
__host__ __device__ inline size_t vis_mm_total(int H, int L, int Sd, int A) {
    return 2 * vis_mm_head_floats(H, L, Sd) + vis_mm_head_floats(H, L, A);
}

constexpr int kVisMmX = 192 + 4; /* floats per game row of a head's input: 147 values, zeros to 192, + 4 (banks) */

/* sixteen k steps (lane group U's x values, BLGP = 4 + U) with the weights in wq4 */
template <int U>
__device__ __forceinline__ void vis_mm_step(const float4 (&wq4)[4], const float (&x)[16], f32x4& e, f32x4& o) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 w4 = wq4[i];
        e = __builtin_amdgcn_mfma_f32_4x4x1f32(w4.x, x[4 * i], e, 0, 0, 4 + U);
        o = __builtin_amdgcn_mfma_f32_4x4x1f32(w4.y, x[4 * i + 1], o, 0, 0, 4 + U);
        e = __builtin_amdgcn_mfma_f32_4x4x1f32(w4.z, x[4 * i + 2], e, 0, 0, 4 + U);
        o = __builtin_amdgcn_mfma_f32_4x4x1f32(w4.w, x[4 * i + 3], o, 0, 0, 4 + U);
    }
}

/* y [4][os] = [relu](b + W x) for the block's four games: xin [4][xs] in LDS (zeros from nin to the next multiple
 * of 64), W in the layout above, b in blob order. */
/* NB: register buffers of sixteen weight rows each (NB - 1 steps requested ahead of the one being multiplied):
 * 4 where a wave has 160+ registers, 2 in the build for four waves per SIMD (the other waves hide the latency) */
template <int NB>
__device__ __attribute__((noinline)) void vis_mm_linear(const float* __restrict__ W, const float* __restrict__ b, const float* xin,
                                              int xs, int nin, int nout, bool relu, float* out, int os, int ncols,
                                              int lane) {
    /* ncols: columns of `out` to write (a multiple of 4, >= nout: zeros past nout) */
    const int c = lane & 3, kb = lane >> 4, n4 = (lane >> 2) * 4;
    const int T = (nin + 15) >> 4; /* steps of sixteen k */
    const int NP = (int)vis_mm_r(nout, 64);
    const float* xrow = xin + c * xs + 16 * kb;
    for (int g0 = 0; g0 < nout; g0 += 64) {
        const float4* __restrict__ wq = reinterpret_cast<const float4*>(W) + g0 + lane;
        f32x4 e, o = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < 4; ++i) e[i] = g0 + n4 + i < nout ? b[g0 + n4 + i] : 0.0f;
        float4 wb[NB][4]; /* NB steps in flight: step t in buffer t % NB */
#pragma unroll
        for (int t = 0; t < NB - 1; ++t) {
            if (t < T) {
#pragma unroll
                for (int i = 0; i < 4; ++i) wb[t][i] = wq[(size_t)(4 * t + i) * NP];
            }
        }
        for (int t0 = 0; t0 < T; t0 += 4) {
            float x[16];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 v = *reinterpret_cast<const float4*>(xrow + 16 * t0 + 4 * i);
                x[4 * i] = v.x, x[4 * i + 1] = v.y, x[4 * i + 2] = v.z, x[4 * i + 3] = v.w;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = t0 + u;
                if (t < T) {
                    if (t + NB - 1 < T) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) wb[(u + NB - 1) % NB][i] = wq[(size_t)(4 * (t + NB - 1) + i) * NP];
                    }
                    switch (u) {
                        case 0: vis_mm_step<0>(wb[0 % NB], x, e, o); break;
                        case 1: vis_mm_step<1>(wb[1 % NB], x, e, o); break;
                        case 2: vis_mm_step<2>(wb[2 % NB], x, e, o); break;
                        default: vis_mm_step<3>(wb[3 % NB], x, e, o); break;
                    }
                }
            }
        }
        f32x4 y = e + o;
        if (relu) {
#pragma unroll
            for (int i = 0; i < 4; ++i) y[i] = !(y[i] > 0.0f) ? 0.0f : y[i];
        }
        if (g0 + n4 < ncols) *reinterpret_cast<float4*>(out + c * os + g0 + n4) = make_float4(y[0], y[1], y[2], y[3]);
    }
}

/* block-shared LDS of the head phase.  Per head a region of four input rows (hxs floats each: the 147 outputs of
 * the head's 1 x 1 convolution, zeros to 192); once the first layer has read them the region also serves as the
 * second activation buffer (a1: [4][as]), which only lives between the two barriers of the head phase — the
 * convolutions of the next simulation rewrite the rows and their zero tails. */
struct VisMmLds {
    float* hx;  /* [3][4][hxs]  inputs of the heads; a1 of head h = its region            */
    float* a0;  /* [3][4][as]   activations                                                */
    float* lg;  /* [3][4][ls]   logits                                                     */
    int hxs, as, ls;
};
__host__ __device__ inline int vis_mm_act_stride(int H) { return (int)vis_mm_r(H, 64) + 4; }
__host__ __device__ inline int vis_mm_lg_stride(int Sd, int A) { return (int)vis_mm_r(Sd > A ? Sd : A, 4) + 4; }
__host__ __device__ inline int vis_mm_hx_stride(int H) {
    const int as = vis_mm_act_stride(H);
    return as > kVisMmX ? as : kVisMmX;
}
__host__ __device__ inline size_t vis_mm_lds_floats(int H, int Sd, int A) {
    return (size_t)3 * 4 * (vis_mm_hx_stride(H) + vis_mm_act_stride(H) + vis_mm_lg_stride(Sd, A));
}
__device__ __forceinline__ VisMmLds vis_mm_carve(float* base, int H, int Sd, int A) {
    VisMmLds M;
    M.hxs = vis_mm_hx_stride(H), M.as = vis_mm_act_stride(H), M.ls = vis_mm_lg_stride(Sd, A);
    M.hx = base;
    M.a0 = M.hx + 3 * 4 * M.hxs;
    M.lg = M.a0 + 3 * 4 * M.as;
    return M;
}

/* head `hd` (0 reward, 1 value, 2 policy) for the four games: M.hx[hd] -> M.lg[hd]; one wavefront */
template <int NB>
__device__ __forceinline__ void vis_mm_head(const Model& m, const VisMmLds& M, int hd, int lane) {
    const mzs_vision_offsets o = vis_offsets(m);
    const size_t fc_base = hd == 0 ? o.d_fc.in_w : (hd == 1 ? o.p_v.in_w : o.p_p.in_w);
    const int n_out = hd == 2 ? m.A : m.Sd;
    const size_t H = (size_t)m.H, in_b = fc_base + H * kVisHidden, mid_w = in_b + H, mid_b = mid_w + (m.L > 0 ? H * H : 0);
    const size_t out_w = mid_b + (m.L > 0 ? H : 0), out_b = out_w + (size_t)n_out * H;
    const VisLin lw = vis_mm_offsets(hd, m.H, m.L, m.Sd, m.A);
    const float* __restrict__ blob = m.blob;
    float* hx = M.hx + (size_t)hd * 4 * M.hxs;
    float* a0 = M.a0 + (size_t)hd * 4 * M.as;
    float* a1 = hx; /* free once the first layer has read it: rows of stride as inside the head's region */
    const int hp = (int)vis_mm_r(m.H, 64);
    vis_mm_linear<NB>(m.w + lw.in_w, blob + in_b, hx, M.hxs, kVisHidden, m.H, true, a0, M.as, hp, lane);
    wave_sync();
    float *src = a0, *dst = a1;
#pragma nounroll
    for (int l = 0; l < m.L; ++l) {
        vis_mm_linear<NB>(m.w + lw.mid_w, blob + mid_b, src, M.as, m.H, m.H, true, dst, M.as, hp, lane);
        wave_sync();
        float* t = src;
        src = dst, dst = t;
    }
    vis_mm_linear<NB>(m.w + lw.out_w, blob + out_b, src, M.as, m.H, n_out, false, M.lg + (size_t)hd * 4 * M.ls, M.ls,
                  (int)vis_mm_r(n_out, 4), lane);
    wave_sync();
}

/* the head's 1 x 1 convolution (+ bias) of x [Ci][49] -> dst [147] (vis_head_mlp's first stage) */
template <class WP>
__device__ __forceinline__ void vis_head_conv(WP cw, size_t conv_w, size_t conv_b, int Ci, const float* x,
                                              float* dst, int lane) {
    constexpr int CM = kVisC + 1;
    for (int co = 0; co < kVisC; ++co) {
        float wv[CM];
#pragma unroll
        for (int ci = 0; ci < CM; ++ci) wv[ci] = cw[conv_w + co * Ci + (ci < Ci ? ci : 0)];
        const float bias = cw[conv_b + co];
        for (int i = lane; i < kVisPix; i += kWave) {
            float xv[CM];
#pragma unroll
            for (int ci = 0; ci < CM; ++ci) xv[ci] = x[(ci < Ci ? ci : 0) * kVisPix + i];
            float acc = bias;
#pragma unroll
            for (int ci = 0; ci < CM; ++ci) {
                const float a2 = mzs_fma(wv[ci], xv[ci], acc);
                acc = ci < Ci ? a2 : acc;
            }
            dst[co * kVisPix + i] = acc;
        }
    }
    if (kVisHidden + lane < 192) dst[kVisHidden + lane] = 0.0f; /* the row's zero tail (the head phase reuses the rows) */
    wave_sync();
}

/* vis_dynamics without its head MLP: hidden state in L.xh + action -> the reward head's input in hrow, the
 * next hidden state in L.xh and gdst (may be null) */
__device__ __forceinline__ void vis_dynamics_conv(const Model& m, const WaveLds& L, int action, float* gdst, float* hrow,
                                                  int lane) {
    const mzs_vision_offsets o = vis_offsets(m);
    float *x = L.ob, *t1 = L.xa, *t2 = L.xb;
    const float pv = (float)(action + 1) / (float)m.A;
    for (int i = lane; i < kVisPix; i += kWave) L.xh[kVisHidden + i] = pv;
    wave_sync();
    const VisConstPtr cw = (VisConstPtr)m.blob; /* uniform addresses: scalar loads, weights as scalar operands */
    vis_conv3_bn<kVisC + 1, kVisC>(cw, o.d_conv, o.d_bn, L.xh, x, lane);
    for (int l = 0; l < m.L; ++l) vis_res_block(cw, o.d_res.conv_a, kVisC, x, MZS_VIS_S4, t1, t2, lane);
    vis_head_conv(cw, o.d_conv_rew, o.d_conv_rew_b, kVisC + 1, L.xh, hrow, lane);
    for (int i = lane; i < kVisHidden; i += kWave) {
        const float v = x[i] > 0.0f ? x[i] : 0.0f;
        L.xh[i] = v;
        if (gdst) gdst[i] = v;
    }
    wave_sync();
}

/* vis_prediction without its head MLPs: hidden state in L.xh (kept) -> the value and policy heads' inputs */
__device__ __forceinline__ void vis_prediction_conv(const Model& m, const WaveLds& L, float* hrow_val, float* hrow_pol,
                                                    int lane) {
    const mzs_vision_offsets o = vis_offsets(m);
    float *x = L.ob, *t1 = L.xa, *t2 = L.xb;
    for (int i = lane; i < kVisHidden; i += kWave) x[i] = L.xh[i];
    wave_sync();
    const VisConstPtr cw = (VisConstPtr)m.blob;
    for (int l = 0; l < m.L; ++l) vis_res_block(cw, o.p_res.conv_a, kVisC, x, MZS_VIS_S4, t1, t2, lane);
    vis_head_conv(cw, o.p_conv_val, o.p_conv_val_b, kVisC, x, hrow_val, lane);
    vis_head_conv(cw, o.p_conv_pol, o.p_conv_pol_b, kVisC, x, hrow_pol, lane);
}

}  // namespace mzs
#endif
