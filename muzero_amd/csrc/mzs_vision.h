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
 * slice (three buffers of 49 x 49 floats — one with a border — for the representation, 7 x 7 x 4 afterwards),
 * and so does a copy of every convolution's and batchnorm's parameters (vis_cache_load, once per kernel);
 * the Linear layers of the head MLPs stream their weights from a store laid out for 16-byte loads
 * (vis_lin_offsets) in global memory.  A search runs the roots in k_inference (representation: one wave per
 * SIMD for its LDS) and the simulations in k_search_fused (dynamics + prediction: up to four waves per SIMD).
 */
#ifndef MZS_VISION_H
#define MZS_VISION_H

namespace mzs {

constexpr int kVisC = MZS_VIS_C, kVisHidden = MZS_VIS_HIDDEN, kVisPix = MZS_VIS_S4 * MZS_VIS_S4;
constexpr int kVisBuf = 2432;  /* floats of one scratch map: >= 49 * 49, >= 3 * 25 * 25, >= 3 * 27 * 27 (with border) */
constexpr int kVisPadBuf = 2608; /* the map with its border at 49 x 49: 51 * 51 */
constexpr int kVisXh = 208;    /* hidden state (147) + action plane (49), rounded */

/* The small parameters (every convolution and batchnorm: blob ranges [0, d_fc.in_w) and [p_res.conv_a,
 * p_v.in_w), 826 + 198 floats at C = 3) are copied once per kernel into the wave's LDS slice (L.vc): a
 * simulation reads them ~35 times, and from the blob every read was a trip to L2 (~0.4 us) that the wave,
 * alone on its SIMD, waited out.  Blob offsets of the first range index the cache as they are, those of
 * the second less vis_cache_delta. */
constexpr int kVisCache = 1024; /* floats; mzs_create checks that the two ranges fit */

/* what to subtract from a blob offset of the second range to index the cache */
__device__ __forceinline__ size_t vis_cache_delta(const mzs_vision_offsets& o) { return o.p_res.conv_a - o.d_fc.in_w; }

__device__ __forceinline__ void vis_cache_load(const float* __restrict__ blob, const mzs_vision_offsets& o, float* vc,
                                               int lane) {
    const int n1 = (int)o.d_fc.in_w, n2 = (int)(o.p_v.in_w - o.p_res.conv_a);
    for (int i = lane; i < n1; i += kWave) vc[i] = blob[i];
    for (int i = lane; i < n2; i += kWave) vc[n1 + i] = blob[o.p_res.conv_a + i];
    wave_sync();
}

/* the same into ONE copy per workgroup (the search kernel: the per-wave LDS slice decides how many waves a
 * SIMD holds); every thread of the block calls it, before any wave leaves */
__device__ __forceinline__ void vis_cache_load_block(const float* __restrict__ blob, const mzs_vision_offsets& o,
                                                     float* vc) {
    const int n1 = (int)o.d_fc.in_w, n2 = (int)(o.p_v.in_w - o.p_res.conv_a);
    for (int i = threadIdx.x; i < n1; i += blockDim.x) vc[i] = blob[i];
    for (int i = threadIdx.x; i < n2; i += blockDim.x) vc[n1 + i] = blob[o.p_res.conv_a + i];
    __syncthreads();
}

/* out [Co][So][So] = conv3x3(in [Ci][Si][Si]), stride s, padding 1, no bias; in: LDS or global; w: the
 * cached weights.  Output channels in turn (their Ci x 9 weights in registers, the same for every lane),
 * one output pixel per lane and pass; the pass's Ci x 9 inputs are all fetched (a padding tap reads a
 * clamped address and is marked) BEFORE the fma chain that consumes them in (ci, ky, kx) order, padding
 * taps skipped by a select — the wave is alone on its SIMD, so a chain that waits for a load per tap (as
 * the plain loop did) is bound by 27-36 load latencies per output. */
template <int CI, int NP = 1>
__device__ __forceinline__ void vis_conv3_t(const float* w, const float* in, int Si, int Co, int s, float* out,
                                            int lane) {
    /* NP: passes whose taps are requested together (the first convolution reads the image from global
     * memory: a pass at a time waits out a trip to HBM per 64 outputs, 38 of them) */
    const int So = (Si - 1) / s + 1, n = So * So;
    const float inv_so = 1.0f / (float)So; /* r / So below: r < 2401, So <= 49, exact with the half offset */
    for (int co = 0; co < Co; ++co) {
        float wv[CI * 9];
#pragma unroll
        for (int j = 0; j < CI * 9; ++j) wv[j] = w[co * CI * 9 + j];
        for (int r0 = 0; r0 < n; r0 += NP * kWave) {
            float xv[NP][CI * 9];
            unsigned okm[NP]; /* bit t: tap t lies inside the map; bit 9: the output exists */
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const bool live = r0 + p * kWave + lane < n;
                const int r = live ? r0 + p * kWave + lane : 0;
                const int oy = (int)(((float)r + 0.5f) * inv_so), ox = r - oy * So;
                int off[9];
                unsigned m = live ? 512u : 0u;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int iy = oy * s + ky - 1, ix = ox * s + kx - 1;
                        const bool v = iy >= 0 && iy < Si && ix >= 0 && ix < Si;
                        m |= v ? 1u << (ky * 3 + kx) : 0u;
                        off[ky * 3 + kx] = v ? iy * Si + ix : 0;
                    }
                }
                okm[p] = m;
#pragma unroll
                for (int ci = 0; ci < CI; ++ci) {
                    const float* ip = in + (size_t)ci * Si * Si;
#pragma unroll
                    for (int t = 0; t < 9; ++t) xv[p][ci * 9 + t] = ip[off[t]];
                }
            }
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                float acc = 0.0f;
#pragma unroll
                for (int ci = 0; ci < CI; ++ci) {
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        const float a2 = mzs_fma(wv[ci * 9 + t], xv[p][ci * 9 + t], acc);
                        acc = (okm[p] >> t) & 1u ? a2 : acc;
                    }
                }
                if (okm[p] & 512u) out[co * n + r0 + p * kWave + lane] = acc;
            }
        }
    }
    wave_sync();
}

__device__ __forceinline__ void vis_conv3(const float* w, const float* in, int Ci, int Si, int Co, int s, float* out,
                                          int lane) {
    switch (Ci) { /* the model's channel counts: C / 2, C, C + 1 with C = MZS_VIS_C */
        case 1: vis_conv3_t<1>(w, in, Si, Co, s, out, lane); return;
        case 2: vis_conv3_t<2>(w, in, Si, Co, s, out, lane); return;
        case 3: vis_conv3_t<3>(w, in, Si, Co, s, out, lane); return;
        default: vis_conv3_t<4>(w, in, Si, Co, s, out, lane); return;
    }
}

/* A residual block's stage, conv3x3(relu(batchnorm(in))) with C -> C channels, in two passes over a map with
 * a border: vis_bn_relu_pad writes relu(batchnorm(in)) into pad [C][S + 2][S + 2] with a border of +0 (the
 * convolution's zero padding, materialised), vis_conv3_pad then reads a pixel's C x 9 taps at constant offsets
 * from one address — no bounds tests, no selects — and runs the C output channels' chains over them in
 * (ci, ky, kx) order.  A padding tap contributes fma(w, +0, acc) == acc bit for bit (acc starts at +0 and a
 * sum is -0 only if both addends are), so adding it and skipping it, as the contract words it, are the same
 * for finite weights.  Measured on the 49 x 49 stage: 800 cycles per 64 outputs with the bounds tests and
 * selects of the plain form (~130 instructions for 9 taps), a third of that in this form. */
/* WP: where the parameters are read from — a pointer into the LDS cache (const float*), or the blob itself
 * through the constant address space (VisConstPtr): the addresses are uniform, so the loads become scalar
 * loads and the weights reach the fma chains as scalar operands (k_search_vis4) */
typedef const __attribute__((address_space(4))) float* VisConstPtr;
template <int C, class WP>
__device__ __forceinline__ void vis_bn_relu_pad(WP cw, const mzs_vis_bn& bn, const float* in, int S, float* pad,
                                                int lane) {
    const int P = S + 2, n = S * S, np = P * P;
    const float inv_p = 1.0f / (float)P; /* q / P below: q < 2601, P <= 51, exact with the half offset */
    float iv[C], mu[C], be[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float g = cw[bn.w + c], var = cw[bn.var + c];
        mu[c] = cw[bn.mean + c], be[c] = cw[bn.b + c];
        iv[c] = g / __builtin_sqrtf(var + 1e-5f);
    }
    for (int q0 = 0; q0 < np; q0 += kWave) {
        const int q = q0 + lane;
        const bool live = q < np;
        const int py = (int)(((float)(live ? q : 0) + 0.5f) * inv_p), px = (live ? q : 0) - py * P;
        const bool inner = live && py >= 1 && py <= S && px >= 1 && px <= S;
        const int src = inner ? (py - 1) * S + px - 1 : 0;
        float xv[C];
#pragma unroll
        for (int c = 0; c < C; ++c) xv[c] = in[c * n + src];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float y = mzs_fma(xv[c] - mu[c], iv[c], be[c]);
            if (live) pad[c * np + q] = inner && y > 0.0f ? y : 0.0f;
        }
    }
    wave_sync();
}

template <int C, bool kHoist, class WP>
__device__ __forceinline__ void vis_conv3_pad(WP cw, size_t w_off, const float* pad, int S, float* out,
                                              int lane) {
    const int P = S + 2, n = S * S, np = P * P;
    /* kHoist (the representation: up to 38 passes per stage): the C x C x 9 weights in registers for all passes;
     * else (one pass on a 7 x 7 map, four waves per SIMD on 128 registers) read as the chains consume them */
    float wv[kHoist ? C * C * 9 : 1];
    if constexpr (kHoist) {
#pragma unroll
        for (int j = 0; j < C * C * 9; ++j) wv[j] = cw[w_off + j];
    }
    const float inv_s = 1.0f / (float)S; /* r / S below: r < 2401, S <= 49, exact with the half offset */
    for (int r0 = 0; r0 < n; r0 += kWave) {
        const bool live = r0 + lane < n;
        const int r = live ? r0 + lane : 0;
        const int oy = (int)(((float)r + 0.5f) * inv_s), ox = r - oy * S;
        const float* tp = pad + oy * P + ox; /* the window's top-left tap */
        float tv[C * 9];
#pragma unroll
        for (int ci = 0; ci < C; ++ci) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) tv[ci * 9 + ky * 3 + kx] = tp[ci * np + ky * P + kx];
            }
        }
#pragma unroll
        for (int co = 0; co < C; ++co) {
            float acc = 0.0f;
#pragma unroll
            for (int j = 0; j < C * 9; ++j) {
                float w;
                if constexpr (kHoist) w = wv[co * C * 9 + j];
                else w = cw[w_off + co * C * 9 + j];
                acc = mzs_fma(w, tv[j], acc);
            }
            if (live) out[co * n + r] = acc;
        }
    }
    wave_sync();
}

/* out [CO][7][7] = relu(batchnorm(conv3x3(in [CI][7][7]))), stride 1, padding 1: the dynamics net's first
 * stage (one pass: 49 pixels), a lane per pixel with its CI x 9 taps read once for all CO chains, as above */
template <int CI, int CO, class WP>
__device__ __forceinline__ void vis_conv3_bn(WP cw, size_t w_off, const mzs_vis_bn& bn, const float* in,
                                             float* out, int lane) {
    constexpr int S = MZS_VIS_S4, n = S * S;
    const bool live = lane < n;
    const int r = live ? lane : 0;
    const int oy = r / S, ox = r - oy * S;
    float tv[CI * 9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int iy = oy + ky - 1, ix = ox + kx - 1;
            const bool v = iy >= 0 && iy < S && ix >= 0 && ix < S;
            const int off = v ? iy * S + ix : 0;
#pragma unroll
            for (int ci = 0; ci < CI; ++ci) {
                const float x = in[ci * n + off];
                tv[ci * 9 + ky * 3 + kx] = v ? x : 0.0f;
            }
        }
    }
#pragma unroll
    for (int co = 0; co < CO; ++co) {
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < CI * 9; ++j) acc = mzs_fma(cw[w_off + co * CI * 9 + j], tv[j], acc);
        const float g = cw[bn.w + co], var = cw[bn.var + co], mu = cw[bn.mean + co], be = cw[bn.b + co];
        const float y = mzs_fma(acc - mu, g / __builtin_sqrtf(var + 1e-5f), be);
        if (live) out[co * n + r] = y > 0.0f ? y : 0.0f;
    }
    wave_sync();
}

/* x += conv_a(R(conv_b(R(conv_a(R(x)))))), R = relu(batchnorm(.)); x [C][S][S]; t1, t2 scratch.  The block's
 * parameters sit at `base` of the cache in mzs_vis_res_layout's order (conv_a, conv_b, batchnorm): plain
 * scalars — offsets selected out of the layout structs ended up as indexed loads from scratch memory */
template <bool kHoist = false, class WP>
__device__ __forceinline__ void vis_res_block(WP blob, size_t base, int C, float* x, int S, float* t1,
                                              float* t2, int lane) {
    const int n = S * S;
    const size_t cc9 = (size_t)C * C * 9;
    mzs_vis_bn bn;
    bn.w = base + 2 * cc9, bn.b = bn.w + C, bn.mean = bn.w + 2 * C, bn.var = bn.w + 3 * C;
    /* one copy of the stage's code per block (a loop, not three inlined stages); t1 holds the map with its border */
    const float* src = x;
#pragma nounroll
    for (int i = 0; i < 3; ++i) {
        const size_t w_off = base + (i == 1 ? cc9 : 0);
        if (C == kVisC) {
            vis_bn_relu_pad<kVisC>(blob, bn, src, S, t1, lane);
            vis_conv3_pad<kVisC, kHoist>(blob, w_off, t1, S, t2, lane);
        } else {
            vis_bn_relu_pad<kVisC / 2>(blob, bn, src, S, t1, lane);
            vis_conv3_pad<kVisC / 2, kHoist>(blob, w_off, t1, S, t2, lane);
        }
        src = t2;
    }
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

/* The Linear layers' weights as the kernels read them: a store of its own (mzs_set_weights builds it, Model::w
 * points at it) holding, head after head (0: reward, 1: value, 2: policy), the matrices in, mid (if L > 0), out,
 * each from an offset that is a multiple of four floats.  A matrix W [nout][nin] is laid out in quads of rows of
 * W^T: element (k, j), k < 4 (nin / 4), at ((k / 4) nout + j) 4 + k % 4 — lane j fetches four consecutive k with
 * one 16-byte load, the 64 lanes of a load touch 1 KB of neighbouring addresses — and the last nin % 4 rows of
 * W^T behind them as [k][j]. */
__host__ __device__ inline size_t vis_r4(size_t n) { return (n + 3) & ~(size_t)3; }
struct VisLin {
    size_t in_w, mid_w, out_w;
};
__host__ __device__ inline size_t vis_lin_head_floats(int H, int L, int n_out) {
    return vis_r4((size_t)H * MZS_VIS_HIDDEN) + (L > 0 ? vis_r4((size_t)H * H) : 0) + vis_r4((size_t)n_out * H);
}
__host__ __device__ inline VisLin vis_lin_offsets(int head, int H, int L, int Sd, int A) {
    (void)A;
    VisLin v;
    v.in_w = (size_t)head * vis_lin_head_floats(H, L, Sd); /* heads 0 and 1 both end in Sd outputs */
    v.mid_w = v.in_w + vis_r4((size_t)H * MZS_VIS_HIDDEN);
    v.out_w = v.mid_w + (L > 0 ? vis_r4((size_t)H * H) : 0);
    return v;
}
__host__ __device__ inline size_t vis_lin_total(int H, int L, int Sd, int A) {
    return 2 * vis_lin_head_floats(H, L, Sd) + vis_lin_head_floats(H, L, A);
}

/* y = [relu](b + W x), contract "dot" (even / odd chains: rows 4q, 4q + 2 feed e, rows 4q + 1, 4q + 3 feed o);
 * W in the quad layout above.  NB rows (NB / 4 loads of 16 bytes) are requested ahead of the chains that consume
 * them (a trip to L2 per batch; NB = 64: three for the 147-wide layer, one for the others; NB = 32 where four
 * waves share a SIMD and 128 registers); a partial last batch reads clamped quads and skips them by a select. */
template <int NB>
__device__ __forceinline__ void vis_linear(const float* __restrict__ W, const float* __restrict__ b, const float* x,
                                           int nout, int nin, bool relu, float* y, int lane) {
    constexpr int QB = NB / 4, QT = 8;
    const int Q = nin >> 2;
    for (int j = lane; j < nout; j += kWave) {
        const float4* __restrict__ wq = reinterpret_cast<const float4*>(W) + j;
        float e = b[j], o = 0.0f;
        int q = 0;
        for (; q + QB <= Q; q += QB) {
            float4 wv[QB], xq[QB];
#pragma unroll
            for (int u = 0; u < QB; ++u) wv[u] = wq[(size_t)(q + u) * nout];
#pragma unroll
            for (int u = 0; u < QB; ++u) xq[u] = *reinterpret_cast<const float4*>(x + 4 * (q + u));
            __builtin_amdgcn_sched_barrier(0); /* every request of the batch leaves before the chains start */
#pragma unroll
            for (int u = 0; u < QB; ++u) {
                e = mzs_fma(wv[u].x, xq[u].x, e);
                o = mzs_fma(wv[u].y, xq[u].y, o);
                e = mzs_fma(wv[u].z, xq[u].z, e);
                o = mzs_fma(wv[u].w, xq[u].w, o);
            }
        }
        for (; q < Q; q += QT) {
            float4 wv[QT], xq[QT];
#pragma unroll
            for (int u = 0; u < QT; ++u) {
                const int qq = q + u < Q ? q + u : Q - 1;
                wv[u] = wq[(size_t)qq * nout];
                xq[u] = *reinterpret_cast<const float4*>(x + 4 * qq);
            }
#pragma unroll
            for (int u = 0; u < QT; ++u) {
                const float e2 = mzs_fma(wv[u].x, xq[u].x, e), o2 = mzs_fma(wv[u].y, xq[u].y, o);
                const float e3 = mzs_fma(wv[u].z, xq[u].z, e2), o3 = mzs_fma(wv[u].w, xq[u].w, o2);
                e = q + u < Q ? e3 : e;
                o = q + u < Q ? o3 : o;
            }
        }
        /* the last nin % 4 rows: [k][j] behind the quads; row 4Q feeds e, 4Q + 1 o, 4Q + 2 e */
        const int nt = nin - 4 * Q;
        if (nt > 0) {
            const float* __restrict__ wt = W + (size_t)4 * Q * nout + j;
            float tw[3], tx[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int uu = u < nt ? u : nt - 1;
                tw[u] = wt[(size_t)uu * nout], tx[u] = x[4 * Q + uu];
            }
            const float e2 = mzs_fma(tw[0], tx[0], e), o2 = mzs_fma(tw[1], tx[1], o);
            e = e2;
            o = nt > 1 ? o2 : o;
            const float e3 = mzs_fma(tw[2], tx[2], e);
            e = nt > 2 ? e3 : e;
        }
        const float v = e + o;
        y[j] = relu && !(v > 0.0f) ? 0.0f : v;
    }
    wave_sync();
}

/* 1x1 convolution (+bias) of x [Ci][49] -> flatten -> Linear, relu, (Linear, relu) x L, Linear: logits.
 * cw: the cached parameters (conv_w, conv_b are blob offsets into them) */
template <int NB, class WP>
__device__ __forceinline__ void vis_head_mlp(const Model& m, const float* __restrict__ blob, WP cw,
                                             size_t conv_w, size_t conv_b, int Ci, size_t fc_base, int head, int n_out,
                                             const float* x, float* logits, float* t1, float* t2, int lane) {
    constexpr int CM = kVisC + 1;
    for (int co = 0; co < kVisC; ++co) {
        float wv[CM];
#pragma unroll
        for (int ci = 0; ci < CM; ++ci) wv[ci] = cw[conv_w + co * Ci + (ci < Ci ? ci : 0)];
        const float bias = cw[conv_b + co];
        for (int i = lane; i < kVisPix; i += kWave) {
            float xv[CM]; /* the taps' loads in flight together, then the chain */
#pragma unroll
            for (int ci = 0; ci < CM; ++ci) xv[ci] = x[(ci < Ci ? ci : 0) * kVisPix + i];
            float acc = bias;
#pragma unroll
            for (int ci = 0; ci < CM; ++ci) {
                const float a2 = mzs_fma(wv[ci], xv[ci], acc);
                acc = ci < Ci ? a2 : acc;
            }
            t1[co * kVisPix + i] = acc;
        }
    }
    wave_sync();
    /* Linear 147 -> H, (H -> H) x L with one tensor, H -> n_out: one copy of vis_linear for all of them */
    /* offsets in mzs_vis_mlp_layout's order from fc_base (= its in_w), as plain scalars */
    const size_t H = (size_t)m.H, in_b = fc_base + H * kVisHidden, mid_w = in_b + H, mid_b = mid_w + (m.L > 0 ? H * H : 0);
    const size_t out_w = mid_b + (m.L > 0 ? H : 0), out_b = out_w + (size_t)n_out * H;
    /* the 147-wide layer has a copy of vis_linear of its own (constant fan-in: its partial last batch is
     * resolved at compile time), the H-wide ones share the second */
    const VisLin lw = vis_lin_offsets(head, m.H, m.L, m.Sd, m.A); /* the weights: offsets into the quad store */
    vis_linear<NB>(m.w + lw.in_w, blob + in_b, t1, m.H, kVisHidden, true, t2, lane);
    const float* src = t2;
    float* dst = t1;
#pragma nounroll
    for (int l = 1; l < m.L + 2; ++l) {
        const bool last = l == m.L + 1;
        vis_linear<NB>(m.w + (last ? lw.out_w : lw.mid_w), blob + (last ? out_b : mid_b), src, last ? n_out : m.H, m.H, !last,
                   last ? logits : dst, lane);
        float* t = const_cast<float*>(src);
        src = dst;
        dst = t;
    }
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
    const float* blob = L.vc; /* every parameter of the representation is in the cache, at its blob offset */
    float *x = L.ob, *t1 = L.xa, *t2 = L.xb;
    vis_conv3_t<kVisC, 4>(blob + o.r_conv_in, obs_g, MZS_VIS_SIDE, kVisC / 2, 2, x, lane); /* 1 x 49 x 49 */
    /* the eight residual blocks through ONE copy of the block's code (the instruction cache again):
     * res_in x 2 | conv_out -> 3 x 25 x 25 | res_out x 2 | pool -> 13 x 13 | res_out x 3 | pool -> 7 x 7 | res */
#pragma unroll
    for (int blk = 0; blk < 8; ++blk) {
        if (blk == 2 || blk == 4 || blk == 7) {
            if (blk == 2)
                vis_conv3(blob + o.r_conv_out, x, kVisC / 2, MZS_VIS_S1, kVisC, 2, t1, lane);
            else
                vis_avgpool(x, kVisC, blk == 4 ? MZS_VIS_S2 : MZS_VIS_S3, t1, lane);
            float* t = x;
            x = t1, t1 = t;
        }
        const size_t base = blk < 2 ? o.r_res_in.conv_a : (blk == 7 ? o.r_res.conv_a : o.r_res_out.conv_a);
        const int S = blk < 2 ? MZS_VIS_S1 : (blk < 4 ? MZS_VIS_S2 : (blk < 7 ? MZS_VIS_S3 : MZS_VIS_S4));
        vis_res_block<true>(blob, base, blk < 2 ? kVisC / 2 : kVisC, x, S, t1, t2, lane);
    }
    for (int i = lane; i < kVisHidden; i += kWave) {
        const float v = x[i];
        L.xh[i] = v;
        if (gdst) gdst[i] = v;
    }
    wave_sync();
}

/* hidden state in L.xh + action -> reward (returned, uniform); next hidden state to L.xh and gdst */
template <int NB = 64>
__device__ __forceinline__ float vis_dynamics(const Model& m, const WaveLds& L, int action, float* gdst, int lane) {
    const mzs_vision_offsets o = vis_offsets(m);
    const float* __restrict__ blob = m.blob;
    float *x = L.ob, *t1 = L.xa, *t2 = L.xb, *logits = L.ob + 256;
    /* cat(state, action plane): the plane holds (a + 1) / A (muzero_model.py:515) */
    const float pv = (float)(action + 1) / (float)m.A;
    for (int i = lane; i < kVisPix; i += kWave) L.xh[kVisHidden + i] = pv;
    wave_sync();
    /* the convolutions' and batchnorms' parameters straight from the blob through the constant address space:
     * uniform addresses, so scalar loads, and the weights reach the fma chains as scalar operands (the
     * simulations of a search; the representation keeps its LDS cache) */
    const VisConstPtr cw = (VisConstPtr)m.blob;
    vis_conv3_bn<kVisC + 1, kVisC>(cw, o.d_conv, o.d_bn, L.xh, x, lane);
    for (int l = 0; l < m.L; ++l) vis_res_block(cw, o.d_res.conv_a, kVisC, x, MZS_VIS_S4, t1, t2, lane);
    vis_head_mlp<NB>(m, blob, cw, o.d_conv_rew, o.d_conv_rew_b, kVisC + 1, o.d_fc.in_w, 0, m.Sd, L.xh, logits, t1, t2, lane);
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
template <int NB = 64>
__device__ __forceinline__ float vis_prediction(const Model& m, const WaveLds& L, Pair* policy, int lane) {
    const mzs_vision_offsets o = vis_offsets(m);
    const float* __restrict__ blob = m.blob;
    float *x = L.ob, *t1 = L.xa, *t2 = L.xb, *logits = L.ob + 256;
    for (int i = lane; i < kVisHidden; i += kWave) x[i] = L.xh[i];
    wave_sync();
    const VisConstPtr cw = (VisConstPtr)m.blob;
    for (int l = 0; l < m.L; ++l) vis_res_block(cw, o.p_res.conv_a, kVisC, x, MZS_VIS_S4, t1, t2, lane);
    float value = 0.0f;
#pragma nounroll
    for (int hd = 0; hd < 2; ++hd) { /* value head, then policy head: one copy of the head's code */
        const size_t cv = hd ? o.p_conv_pol : o.p_conv_val, cb = hd ? o.p_conv_pol_b : o.p_conv_val_b;
        vis_head_mlp<NB>(m, blob, cw, cv, cb, kVisC, hd ? o.p_p.in_w : o.p_v.in_w, 1 + hd, hd ? m.A : m.Sd, x, logits, t1, t2, lane);
        if (hd == 0) {
            value = seg_support_scalar(logits, m.Sd, lane);
            wave_sync();
        }
    }
    *policy = seg_softmax(seg_load(logits, m.A, lane));
    wave_sync();
    return value;
}

}  // namespace mzs
#endif
