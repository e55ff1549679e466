/*
 * mzs_device.h — wave-level device functions of the MuZero search path (gfx950).
 *
 * One 64-lane wavefront owns one game.  The game's tree lives in that wave's LDS
 * slice as 16-byte node records; lanes 0..A-1 score one child each, lane pairs
 * hold the (2i, 2i+1) elements of a logit vector for softmax / min-max reductions,
 * lane j owns output neuron j of a dense layer.  No cross-wave communication.
 *
 * Reference code restated (file:line in /root/reference):
 *   wave_select   monte_carlo_tree_search.py:155-177  ucb_score / select_child / descent
 *   wave_backup   monte_carlo_tree_search.py:203-209  back_propagate_and_update_min_max_bound
 *   wave_expand   monte_carlo_tree_search.py:186-200
 *   model_*       neural_network_mlp_model.py:43-44, :86-88, :127-128, :135-143;
 *                 muzero_model.py:571-587, :877-879, :898-903
 * Arithmetic contract: include/mzs_numeric.h.
 */
#ifndef MZS_DEVICE_H
#define MZS_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mzs_layout.h"
#include "../../include/mzs_numeric.h"

namespace mzs {

constexpr int kWave = 64;

/* One dense layer in the device weight store: W^T as [rows][n_pad] (lane-contiguous
 * output neurons), bias [n_pad]; padding columns are zero. */
struct Layer {
    int w_off;  /* float offset of W^T                               */
    int b_off;  /* float offset of bias                              */
    int k;      /* fan-in accumulated by the fma chain               */
    int n_pad;  /* padded fan-out, multiple of 64                    */
};

/* One head of the 'lstm_model' (neural_network_lstm_model.py:33-47, :63-77): Linear, then an
 * L-layer nn.LSTM stepped once from zero state.  gate[l] is the fused (i | g | o) projection of
 * layer l, each gate padded to out_pad columns, bias = bias_ih; bhh_off[l] = bias_hh in the same
 * column order (added after the chain, as torch adds the two Linear results). */
constexpr int kLstmMaxLayers = MZS_LSTM_MAX_LAYERS;
struct LstmHead {
    Layer lin;
    Layer gate[kLstmMaxLayers];
    int bhh_off[kLstmMaxLayers];
    int n_out, out_pad;
};

struct Model {
    int obs_dim, A, Sd, H, L;
    int kind;       /* MZS_MODEL_MLP / MZS_MODEL_LSTM / MZS_MODEL_VISION (include/mzs_layout.h) */
    int bf16;       /* the handle's precision: MZS_PRECISION_BF16 = contract "dot_bf16" (w then holds bf16-rounded
                       weights), MZS_PRECISION_BF16X3 = "dot_bf16x3" (w = the fp32 store, split on use) */
    int seg_state;  /* offset of the next-state segment in the dynamics heads (reward at 0) */
    int seg_value;  /* offset of the value segment in the prediction heads (policy at 0)     */
    Layer r_in, r_mid, r_out, d_in, d_mid, d_heads, p_in, p_mid, p_heads;
    LstmHead l_rew, l_st, l_pol, l_val; /* LSTM model only (r_in = its representation Linear) */
    const float* w; /* device weight store */
    int hid;           /* floats of one hidden state: Sd, or 147 for the vision model */
    const float* blob; /* master weights in blob order (what the vision model's functions read) */
};

/* 16-byte node record (replaces monte_carlo_tree_search.py:6-20 Node) */
struct __attribute__((aligned(16))) Node {
    int visit;
    float vsum;
    float prior;
    float reward;
};

/* Per-wave LDS slice */
struct WaveLds {
    Node* nodes;          /* [n_nodes]                                         */
    unsigned short* cb;   /* [n_nodes] first child index, 0 = leaf              */
    unsigned short* path; /* [S+2] search_path                                  */
    float* xh;            /* hidden state / observation (layer input)           */
    float* xa;            /* trunk activations, ping                            */
    float* xb;            /* trunk activations, pong                            */
    float* ob;            /* head outputs (logit segments)                      */
    float* vc;            /* vision model: cached convolution / batchnorm parameters */
};

__device__ __forceinline__ void wave_sync() {
    /* LDS operations of one wave execute in order; this only stops the compiler
     * from moving LDS accesses of different lanes across the phase boundary. */
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ float readlane_f32(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

/* ---- DPP cross-lane moves (row = 16 lanes): single VALU instructions, no LDS ---- */
template <int kCtrl>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(
        __builtin_amdgcn_update_dpp(0, __float_as_int(v), kCtrl, 0xF, 0xF, false));
}
constexpr int kDppQuadXor1 = 0xB1;      /* quad_perm [1,0,3,2] */
constexpr int kDppQuadXor2 = 0x4E;      /* quad_perm [2,3,0,1] */
constexpr int kDppRowHalfMirror = 0x141;
constexpr int kDppRowMirror = 0x140;
/* After these four in-row steps every lane of a 16-lane row holds the row's result.
 * Contract "sum": adjacent-pair tree (a+b == b+a bitwise, so the mirrors stand in for
 * the xor-4 / xor-8 exchanges). */
__device__ __forceinline__ float row_tree_sum(float v) {
    v = v + dpp_mov<kDppQuadXor1>(v);
    v = v + dpp_mov<kDppQuadXor2>(v);
    v = v + dpp_mov<kDppRowHalfMirror>(v);
    v = v + dpp_mov<kDppRowMirror>(v);
    return v;
}

__device__ __forceinline__ float row_max(float v) {
    float t;
    t = dpp_mov<kDppQuadXor1>(v);
    v = t > v ? t : v;
    t = dpp_mov<kDppQuadXor2>(v);
    v = t > v ? t : v;
    t = dpp_mov<kDppRowHalfMirror>(v);
    v = t > v ? t : v;
    t = dpp_mov<kDppRowMirror>(v);
    v = t > v ? t : v;
    return v;
}

/* uniform results of the two 32-lane halves (rows 0+1 and rows 2+3) */
struct Halves {
    float lo, hi;
};

__device__ __forceinline__ Halves halves_tree_sum(float v) {
    v = row_tree_sum(v);
    Halves h;
    h.lo = readlane_f32(v, 0) + readlane_f32(v, 16);
    h.hi = readlane_f32(v, 32) + readlane_f32(v, 48);
    return h;
}

__device__ __forceinline__ Halves halves_max(float v) {
    v = row_max(v);
    const float r0 = readlane_f32(v, 0), r1 = readlane_f32(v, 16);
    const float r2 = readlane_f32(v, 32), r3 = readlane_f32(v, 48);
    Halves h;
    h.lo = r1 > r0 ? r1 : r0;
    h.hi = r3 > r2 ? r3 : r2;
    return h;
}

/* order-free reductions over the wave: result uniform */
__device__ __forceinline__ float wave_max(float v) {
    const Halves h = halves_max(v);
    return h.hi > h.lo ? h.hi : h.lo;
}

__device__ __forceinline__ float wave_min(float v) { return -wave_max(-v); }

/* contract "sum" over 64 lanes: adjacent-pair tree; lane i already holds
 * v[2i] + v[2i+1]; the last level adds the two half sums. */
__device__ __forceinline__ float wave_tree_sum(float v) {
    const Halves h = halves_tree_sum(v);
    return h.lo + h.hi;
}

/* ---- the bf16 contracts of the any-shape kernels, on the matrix core, ONE game per wavefront ----
 * v_mfma_f32_4x4x4_16B_bf16 is sixteen independent 4 x 4 x 4 products: in block b, lane 4 b + i supplies row i
 * of A and lane 4 b + j column j of B, and lane 4 b + j receives column j of D.  Every lane passes its own
 * neuron's four weights as its A row and the SAME four inputs as its B column; then D[i][j] = the dot product of
 * neuron 4 b + i whatever j, and a lane finds its own neuron's in element lane & 3 of its result.  Three quarters
 * of the instruction are redundant — it still replaces ~40 VALU instructions of the software restatement
 * (mzs_dot4_bf16: exact products, an exact sum, one rounding), which made these paths 5-100 x slower than fp32. */
typedef short dev_s16x4 __attribute__((ext_vector_type(4)));
typedef float dev_f32x4 __attribute__((ext_vector_type(4)));
typedef float dev_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 dev_bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned dev_u32x2 __attribute__((ext_vector_type(2)));

/* two floats -> packed bfloat16 pair, round to nearest even (v_cvt_pk_bf16_f32 = mzs_f2bf16 on each) */
__device__ __forceinline__ unsigned dev_pack_bf16(float lo, float hi) {
    const dev_f32x2 v = {lo, hi};
    const dev_bf16x2 r = __builtin_convertvector(v, dev_bf16x2);
    unsigned u;
    __builtin_memcpy(&u, &r, 4);
    return u;
}
__device__ __forceinline__ dev_s16x4 dev_s16(unsigned a, unsigned b) {
    const dev_u32x2 v = {a, b};
    dev_s16x4 r;
    __builtin_memcpy(&r, &v, 8);
    return r;
}
/* mzs_bf16_split on two floats: packed hi halves, packed lo halves */
__device__ __forceinline__ void dev_split_pack(float a, float b, unsigned& hi, unsigned& lo) {
    hi = dev_pack_bf16(a, b);
    const float ah = __uint_as_float(hi << 16), bh = __uint_as_float(hi & 0xffff0000u);
    lo = dev_pack_bf16(a - ah, b - bh);
}
/* mzs_dot4_bf16 of this lane's four weights with the wave's four inputs (every lane must be active) */
__device__ __forceinline__ float dev_dot4_mfma(const dev_s16x4 w, const dev_s16x4 x, int lane) {
    const dev_f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    const dev_f32x4 d = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(w, x, zero, 0, 0, 0);
    const int i = lane & 3;
    const float lo = i & 1 ? d[1] : d[0], hi = i & 1 ? d[3] : d[2];
    return i & 2 ? hi : lo;
}

/* y = act(bias + W x): lane j owns neuron j (and j+64, ...); x is broadcast from LDS.
 * extra_row >= 0 adds that row of W^T after the chain (the one-hot action column of
 * the dynamics input, neural_network_mlp_model.py:87 + muzero_model.py:502). */
template <bool kElu>
__device__ __forceinline__ void wave_dense_bf16(const float* __restrict__ w, const Layer ly, const float* x,
                                                float* out, int extra_row, int lane) {
    /* contract "dot_bf16" of the any-shape kernels.  The store holds the weights already rounded to bfloat16
     * (as floats): their high halves. */
    const int ng = (ly.k + 3) / 4;
    for (int j0 = 0; j0 < ly.n_pad; j0 += kWave) {
        const int j = j0 + lane;
        const float* __restrict__ wt = w + ly.w_off + j;
        float e = w[ly.b_off + j], o = 0.0f;
        for (int g = 0; g < ng; ++g) {
            unsigned wb[4];
            float xv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = 4 * g + q;
                const bool in = k < ly.k;
                wb[q] = in ? (unsigned)mzs_f2bits(wt[(size_t)k * ly.n_pad]) : 0u;
                xv[q] = in ? x[k] : 0.0f;
            }
            const dev_s16x4 w4 = dev_s16((wb[0] >> 16) | (wb[1] & 0xffff0000u), (wb[2] >> 16) | (wb[3] & 0xffff0000u));
            const dev_s16x4 x4 = dev_s16(dev_pack_bf16(xv[0], xv[1]), dev_pack_bf16(xv[2], xv[3]));
            const float d = dev_dot4_mfma(w4, x4, lane);
            if (g & 1)
                o = g == 1 ? d : o + d;
            else
                e = e + d;
        }
        float acc = e + o;
        if (extra_row >= 0) acc = acc + wt[(size_t)extra_row * ly.n_pad];
        if (kElu) acc = mzs_elu(acc);
        out[j] = acc;
    }
    wave_sync();
}

/* contract "dot_bf16x3" of the any-shape kernels: weights (from the fp32 store) and inputs split into hi + lo
 * bfloat16 pairs, three products per group, four chains (include/mzs_numeric.h) */
template <bool kElu>
__device__ __forceinline__ void wave_dense_bf16x3(const float* __restrict__ w, const Layer ly, const float* x,
                                                  float* out, int extra_row, int lane) {
    const int ng = (ly.k + 3) / 4;
    for (int j0 = 0; j0 < ly.n_pad; j0 += kWave) {
        const int j = j0 + lane;
        const float* __restrict__ wt = w + ly.w_off + j;
        float e = w[ly.b_off + j], o = 0.0f, s = 0.0f, t = 0.0f;
        for (int g = 0; g < ng; ++g) {
            float wv[4], xv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = 4 * g + q;
                const bool in = k < ly.k;
                wv[q] = in ? wt[(size_t)k * ly.n_pad] : 0.0f;
                xv[q] = in ? x[k] : 0.0f;
            }
            unsigned wh0, wl0, wh1, wl1, xh0, xl0, xh1, xl1;
            dev_split_pack(wv[0], wv[1], wh0, wl0);
            dev_split_pack(wv[2], wv[3], wh1, wl1);
            dev_split_pack(xv[0], xv[1], xh0, xl0);
            dev_split_pack(xv[2], xv[3], xh1, xl1);
            const dev_s16x4 wh = dev_s16(wh0, wh1), wl = dev_s16(wl0, wl1), xh = dev_s16(xh0, xh1), xl = dev_s16(xl0, xl1);
            const float a = dev_dot4_mfma(wh, xh, lane), b = dev_dot4_mfma(wh, xl, lane), c = dev_dot4_mfma(wl, xh, lane);
            if (g & 1)
                o = g == 1 ? a : o + a;
            else
                e = e + a;
            s = g == 0 ? b : s + b;
            t = g == 0 ? c : t + c;
        }
        float acc = (e + o) + (s + t);
        if (extra_row >= 0) acc = acc + wt[(size_t)extra_row * ly.n_pad];
        if (kElu) acc = mzs_elu(acc);
        out[j] = acc;
    }
    wave_sync();
}

template <bool kElu>
__device__ __forceinline__ void wave_dense(const float* __restrict__ w, const Layer ly,
                                           const float* x, float* out, int extra_row, int lane, int bf16 = 0) {
    if (bf16 == MZS_PRECISION_BF16X3) {
        wave_dense_bf16x3<kElu>(w, ly, x, out, extra_row, lane);
        return;
    }
    if (bf16) {
        wave_dense_bf16<kElu>(w, ly, x, out, extra_row, lane);
        return;
    }
    for (int j0 = 0; j0 < ly.n_pad; j0 += kWave) {
        const int j = j0 + lane;
        const float* __restrict__ wt = w + ly.w_off + j;
        float acc = w[ly.b_off + j], odd = 0.0f;
        int k = 0;
        for (; k + 4 <= ly.k; k += 4) {
            const float4 xv = *reinterpret_cast<const float4*>(x + k);
            const float w0 = wt[(size_t)(k + 0) * ly.n_pad];
            const float w1 = wt[(size_t)(k + 1) * ly.n_pad];
            const float w2 = wt[(size_t)(k + 2) * ly.n_pad];
            const float w3 = wt[(size_t)(k + 3) * ly.n_pad];
            acc = mzs_fma(w0, xv.x, acc);
            odd = mzs_fma(w1, xv.y, odd);
            acc = mzs_fma(w2, xv.z, acc);
            odd = mzs_fma(w3, xv.w, odd);
        }
        for (; k < ly.k; ++k) {
            if (k & 1)
                odd = mzs_fma(wt[(size_t)k * ly.n_pad], x[k], odd);
            else
                acc = mzs_fma(wt[(size_t)k * ly.n_pad], x[k], acc);
        }
        acc = acc + odd;
        if (extra_row >= 0) acc = acc + wt[(size_t)extra_row * ly.n_pad];
        if (kElu) acc = mzs_elu(acc);
        out[j] = acc;
    }
    wave_sync();
}

/* tied mid layer applied L times (neural_network_mlp_model.py:38-39); returns the
 * buffer holding the result */
__device__ __forceinline__ float* wave_mid_layers(const Model& m, const Layer ly, float* in,
                                                  float* other, int lane) {
    for (int l = 0; l < m.L; ++l) {
        wave_dense<true>(m.w, ly, in, other, -1, lane, m.bf16);
        float* t = in;
        in = other;
        other = t;
    }
    return in;
}

/* elements (2*lane, 2*lane+1) of a logit segment of length n <= 128 */
struct Pair {
    float a, b;
    bool va, vb;
};

__device__ __forceinline__ Pair seg_load(const float* seg, int n, int lane) {
    Pair p;
    const int i = 2 * lane;
    p.va = i < n;
    p.vb = i + 1 < n;
    p.a = p.va ? seg[i] : 0.0f;
    p.b = p.vb ? seg[i + 1] : 0.0f;
    return p;
}

/* softmax probabilities of the pair (contract exp + sum) */
__device__ __forceinline__ Pair seg_softmax(const Pair q) {
    const float ninf = -__builtin_inff();
    float m = wave_max(fmaxf(q.va ? q.a : ninf, q.vb ? q.b : ninf));
    Pair e = q;
    e.a = q.va ? mzs_expf(q.a - m) : 0.0f;
    e.b = q.vb ? mzs_expf(q.b - m) : 0.0f;
    const float Z = wave_tree_sum(e.a + e.b);
    const float rz = 1.0f / Z;   /* contract: one reciprocal, corrected products (mzs_div_r2) */
    e.a = mzs_div_r2(e.a, Z, rz);
    e.b = mzs_div_r2(e.b, Z, rz);
    return e;
}

/* inverse_transform_with_support (muzero_model.py:571-587) of a logit segment */
__device__ __forceinline__ float seg_support_scalar(const float* seg, int n, int lane) {
    const Pair p = seg_softmax(seg_load(seg, n, lane));
    const int half = n / 2, i = 2 * lane;
    const float ta = p.va ? (float)(i - half) * p.a : 0.0f;
    const float tb = p.vb ? (float)(i + 1 - half) * p.b : 0.0f;
    const float y = wave_tree_sum(ta + tb);
    return mzs_inverse_value_transform(y);
}

/* scale_to_bound_action (neural_network_mlp_model.py:135-143): normalised row to
 * LDS `dst` and, if not null, to global `gdst` */
__device__ __forceinline__ void seg_scale_to_bound(const float* seg, int n, float* dst,
                                                   float* gdst, int lane) {
    const Pair q = seg_load(seg, n, lane);
    const float pinf = __builtin_inff();
    const float mn = wave_min(fminf(q.va ? q.a : pinf, q.vb ? q.b : pinf));
    const float mx = wave_max(fmaxf(q.va ? q.a : -pinf, q.vb ? q.b : -pinf));
    const float sc = mzs_bound_scale(mn, mx), rs = 1.0f / sc;
    const int i = 2 * lane;
    if (q.va) {
        const float y = mzs_div_r2(q.a - mn, sc, rs);
        dst[i] = y;
        if (gdst) gdst[i] = y;
    }
    if (q.vb) {
        const float y = mzs_div_r2(q.b - mn, sc, rs);
        dst[i + 1] = y;
        if (gdst) gdst[i + 1] = y;
    }
    wave_sync();
}

/* One LSTM head on the row x (LDS): h of the last layer, n_out values, in the returned LDS
 * buffer (L.xa or L.xb).  Lane j owns unit j (and j + 64): the three gate chains of a unit are
 * three columns of one fused dense layer, then the cell step is lane-local. */
__device__ __forceinline__ float* lstm_head(const Model& m, const LstmHead& hd, const WaveLds& L,
                                            const float* x, int extra_row, int lane) {
    wave_dense<false>(m.w, hd.lin, x, L.xa, extra_row, lane, m.bf16);
    float *cur = L.xa, *nxt = L.xb;
    for (int l = 0; l < m.L; ++l) {
        wave_dense<false>(m.w, hd.gate[l], cur, L.ob, -1, lane, m.bf16);
        const float* bh = m.w + hd.bhh_off[l];
        for (int j = lane; j < hd.n_out; j += kWave) {
            const float gi = L.ob[j] + bh[j];
            const float gg = L.ob[hd.out_pad + j] + bh[hd.out_pad + j];
            const float go = L.ob[2 * hd.out_pad + j] + bh[2 * hd.out_pad + j];
            nxt[j] = mzs_lstm_cell0(gi, gg, go);
        }
        wave_sync();
        float* t = cur;
        cur = nxt;
        nxt = t;
    }
    return cur;
}

/* copy n values LDS -> LDS (and global if not null) */
__device__ __forceinline__ void seg_copy(const float* src, int n, float* dst, float* gdst, int lane) {
    for (int j = lane; j < n; j += kWave) {
        const float v = src[j];
        dst[j] = v;
        if (gdst) gdst[j] = v;
    }
    wave_sync();
}

}  // namespace mzs
#include "mzs_vision.h"
namespace mzs {

/* Representation_function.forward: input in L.xh[0..obs) (vision model: the image at obs_g, global),
 * hidden state to L.xh and gdst */
/* kVision: the kernels are instantiated once for the vision model and once for the others, so that
 * the convolutions' register appetite (dozens of taps in flight) does not cost the MLP / LSTM
 * instantiation its occupancy */
template <bool kVision>
__device__ __forceinline__ void model_representation(const Model& m, const WaveLds& L, const float* obs_g,
                                                     float* gdst, int lane) {
    if constexpr (kVision) {
        vis_representation(m, L, obs_g, gdst, lane);
        return;
    }
    if (m.kind == MZS_MODEL_LSTM) { /* one Linear (neural_network_lstm_model.py:19-21) */
        wave_dense<false>(m.w, m.r_in, L.xh, L.ob, -1, lane, m.bf16);
        seg_copy(L.ob, m.Sd, L.xh, gdst, lane);
        return;
    }
    wave_dense<true>(m.w, m.r_in, L.xh, L.xa, -1, lane, m.bf16);
    float* t = wave_mid_layers(m, m.r_mid, L.xa, L.xb, lane);
    wave_dense<false>(m.w, m.r_out, t, L.ob, -1, lane, m.bf16);
    seg_scale_to_bound(L.ob, m.Sd, L.xh, gdst, lane);
}

/* Dynamics_function.forward on L.xh + action: reward returned (uniform), next hidden
 * state to L.xh and gdst */
template <bool kVision, int kVisBatch = 64>
__device__ __forceinline__ float model_dynamics(const Model& m, const WaveLds& L, int action,
                                                float* gdst, int lane) {
    if constexpr (kVision) return vis_dynamics<kVisBatch>(m, L, action, gdst, lane);
    if (m.kind == MZS_MODEL_LSTM) {
        const float* r = lstm_head(m, m.l_rew, L, L.xh, m.Sd + action, lane);
        const float reward = seg_support_scalar(r, m.Sd, lane);
        wave_sync();
        const float* s = lstm_head(m, m.l_st, L, L.xh, m.Sd + action, lane);
        seg_copy(s, m.Sd, L.xh, gdst, lane);
        return reward;
    }
    wave_dense<true>(m.w, m.d_in, L.xh, L.xa, m.Sd + action, lane, m.bf16);
    float* t = wave_mid_layers(m, m.d_mid, L.xa, L.xb, lane);
    wave_dense<false>(m.w, m.d_heads, t, L.ob, -1, lane, m.bf16);
    const float reward = seg_support_scalar(L.ob, m.Sd, lane);
    seg_scale_to_bound(L.ob + m.seg_state, m.Sd, L.xh, gdst, lane);
    return reward;
}

/* Prediction_function.forward on L.xh: softmax policy as lane pairs, value (uniform) */
template <bool kVision, int kVisBatch = 64>
__device__ __forceinline__ float model_prediction(const Model& m, const WaveLds& L, Pair* policy,
                                                  int lane) {
    if constexpr (kVision) return vis_prediction<kVisBatch>(m, L, policy, lane);
    if (m.kind == MZS_MODEL_LSTM) {
        const float* pl = lstm_head(m, m.l_pol, L, L.xh, -1, lane);
        *policy = seg_softmax(seg_load(pl, m.A, lane));
        wave_sync();
        const float* vl = lstm_head(m, m.l_val, L, L.xh, -1, lane);
        return seg_support_scalar(vl, m.Sd, lane);
    }
    wave_dense<true>(m.w, m.p_in, L.xh, L.xa, -1, lane, m.bf16);
    float* t = wave_mid_layers(m, m.p_mid, L.xa, L.xb, lane);
    wave_dense<false>(m.w, m.p_heads, t, L.ob, -1, lane, m.bf16);
    *policy = seg_softmax(seg_load(L.ob, m.A, lane));
    return seg_support_scalar(L.ob + m.seg_value, m.Sd, lane);
}

/* ------------------------------------------------------------------ search */

struct SearchConst {
    int A, S, n_nodes;
    float g32;            /* discount rounded to fp32 (python float * np.float32)    */
    const double* pbc;    /* device [S+2] sqrt(n)*pb_c(n)                             */
    const double* noise;  /* this game's tie-break draws                             */
    long long noise_cap;  /* draws available                                         */
};

struct SearchState {
    float mn, mx;          /* MinMaxStats                                            */
    double root_prior;     /* lane a: prior of root child a after noise (f64)        */
    long long consumed;    /* tie-break draws consumed                               */
    long long win_base;    /* first draw held in the register window                 */
    double win;            /* lane l holds draw win_base + l                         */
    long long path_total;
    int status;
};

__device__ __forceinline__ void noise_refill(const SearchConst& c, SearchState& s, int lane) {
    s.win_base = s.consumed;
    const long long i = s.win_base + lane;
    s.win = i < c.noise_cap ? c.noise[i] : 0.0;
}

/* Descent from the root to an unexpanded node. Returns the leaf; *depth = number of
 * selections, *action = last action, *parent_base = child_base of the leaf's parent. */
__device__ __forceinline__ int wave_select(const SearchConst& c, SearchState& s, const WaveLds& L,
                                           int* depth, int* action, int* parent_base, int lane) {
    int node = 0, d = 0, act = 0, base = 0;
    int cbn = __builtin_amdgcn_readfirstlane((int)L.cb[0]);
    if (lane == 0) L.path[0] = 0;
    while (cbn != 0) {
        base = cbn;
        const int np = __builtin_amdgcn_readfirstlane(L.nodes[node].visit);
        const double tn = c.pbc[np];
        if (s.consumed + c.A > s.win_base + kWave) noise_refill(c, s, lane);
        if (s.consumed + c.A > c.noise_cap) s.status = 1;
        const int src = (int)(s.consumed - s.win_base) + lane;
        const double u = __shfl(s.win, src & (kWave - 1));
        double score = 0.0;
        int mycb = 0;
        if (lane < c.A) {
            /* ucb_score (mcts.py:155-163) */
            const Node ch = L.nodes[base + lane];
            mycb = L.cb[base + lane];
            const double pd = node == 0 ? s.root_prior : (double)ch.prior;
            const double ps = (tn * pd) / (double)(ch.visit + 1);
            double vs = 0.0;
            if (ch.visit > 0) {
                const float q = ch.vsum / (float)ch.visit;
                float x = c.g32 * q;
                x = ch.reward + x;
                if (s.mx > s.mn) x = (x - s.mn) / (s.mx - s.mn);
                vs = (double)x;
            }
            score = (ps + vs) + u;
        }
        /* max over (score, action) tuples: ties -> larger action (mcts.py:167) */
        double best = readlane_f64(score, 0);
        int besta = 0;
        for (int a = 1; a < c.A; ++a) {
            const double sc = readlane_f64(score, a);
            if (sc >= best) {
                best = sc;
                besta = a;
            }
        }
        s.consumed += c.A;
        act = besta;
        node = base + besta;
        cbn = __builtin_amdgcn_readlane(mycb, besta);
        ++d;
        if (lane == 0) L.path[d] = (unsigned short)node;
    }
    s.path_total += d;
    *depth = d;
    *action = act;
    *parent_base = base;
    wave_sync();
    return node;
}

/* update_reward... + create_new_node... (mcts.py:186-200): children of expansion e */
__device__ __forceinline__ void wave_expand(const SearchConst& c, const WaveLds& L, int leaf,
                                            int e, float reward, const Pair pol, int lane) {
    const int cbase = 1 + e * c.A;
    const int i = 2 * lane;
    if (pol.va) L.nodes[cbase + i].prior = pol.a;
    if (pol.vb) L.nodes[cbase + i + 1].prior = pol.b;
    if (lane == 0) {
        L.nodes[leaf].reward = reward;
        L.cb[leaf] = (unsigned short)cbase;
    }
    wave_sync();
}

/* back_propagate_and_update_min_max_bound (mcts.py:203-209): path[0..depth], leaf last.
 * The value recurrence is a serial fp32 chain (product and sum rounded separately);
 * the per-node updates and the min/max of the new means are done one node per lane. */
template <typename State>
__device__ __forceinline__ void wave_backup(const SearchConst& c, State& s, const WaveLds& L,
                                            int depth, float value, int lane) {
    float v = value;
    for (int top = depth; top >= 0; top -= kWave) {
        const int cnt = top + 1 < kWave ? top + 1 : kWave;
        const bool valid = lane < cnt;
        const int node = valid ? L.path[top - lane] : 0;
        Node nd = L.nodes[node];
        float myv = 0.0f;
        for (int k = 0; k < cnt; ++k) {
            const float r = readlane_f32(nd.reward, k);
            myv = lane == k ? v : myv;
            const float gv = c.g32 * v;
            v = r + gv;
        }
        const float pinf = __builtin_inff();
        float q = 0.0f;
        if (valid) {
            nd.vsum = nd.vsum + myv;
            nd.visit += 1;
            q = nd.vsum / (float)nd.visit;
            L.nodes[node].vsum = nd.vsum;
            L.nodes[node].visit = nd.visit;
        }
        const float qmx = wave_max(valid ? q : -pinf);
        const float qmn = wave_min(valid ? q : pinf);
        s.mx = qmx > s.mx ? qmx : s.mx;
        s.mn = qmn < s.mn ? qmn : s.mn;
    }
    wave_sync();
}

}  // namespace mzs
#endif
