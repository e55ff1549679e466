/*
 * mzs_quad.h — four games per wavefront, dense layers on v_mfma_f32_4x4x1_16b_f32.
 *
 * The fused search for the small MLP shapes of the benchmark configs (L = 0, H <= 64,
 * S_dim <= 32, A <= 4; compile-time SD / NA / H).  Same arithmetic as the generic kernel
 * (include/mzs_numeric.h), bit for bit: an f32 MFMA is a k-ordered fmaf chain.
 *
 * Lane roles.  lane l: game c = l & 3, block b = l >> 2.
 *   MFMA (16 blocks of 4x4, K = 1 per instruction):  D[r][lane (b, c)] += A[lane 4b + r] * B
 *     A operand: weights, lane 4b + r holds W[neuron n(b, r)][k]             (registers, all search long)
 *     B operand: inputs, lane 16g + 4bb + c holds x[game c][k = 16g + q] in register q;
 *                BLGP = 4 + g broadcasts lane group g to all sixteen blocks
 *     D: lane (b, c) register r = pre-activation of neuron n(b, r) for game c
 *   n(b, r) = 32 * half + 4 * eb + r, half = b >> 3, eb = b1 + 2 b0 + 4 b2 (bits of b): with
 *   this numbering the adjacent-pair sum tree of contract "sum" is: four registers in the lane,
 *   row_ror:8, row_ror:4, then the neighbouring row (v_permlane16_swap) — the two 32-lane
 *   halves reduce separately and hold the two heads of a net (reward | next state,
 *   policy | value).
 *   Tree work: lanes with equal (c, a = b % A') score child a of game c's current node
 *   (A' = A rounded up to a power of two); b is the path level in backup.
 *
 * Reference code restated (file:line in /root/reference): as mzs_device.h.
 */
#ifndef MZS_QUAD_H
#define MZS_QUAD_H

#include "mzs_device.h"

namespace mzs {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int kDppRowRor4 = 0x124;
constexpr int kDppRowRor8 = 0x128;

struct QuadLane {
    int lane, c, b, grp, half, eb, n0; /* n0 = first of the four neurons of this lane's D registers */
};

__device__ __forceinline__ QuadLane quad_lane() {
    QuadLane q;
    q.lane = lane_id();
    q.c = q.lane & 3;
    q.b = q.lane >> 2;
    q.grp = q.lane >> 4;
    q.half = q.lane >> 5;
    q.eb = ((q.b >> 1) & 1) | ((q.b & 1) << 1) | (q.b & 4);
    q.n0 = 32 * q.half + 4 * q.eb;
    return q;
}

/* neuron whose weights lane l holds as the A operand */
__device__ __forceinline__ int quad_neuron_of_lane(int l) {
    const int b = l >> 2;
    const int eb = ((b >> 1) & 1) | ((b & 1) << 1) | (b & 4);
    return 32 * (l >> 5) + 4 * eb + (l & 3);
}

/* {value in the even row, value in the odd row} of this lane's row pair / {low half, high half} */
__device__ __forceinline__ void rows_pair(float v, float& x, float& y) {
    const u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    x = __uint_as_float(r[0]);
    y = __uint_as_float(r[1]);
}

__device__ __forceinline__ void halves_pair(float v, float& lo, float& hi) {
    const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    lo = __uint_as_float(r[0]);
    hi = __uint_as_float(r[1]);
}

__device__ __forceinline__ float low_half(float v) {
    float lo, hi;
    halves_pair(v, lo, hi);
    return lo;
}

__device__ __forceinline__ float high_half(float v) {
    float lo, hi;
    halves_pair(v, lo, hi);
    return hi;
}

/* No NaNs on this path: plain v_max_f32 (fmaxf() would add a canonicalising v_max per operand). */
__device__ __forceinline__ float fmax_sel(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

/* max(v, v of the lane 8 / 4 further round the row) in one DPP instruction.  The s_nop covers the
 * VALU-write -> DPP-read hazard, which the compiler cannot see inside an asm string. */
__device__ __forceinline__ float max_ror8(float v) {
    float r;
    asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v));
    return r;
}
__device__ __forceinline__ float max_ror4(float v) {
    float r;
    asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v));
    return r;
}

/* reductions over the 8 blocks of this lane's half, per game (result in every lane of the half) */
__device__ __forceinline__ float half_sum(float v) {
    v = v + dpp_mov<kDppRowRor8>(v);
    v = v + dpp_mov<kDppRowRor4>(v);
    float x, y;
    rows_pair(v, x, y);
    return x + y;
}

__device__ __forceinline__ float half_max(float v) {
    v = max_ror8(v);
    v = max_ror4(v);
    float x, y;
    rows_pair(v, x, y);
    return fmax_sel(x, y);
}

/* over all 16 lanes of the game */
__device__ __forceinline__ float game_max(float v) {
    v = half_max(v);
    float lo, hi;
    halves_pair(v, lo, hi);
    return fmax_sel(lo, hi);
}

__device__ __forceinline__ float sum4(const f32x4 v) { return (v[0] + v[1]) + (v[2] + v[3]); }
__device__ __forceinline__ float max4(const f32x4 v) {
    return fmax_sel(fmax_sel(v[0], v[1]), fmax_sel(v[2], v[3]));
}

/* contract "dot" for the four games: even k accumulate into e (starts at the bias), odd k
 * into o (starts at zero); two independent MFMA chains issue back to back */
template <int K, int k = 0>
__device__ __forceinline__ void quad_chain2(const float* w, const float* x, f32x4& e, f32x4& o) {
    if constexpr (k < K) {
        if constexpr (k & 1)
            o = __builtin_amdgcn_mfma_f32_4x4x1f32(w[k], x[k & 15], o, 0, 0, 4 + (k >> 4));
        else
            e = __builtin_amdgcn_mfma_f32_4x4x1f32(w[k], x[k & 15], e, 0, 0, 4 + (k >> 4));
        quad_chain2<K, k + 1>(w, x, e, o);
    }
}

template <int K>
__device__ __forceinline__ f32x4 quad_dense(const float* w, const float* x, const f32x4 bias) {
    f32x4 e = bias, o = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    quad_chain2<K>(w, x, e, o);
    return e + o;
}

/* B-operand registers: x[q] = row[16 * g + q] */
__device__ __forceinline__ void quad_load_x(const float* row, int g, float (&x)[16]) {
    const float4* p = reinterpret_cast<const float4*>(row + 16 * g);
    const float4 v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3];
    x[0] = v0.x, x[1] = v0.y, x[2] = v0.z, x[3] = v0.w;
    x[4] = v1.x, x[5] = v1.y, x[6] = v1.z, x[7] = v1.w;
    x[8] = v2.x, x[9] = v2.y, x[10] = v2.z, x[11] = v2.w;
    x[12] = v3.x, x[13] = v3.y, x[14] = v3.z, x[15] = v3.w;
}

/* kMid: the tied mid layer of each net (neural_network_mlp_model.py:38-39, applied m.L times)
 * is register resident too: H more weight rows per net */
template <int SD, int NA, int H, bool kMid = false>
struct QuadWeights {
    float d_in[SD], d_h[H], p_in[SD], p_h[H];
    float d_mid[kMid ? H : 1], p_mid[kMid ? H : 1];
    f32x4 b_d_in, b_d_h, b_p_in, b_p_h, b_d_mid, b_p_mid;
};

/* 16-byte tree node of this kernel: everything pUCT reads about a child in one ds_read_b128.
 * x0 = reward + f32(gamma) * value() is refreshed by backup (the un-normalised value score of
 * mcts.py:158-160); value sums live in their own array. */
struct __attribute__((aligned(16))) QNode {
    unsigned vc;   /* child_base | visit_count << 16 */
    float x0;
    float prior;
    float reward;
};

/* per-wave LDS */
/* LDS row strides (floats) chosen so that the sixteen 16-byte pieces one ds_read_b128 of the B operands
 * fetches — rows of the four games x lane groups — fall into disjoint banks (64 banks of 4 bytes; with
 * the natural strides 64 / 32 the four games' rows share theirs: a 4-way / 2-way conflict on the
 * critical path of every layer).  Game records: see make_quad_layout. */
constexpr int kTrunkStride = 68; /* trunk activations of a game */
constexpr int kHidRowStride = 36; /* kHidG: the one hidden-state row per game */

struct QuadLds {
    float* trunk;            /* [4][kTrunkStride] trunk activations, game-major            */
    double* nwin;            /* [64] tie-break draws of the next descent: [level][a][game]  */
    const float* onehot;     /* block-shared [NA][64] action columns of the dynamics input  */
    const double* pbc;       /* block-shared [S+2] sqrt(n) * pb_c(n)                        */
    const double* inv64;     /* block-shared [S+2] 1/n, correctly rounded                   */
    const float* inv32;      /* block-shared [S+2] 1/n, correctly rounded                   */
    /* this lane's game */
    QNode* nodes;
    float* vsum;             /* [n_nodes] Node.value_sum                                    */
    uint2* path;             /* [S+2] {node, reward bits}                                   */
    float* hid;              /* [S+1][32]                                                   */
    unsigned short* leafs;   /* [S]                                                         */
};

struct QuadLayout {
    int off_inv64, off_inv32, off_onehot, shared_bytes; /* block-shared part: pbc at 0  */
    int off_tabs;                                       /* small-vector tables (mzs_quadw.h WideTabs), else 0 */
    int off_nwin, off_games, wave_bytes;                /* per wave: trunk at 0         */
    int g_vsum, g_path, g_hid, g_leafs, g_bytes;        /* per game: nodes at 0         */
};

template <int SD, int NA, int H, bool kMid>
__device__ __forceinline__ void quad_load_weights(const Model& m, QuadWeights<SD, NA, H, kMid>& W,
                                                  const QuadLane& q) {
    const float* __restrict__ w = m.w;
    const int n = quad_neuron_of_lane(q.lane);
#pragma unroll
    for (int k = 0; k < SD; ++k) W.d_in[k] = w[m.d_in.w_off + k * 64 + n];
#pragma unroll
    for (int k = 0; k < H; ++k) W.d_h[k] = w[m.d_heads.w_off + k * 64 + n];
#pragma unroll
    for (int k = 0; k < SD; ++k) W.p_in[k] = w[m.p_in.w_off + k * 64 + n];
#pragma unroll
    for (int k = 0; k < H; ++k) W.p_h[k] = w[m.p_heads.w_off + k * 64 + n];
    const auto b4 = [&](int off) {
        const float4 v = *reinterpret_cast<const float4*>(w + off + q.n0);
        return f32x4{v.x, v.y, v.z, v.w};
    };
    W.b_d_in = b4(m.d_in.b_off);
    W.b_d_h = b4(m.d_heads.b_off);
    W.b_p_in = b4(m.p_in.b_off);
    W.b_p_h = b4(m.p_heads.b_off);
    if constexpr (kMid) {
#pragma unroll
        for (int k = 0; k < H; ++k) W.d_mid[k] = w[m.d_mid.w_off + k * 64 + n];
#pragma unroll
        for (int k = 0; k < H; ++k) W.p_mid[k] = w[m.p_mid.w_off + k * 64 + n];
        W.b_d_mid = b4(m.d_mid.b_off);
        W.b_p_mid = b4(m.p_mid.b_off);
    }
}

/* ---- the contract's exp / expm1 / corrected quotient on four values at once: the operation
 * sequence of mzs_expf / mzs_expm1f_neg / mzs_div_r2 per element (include/mzs_numeric.h), written
 * on vectors so that the multiplies, adds and fmas pair up into v_pk_{mul,add,fma}_f32 (two IEEE
 * operations per instruction: same bits, about a third fewer VALU instructions in the layer
 * epilogues) */
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 fma4(const f32x4 a, const f32x4 b, const f32x4 c) {
    return __builtin_elementwise_fma(a, b, c);
}
__device__ __forceinline__ f32x4 splat4(float v) { return f32x4{v, v, v, v}; }

__device__ __forceinline__ f32x4 expm1_poly4(const f32x4 r) {
    f32x4 p = splat4(1.9875691500e-4f);
    p = fma4(p, r, splat4(1.3981999507e-3f));
    p = fma4(p, r, splat4(8.3334519073e-3f));
    p = fma4(p, r, splat4(4.1665795894e-2f));
    p = fma4(p, r, splat4(1.6666665459e-1f));
    p = fma4(p, r, splat4(5.0000001201e-1f));
    const f32x4 r2 = r * r;
    return fma4(p, r2, r);
}
/* Cody-Waite split of the clamped argument; 2^n of each element in s */
__device__ __forceinline__ f32x4 exp_reduce4(const f32x4 xc, f32x4* s) {
    const f32x4 t = xc * splat4(1.44269504088896341f);
    const f32x4 n = __builtin_elementwise_rint(t);
    f32x4 r = fma4(n, splat4(-0.693359375f), xc);
    r = fma4(n, splat4(2.12194440e-4f), r);
    const i32x4 ni = __builtin_convertvector(n, i32x4);
    *s = f32x4{mzs_pow2i(ni[0]), mzs_pow2i(ni[1]), mzs_pow2i(ni[2]), mzs_pow2i(ni[3])};
    return r;
}
__device__ __forceinline__ f32x4 clamp4(const f32x4 x, float lo, float hi) {
    f32x4 y;
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i] = __builtin_fmaxf(__builtin_fminf(x[i], hi), lo);
    return y;
}
__device__ __forceinline__ f32x4 expf4(const f32x4 x) {
    f32x4 s;
    const f32x4 r = exp_reduce4(clamp4(x, -87.0f, 88.0f), &s);
    const f32x4 e = expm1_poly4(r) + splat4(1.0f);
    return e * s;
}
__device__ __forceinline__ f32x4 expm1_neg4(const f32x4 x) {
    f32x4 s;
    const f32x4 r = exp_reduce4(clamp4(x, -87.0f, 0.0f), &s);
    return fma4(s, expm1_poly4(r), s - splat4(1.0f));
}
/* kPacked = false: the scalar contract functions element by element — for kernels whose register
 * file is full (the packed forms want aligned register pairs; measured: the kMid kernels and
 * k_search_quadw lose 2-10 % with them, the others gain 2 %) */
template <bool kPacked = true>
__device__ __forceinline__ f32x4 elu4(const f32x4 v) {
    if constexpr (!kPacked) return f32x4{mzs_elu(v[0]), mzs_elu(v[1]), mzs_elu(v[2]), mzs_elu(v[3])};
    const f32x4 y = expm1_neg4(v);
    return f32x4{v[0] > 0.0f ? v[0] : y[0], v[1] > 0.0f ? v[1] : y[1], v[2] > 0.0f ? v[2] : y[2],
                 v[3] > 0.0f ? v[3] : y[3]};
}
template <bool kPacked = true>
__device__ __forceinline__ f32x4 exp4(const f32x4 x) {
    if constexpr (!kPacked) return f32x4{mzs_expf(x[0]), mzs_expf(x[1]), mzs_expf(x[2]), mzs_expf(x[3])};
    return expf4(x);
}
/* a[i] / d with r = RN(1 / d) */
template <bool kPacked = true>
__device__ __forceinline__ f32x4 div_r2_4(const f32x4 a, float d, float r) {
    if constexpr (!kPacked)
        return f32x4{mzs_div_r2(a[0], d, r), mzs_div_r2(a[1], d, r), mzs_div_r2(a[2], d, r), mzs_div_r2(a[3], d, r)};
    const f32x4 dv = splat4(d), rv = splat4(r);
    f32x4 q = a * rv;
    q = fma4(fma4(-q, dv, a), rv, q);
    return fma4(fma4(-q, dv, a), rv, q);
}

__device__ __forceinline__ void store4(float* p, const f32x4 v) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}

/* ELU(trunk) of the four games -> LDS, back as the B operand of the head layer */
template <bool kPacked = true>
__device__ __forceinline__ void quad_trunk_to_x(const QuadLds& L, const QuadLane& q, const f32x4 acc,
                                                float (&x)[16]) {
    store4(L.trunk + q.c * kTrunkStride + q.n0, elu4<kPacked>(acc));
    wave_sync();
    quad_load_x(L.trunk + q.c * kTrunkStride, q.grp, x);
    wave_sync();
}

/* a / d given r = RN(1/d): product, then fma residual corrections (Markstein).  Each step
 * takes a faithful quotient to the correctly rounded one; one step is exhaustively verified for
 * fp32 with integer d <= 512 (tests/test_division.py), elsewhere two steps are used.  No
 * overflow / underflow on this path: operands are visit counts, values and priors. */
__device__ __forceinline__ float div_r1(float a, float d, float r) { return mzs_div_r1(a, d, r); }
__device__ __forceinline__ float div_r2(float a, float d, float r) { return mzs_div_r2(a, d, r); }
__device__ __forceinline__ double div_r2(double a, double d, double r) {
    double q = a * r;
    q = __builtin_fma(__builtin_fma(-q, d, a), r, q);
    return __builtin_fma(__builtin_fma(-q, d, a), r, q);
}

/* softmax over the valid logits of this lane's half (contract exp + sum) */
template <bool kPacked = true>
__device__ __forceinline__ f32x4 quad_softmax(const f32x4 o, const bool (&valid)[4]) {
    const float ninf = -__builtin_inff();
    const float m = half_max(max4(f32x4{valid[0] ? o[0] : ninf, valid[1] ? o[1] : ninf,
                                        valid[2] ? o[2] : ninf, valid[3] ? o[3] : ninf}));
    f32x4 e = exp4<kPacked>(o - splat4(m));
#pragma unroll
    for (int r = 0; r < 4; ++r) e[r] = valid[r] ? e[r] : 0.0f;
    const float Z = half_sum(sum4(e));
    return div_r2_4<kPacked>(e, Z, 1.0f / Z);
}

/* expectation over the support i - SD/2 of probabilities p (muzero_model.py:580-584); the
 * inverse value transform is applied once for reward and value together (quad_inverse2) */
template <int SD>
__device__ __forceinline__ float quad_support_mean(const f32x4 p, const bool (&valid)[4], int i0) {
    f32x4 t;
#pragma unroll
    for (int r = 0; r < 4; ++r) t[r] = valid[r] ? (float)(i0 + r - SD / 2) * p[r] : 0.0f;
    return half_sum(sum4(t));
}

/* mzs_inverse_value_transform (muzero_model.py:585-586) with its division by the constant 0.002
 * done as a corrected-reciprocal quotient (same correctly rounded result, see div_r2) */
__device__ __forceinline__ float quad_inverse_value_transform(float y) {
    const float a = __builtin_fabsf(y);
    float t = (a + 1.0f) + 0.001f;
    t = 0.004f * t;
    t = 1.0f + t;
    float r = __builtin_sqrtf(t);
    r = r - 1.0f;
    {
        const float d = 0.002f, rc = 1.0f / 0.002f; /* constant-folded, correctly rounded */
        float q = r * rc;
        q = mzs_fma(mzs_fma(-q, d, r), rc, q);
        r = mzs_fma(mzs_fma(-q, d, r), rc, q);
    }
    r = r * r;
    r = r - 1.0f;
    const float sg = (y > 0.0f) ? 1.0f : ((y < 0.0f) ? -1.0f : 0.0f);
    return sg * r;
}

/* scale_to_bound_action (neural_network_mlp_model.py:135-143) of this lane's half */
template <bool kPacked = true>
__device__ __forceinline__ f32x4 quad_scale_to_bound(const f32x4 o, const bool (&valid)[4]) {
    const float ninf = -__builtin_inff();
    const float mx = half_max(max4(f32x4{valid[0] ? o[0] : ninf, valid[1] ? o[1] : ninf,
                                         valid[2] ? o[2] : ninf, valid[3] ? o[3] : ninf}));
    const float mn = -half_max(max4(f32x4{valid[0] ? -o[0] : ninf, valid[1] ? -o[1] : ninf,
                                          valid[2] ? -o[2] : ninf, valid[3] ? -o[3] : ninf}));
    const float sc = mzs_bound_scale(mn, mx), rs = 1.0f / sc;
    f32x4 y = div_r2_4<kPacked>(o - splat4(mn), sc, rs);
#pragma unroll
    for (int r = 0; r < 4; ++r) y[r] = valid[r] ? y[r] : 0.0f;
    return y;
}

/* Prediction_function on the hidden-state row `h` of each game (LDS): policy logits in the
 * low half (block eb = 0 holds actions 0..3: lanes 0..3), value logits in the high half; the two
 * softmaxes run in the same instructions.  Returns the value's support expectation (before the
 * inverse transform) in the high half's lanes. */
template <int SD, int NA, int H, bool kMid>
__device__ __forceinline__ float quad_prediction(const QuadWeights<SD, NA, H, kMid>& W, const QuadLds& L,
                                                 const QuadLane& q, const float* h, f32x4* prior, int n_mid) {
    float x[16];
    quad_load_x(h, q.grp & 1, x);
    f32x4 acc = quad_dense<SD>(W.p_in, x, W.b_p_in);
    quad_trunk_to_x<!kMid>(L, q, acc, x);
    if constexpr (kMid) {
        for (int l = 0; l < n_mid; ++l) {
            acc = quad_dense<H>(W.p_mid, x, W.b_p_mid);
            quad_trunk_to_x<!kMid>(L, q, acc, x);
        }
    }
    acc = quad_dense<H>(W.p_h, x, W.b_p_h);
    const int i0 = 4 * q.eb, lim = q.half ? SD : NA;
    const bool valid[4] = {i0 < lim, i0 + 1 < lim, i0 + 2 < lim, i0 + 3 < lim};
    const f32x4 p = quad_softmax<!kMid>(acc, valid);
    *prior = p;
    const bool vv[4] = {valid[0] && q.half, valid[1] && q.half, valid[2] && q.half, valid[3] && q.half};
    return quad_support_mean<SD>(p, vv, i0);
}

/* Dynamics_function on row `h` + action: reward logits in the low half, next-state
 * pre-activations in the high half; the new hidden state goes to row `hn` (LDS).  Returns the
 * reward's support expectation (before the inverse transform) in the low half's lanes. */
template <int SD, int NA, int H, bool kMid>
__device__ __forceinline__ float quad_dynamics(const QuadWeights<SD, NA, H, kMid>& W, const QuadLds& L,
                                               const QuadLane& q, const float* h, int action, float* hn,
                                               int n_mid) {
    float x[16];
    quad_load_x(h, q.grp & 1, x);
    const float4 oh = *reinterpret_cast<const float4*>(L.onehot + action * 64 + q.n0);
    f32x4 acc = quad_dense<SD>(W.d_in, x, W.b_d_in);
    /* one-hot action column (muzero_model.py:502): fma(1, w, acc) == acc + w */
    acc = f32x4{acc[0] + oh.x, acc[1] + oh.y, acc[2] + oh.z, acc[3] + oh.w};
    quad_trunk_to_x<!kMid>(L, q, acc, x);
    if constexpr (kMid) {
        for (int l = 0; l < n_mid; ++l) {
            acc = quad_dense<H>(W.d_mid, x, W.b_d_mid);
            quad_trunk_to_x<!kMid>(L, q, acc, x);
        }
    }
    acc = quad_dense<H>(W.d_h, x, W.b_d_h);
    const int i0 = 4 * q.eb;
    const bool valid[4] = {i0 < SD, i0 + 1 < SD, i0 + 2 < SD, i0 + 3 < SD};
    /* reward head */
    const f32x4 p = quad_softmax<!kMid>(acc, valid);
    const float reward = quad_support_mean<SD>(p, valid, i0);
    /* next state */
    const f32x4 y = quad_scale_to_bound<!kMid>(acc, valid);
    if (q.half) store4(hn + i0, y);
    wave_sync();
    return reward;
}

/* Representation_function on the observations staged in L.trunk ([game][64]); weights streamed
 * from L2 (once per search).  Hidden state to row `h0` of each game. */
template <int SD, int NA, int H>
__device__ __forceinline__ void quad_representation(const Model& m, const QuadLds& L, const QuadLane& q,
                                                    float* h0) {
    const float* __restrict__ w = m.w;
    const int n = quad_neuron_of_lane(q.lane);
    const auto b4 = [&](int off) {
        const float4 v = *reinterpret_cast<const float4*>(w + off + q.n0);
        return f32x4{v.x, v.y, v.z, v.w};
    };
    f32x4 acc = b4(m.r_in.b_off), odd = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    for (int k = 0; k < m.obs_dim; ++k) {
        const float wk = w[m.r_in.w_off + k * 64 + n], xk = L.trunk[q.c * kTrunkStride + k];
        if (k & 1)
            odd = __builtin_amdgcn_mfma_f32_4x4x1f32(wk, xk, odd, 0, 0, 0);
        else
            acc = __builtin_amdgcn_mfma_f32_4x4x1f32(wk, xk, acc, 0, 0, 0);
    }
    acc = acc + odd;
    wave_sync();
    float x[16];
    quad_trunk_to_x(L, q, acc, x);
    float wk[H];
    for (int l = 0; l < m.L; ++l) { /* tied mid layer, streamed like the rest of this net */
#pragma unroll
        for (int k = 0; k < H; ++k) wk[k] = w[m.r_mid.w_off + k * 64 + n];
        acc = quad_dense<H>(wk, x, b4(m.r_mid.b_off));
        quad_trunk_to_x(L, q, acc, x);
    }
#pragma unroll
    for (int k = 0; k < H; ++k) wk[k] = w[m.r_out.w_off + k * 64 + n];
    acc = quad_dense<H>(wk, x, b4(m.r_out.b_off));
    const int i0 = 4 * q.eb;
    const bool valid[4] = {i0 < SD, i0 + 1 < SD, i0 + 2 < SD, i0 + 3 < SD};
    const f32x4 y = quad_scale_to_bound(acc, valid);
    if (!q.half) store4(h0 + i0, y);
    wave_sync();
}

/* ------------------------------------------------------------------ search */

struct QuadState {
    float mn, mx;       /* MinMaxStats of this lane's game        */
    float rinv;         /* 1 / (mx - mn), refreshed by backup      */
    double rootp;       /* root-child prior of (game, a) after noise */
    int consumed;       /* tie-break draws consumed               */
    int path_total;
    int status;
};

/* row rotations read a valid lane everywhere: bound_ctrl spares the zero-initialised destination */
template <int kCtrl>
__device__ __forceinline__ int quad_dpp_i32(int v) {
    return __builtin_amdgcn_mov_dpp(v, kCtrl, 0xF, 0xF, true);
}

template <int kCtrl>
__device__ __forceinline__ double quad_dpp_f64(double v) {
    const int lo = quad_dpp_i32<kCtrl>(__double2loint(v));
    const int hi = quad_dpp_i32<kCtrl>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

/* max over (score, action) tuples with the chosen child's record riding along */
template <int kCtrl>
__device__ __forceinline__ void quad_argmax_step(double& bs, int& ba, unsigned& pay, float& rew) {
    const double os = quad_dpp_f64<kCtrl>(bs);
    const int oa = quad_dpp_i32<kCtrl>(ba);
    const unsigned op = (unsigned)quad_dpp_i32<kCtrl>((int)pay);
    const float orw = __int_as_float(quad_dpp_i32<kCtrl>(__float_as_int(rew)));
    /* ties -> larger action (mcts.py:167) */
    const bool take = (os > bs) | ((os == bs) & (oa > ba));
    bs = take ? os : bs;
    ba = take ? oa : ba;
    pay = take ? op : pay;
    rew = take ? orw : rew;
}

/* RN(1 / d) in f64 for a visit count d: hardware estimate + two Newton steps instead of a second
 * dependent LDS read per tree level (the table lookup needs the child's record first).  Equal
 * to the correctly rounded quotient for every 1 <= d <= 65536 — all the counts a tree of at most
 * 65 535 nodes can hold — which mzs_selftest_reciprocals checks on the device in use
 * (tests/test_gpu_parity.py). */
__device__ __forceinline__ double quad_rcp_count(int d) {
    const double x = (double)d;
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    return __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
}

template <int NA>
struct QuadSelect {
    int leaf, depth, action, pbase;
};

/* Descent of the four games at once, one tree level per iteration (mcts.py:155-177).
 * Lane (c, a) scores child a of game c's current node; games that reached a leaf idle.
 * Straight-line per level: no data-dependent branches. */
template <int NA, bool kTableRcp = false>
__device__ __forceinline__ QuadSelect<NA> quad_select(const QuadLds& L, QuadState& s, const QuadLane& q,
                                                      const double* __restrict__ noise, int noise_cap,
                                                      int sim) {
    constexpr int NAL = NA <= 2 ? 2 : 4; /* lanes per sibling group */
    constexpr int kWin = 16 / NAL;       /* levels held by the LDS window */
    const int a = q.b & (NAL - 1);
    const int aa = a < NA ? a : 0;
    const bool norm = s.mx > s.mn;
    const float rng = s.mx - s.mn;
    QuadSelect<NA> r;
    r.leaf = 0, r.depth = 0, r.action = 0, r.pbase = 0;
    int cbn = 1;     /* the root is expanded: its children start at node 1 */
    int pv = sim;    /* root visit count = completed simulations */
    for (int t = 0;; ++t) {
        const bool active = cbn != 0;
        if (__builtin_amdgcn_ballot_w64(active) == 0) break;
        if (t > 0 && (t & (kWin - 1)) == 0) {
            /* deeper than the window: fetch the next kWin levels' draws */
            wave_sync();
            const int ni = s.consumed + (t + q.b / NAL) * NA + a;
            L.nwin[q.lane] = (noise && a < NA && ni < noise_cap) ? noise[ni] : 0.0;
            wave_sync();
        }
        const int child = (active ? cbn : 1) + aa;
        const QNode ch = L.nodes[child];
        const double tn = L.pbc[pv];
        const int n = (int)(ch.vc >> 16);
        /* 1 / (n + 1): computed (no second dependent LDS read per level) unless the kernel has no
         * registers to spare for it */
        const double iv = kTableRcp ? L.inv64[n + 1] : quad_rcp_count(n + 1);
        const double u = L.nwin[((t & (kWin - 1)) * NAL + a) * 4 + q.c];
        /* ucb_score (mcts.py:155-163) */
        const double pd = t == 0 ? s.rootp : (double)ch.prior;
        const double ps = div_r2(tn * pd, (double)(n + 1), iv);
        const float xn = div_r2(ch.x0 - s.mn, rng, s.rinv);
        const float x = norm ? xn : ch.x0;
        const double vs = n > 0 ? (double)x : 0.0;
        double bs = (ps + vs) + u;
        if (NA < NAL) bs = a < NA ? bs : -__builtin_inf();
        int ba = a;
        unsigned pay = ch.vc;
        float rew = ch.reward;
        if (NAL == 4) quad_argmax_step<kDppRowRor8>(bs, ba, pay, rew);
        quad_argmax_step<kDppRowRor4>(bs, ba, pay, rew);
        r.pbase = active ? cbn : r.pbase;
        r.action = active ? ba : r.action;
        r.leaf = active ? cbn + ba : r.leaf;
        r.depth += active ? 1 : 0;
        pv = active ? (int)(pay >> 16) : pv;
        if (active && q.b == 0) L.path[r.depth] = make_uint2((unsigned)r.leaf, __float_as_uint(rew));
        cbn = active ? (int)(pay & 0xffffu) : 0;
    }
    s.consumed += NA * r.depth;
    if (s.consumed > noise_cap) s.status = 1;
    s.path_total += r.depth;
    wave_sync();
    return r;
}

/* draws of the next descent: lane (c, a, level j = b / NAL) fetches draw consumed + j * NA + a */
template <int NA>
__device__ __forceinline__ double quad_noise_prefetch(const QuadState& s, const QuadLane& q,
                                                      const double* __restrict__ noise, int noise_cap) {
    constexpr int NAL = NA <= 2 ? 2 : 4;
    const int a = q.b & (NAL - 1), j = q.b / NAL;
    const int ni = s.consumed + j * NA + a;
    return (noise && a < NA && ni < noise_cap) ? noise[ni] : 0.0;
}

/* back_propagate_and_update_min_max_bound (mcts.py:203-209) for the four games: the value
 * recurrence is a serial fp32 chain per game (product and sum rounded separately); the node
 * updates of up to kChunk path levels run one per lane. */
__device__ __forceinline__ void quad_backup(const QuadLds& L, QuadState& s, const QuadLane& q, int depth,
                                            float value, float g32) {
    constexpr int kChunk = 8;
    int dmax = depth;
    dmax = max(dmax, __builtin_amdgcn_update_dpp(0, dmax, kDppQuadXor1, 0xF, 0xF, false));
    dmax = max(dmax, __builtin_amdgcn_update_dpp(0, dmax, kDppQuadXor2, 0xF, 0xF, false));
    dmax = __builtin_amdgcn_readfirstlane(dmax);
    const int slot = q.b & (kChunk - 1);
    const float ninf = -__builtin_inff();
    float v = value;
    for (int hi = dmax; hi >= 0; hi -= kChunk) {
        float rw[kChunk];
#pragma unroll
        for (int k = 0; k < kChunk; ++k) rw[k] = __uint_as_float(L.path[hi - k > 0 ? hi - k : 0].y);
        const int lvl = hi - slot;
        const bool mine = q.b < kChunk && lvl >= 0 && lvl <= depth;
        const uint2 pe = L.path[mine ? lvl : 0];
        const int pn = (int)pe.x;
        const unsigned vc = L.nodes[pn].vc;
        const float vs0 = L.vsum[pn];
        float myv = 0.0f;
#pragma unroll
        for (int k = 0; k < kChunk; ++k) {
            const bool on = hi - k >= 0 && hi - k <= depth;
            myv = (on && slot == k) ? v : myv;
            const float gv = g32 * v;
            const float nv = rw[k] + gv;
            v = on ? nv : v;
        }
        const int visit = (int)(vc >> 16) + 1;
        const float vsum = vs0 + myv;
        const float qv = div_r1(vsum, (float)visit, L.inv32[visit]);
        if (mine) {
            const float gq = g32 * qv;
            const float x0 = __uint_as_float(pe.y) + gq;
            *reinterpret_cast<uint2*>(&L.nodes[pn]) = make_uint2(vc + 0x10000u, __float_as_uint(x0));
            L.vsum[pn] = vsum;
        }
        const float qmx = game_max(mine ? qv : ninf);
        const float qmn = -game_max(mine ? -qv : ninf);
        s.mx = fmax_sel(s.mx, qmx);
        s.mn = __builtin_fminf(s.mn, qmn);
        wave_sync();
    }
    s.rinv = 1.0f / (s.mx - s.mn);
}

}  // namespace mzs
#endif
