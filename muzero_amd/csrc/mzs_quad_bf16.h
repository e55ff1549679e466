/*
 * mzs_quad_bf16.h — the four-games-per-wavefront model functions of mzs_quad.h under contract
 * "dot_bf16" (include/mzs_numeric.h): dense layers on v_mfma_f32_4x4x4_16B_bf16, K = 4 per
 * instruction, always with a ZERO accumulator; the group results are added with IEEE fp32 adds
 * (two interleaved chains).  Everything after the dot — bias order, one-hot column, ELU, softmax,
 * scale_to_bound, inverse value transform — is the fp32 code of mzs_quad.h.
 *
 * Lane roles are those of mzs_quad.h.  MFMA operands:
 *   A: lane 4b + r holds four consecutive-k weights of neuron n(b, r) as packed bf16 (2 VGPRs per
 *      group of four k): half the registers of the fp32 kernel, a quarter of its MFMAs
 *   B: lane 16g + 4bb + c holds x[game c][16g .. 16g + 15] as 16 bf16 (8 VGPRs); BLGP = 4 + g
 *      broadcasts lane group g (block b reads lanes 16g + 4 (b % 4) + c: the four bb copies are equal)
 *   D: lane (b, c) register r = sum over the four k for neuron n(b, r), game c
 *
 * Reference code restated: as mzs_quad.h (neural_network_mlp_model.py:43-44, :86-88, :127-128).
 */
#ifndef MZS_QUAD_BF16_H
#define MZS_QUAD_BF16_H

#include "mzs_quad.h"

namespace mzs {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

/* two floats -> packed bf16 pair, round to nearest even (v_cvt_pk_bf16_f32; checked against the
 * integer formula of mzs_f2bf16 on 2^20 values incl. ties: tools/mfma_probe_bf16.hip) */
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    const bf16x2_t r = __builtin_convertvector(v, bf16x2_t);
    unsigned u;
    __builtin_memcpy(&u, &r, 4);
    return u;
}

/* bf16 high halves of two floats whose low 16 bits are already zero (the rounded weight store) */
__device__ __forceinline__ unsigned pack_hi(float lo, float hi) {
    return ((unsigned)mzs_f2bits(lo) >> 16) | ((unsigned)mzs_f2bits(hi) & 0xffff0000u);
}

__device__ __forceinline__ s16x4 as_s16x4(unsigned a, unsigned b) {
    const u32x2 v = {a, b};
    s16x4 r;
    __builtin_memcpy(&r, &v, 8);
    return r;
}

constexpr int quad_groups(int K) { return (K + 3) / 4; }

template <int SD, int NA, int H, bool kMid = false>
struct QuadWeightsB {
    s16x4 d_in[quad_groups(SD)], d_h[quad_groups(H)], p_in[quad_groups(SD)], p_h[quad_groups(H)];
    s16x4 d_mid[kMid ? quad_groups(H) : 1], p_mid[kMid ? quad_groups(H) : 1];
    f32x4 b_d_in, b_d_h, b_p_in, b_p_h, b_d_mid, b_p_mid;
};

/* K weights of neuron n from W^T [k][64] (rounded store) as packed groups of four, zero past K */
template <int K>
__device__ __forceinline__ void quad_pack_rows(const float* __restrict__ wt, int n, s16x4 (&out)[quad_groups(K)]) {
#pragma unroll
    for (int g = 0; g < quad_groups(K); ++g) {
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = 4 * g + q < K ? wt[(4 * g + q) * 64 + n] : 0.0f;
        out[g] = as_s16x4(pack_hi(v[0], v[1]), pack_hi(v[2], v[3]));
    }
}

template <int SD, int NA, int H, bool kMid>
__device__ __forceinline__ void quad_load_weights_b(const Model& m, QuadWeightsB<SD, NA, H, kMid>& W,
                                                    const QuadLane& q) {
    const float* __restrict__ w = m.w;
    const int n = quad_neuron_of_lane(q.lane);
    quad_pack_rows<SD>(w + m.d_in.w_off, n, W.d_in);
    quad_pack_rows<H>(w + m.d_heads.w_off, n, W.d_h);
    quad_pack_rows<SD>(w + m.p_in.w_off, n, W.p_in);
    quad_pack_rows<H>(w + m.p_heads.w_off, n, W.p_h);
    const auto b4 = [&](int off) {
        const float4 v = *reinterpret_cast<const float4*>(w + off + q.n0);
        return f32x4{v.x, v.y, v.z, v.w};
    };
    W.b_d_in = b4(m.d_in.b_off);
    W.b_d_h = b4(m.d_heads.b_off);
    W.b_p_in = b4(m.p_in.b_off);
    W.b_p_h = b4(m.p_heads.b_off);
    if constexpr (kMid) {
        quad_pack_rows<H>(w + m.d_mid.w_off, n, W.d_mid);
        quad_pack_rows<H>(w + m.p_mid.w_off, n, W.p_mid);
        W.b_d_mid = b4(m.d_mid.b_off);
        W.b_p_mid = b4(m.p_mid.b_off);
    }
}

/* B-operand registers of lane group g from an fp32 row (hidden state, observation): 16 floats
 * row[16g .. 16g + 15] rounded to bf16, four per K group */
__device__ __forceinline__ void quad_load_xb_f32(const float* row, int g, s16x4 (&x)[4]) {
    float t[16];
    quad_load_x(row, g, t);
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = as_s16x4(pack_bf16(t[4 * i], t[4 * i + 1]), pack_bf16(t[4 * i + 2], t[4 * i + 3]));
}

/* ... from a bf16 row (the trunk activations the previous layer left in LDS): two ds_read_b128 */
__device__ __forceinline__ void quad_load_xb(const unsigned short* row, int g, s16x4 (&x)[4]) {
    const uint4* p = reinterpret_cast<const uint4*>(row + 16 * g);
    const uint4 v0 = p[0], v1 = p[1];
    x[0] = as_s16x4(v0.x, v0.y);
    x[1] = as_s16x4(v0.z, v0.w);
    x[2] = as_s16x4(v1.x, v1.y);
    x[3] = as_s16x4(v1.z, v1.w);
}

/* a + b as four v_add_f32 (each result passes through an empty asm, so the compiler cannot re-pack the adds into
 * v_pk_add_f32; the adds themselves are the compiler's, matrix-result hazards included).  For the bf16x3 chains,
 * three adds per matrix instruction: beside matrix instructions a packed fp32 add costs more than two plain ones
 * (measured on one box: k_search_quad bf16x3 -6 %, k_search_quadw2 -9 %, k_search_quadl -4 %).  The fast contract's
 * single chain measured the other way (tag 123 +3 %, tag 127 +14 %) and keeps the packed adds. */
__device__ __forceinline__ f32x4 add4s(const f32x4 a, const f32x4 b) {
    f32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float ri = a[i] + b[i];
        asm("" : "+v"(ri));
        r[i] = ri;
    }
    return r;
}

/* contract "dot_bf16" for the four games: NG independent MFMAs with a zero accumulator, even
 * groups added to the bias chain, odd groups to the chain that starts at group 1 */
template <int I>
__device__ __forceinline__ f32x4 quad_mfma_b(const s16x4 w, const s16x4 (&x)[4]) {
    /* K group I lives in lane group I / 4 (BLGP = 4 + I / 4), register pair I % 4; zero accumulator */
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    return __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(w, x[I & 3], zero, 0, 0, 4 + (I >> 2));
}

/* Software pipeline: the MFMA of group I + 2 issues before the two packed adds that consume group I,
 * each into registers of its own, so no add waits on the matrix pipe (a 4x4x4 MFMA occupies it
 * for 8 cycles and its result needs ~5 more before a VALU instruction may read it) */
template <int NG, int I = 0>
__device__ __forceinline__ void quad_dense_b_step(const s16x4 (&w)[NG], const s16x4 (&x)[4], f32x4& e, f32x4& o,
                                                  const f32x4 g0, const f32x4 g1) {
    if constexpr (I < NG) {
        f32x4 g2 = g0;
        if constexpr (I + 2 < NG) g2 = quad_mfma_b<I + 2>(w[I + 2], x);
        if constexpr (I & 1) {
            if constexpr (I == 1)
                o = g0;
            else
                o = o + g0;
        } else {
            e = e + g0;
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); /* one MFMA ...            */
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0); /* ... then two VALU (adds) */
        quad_dense_b_step<NG, I + 1>(w, x, e, o, g1, g2);
    }
}

template <int NG>
__device__ __forceinline__ f32x4 quad_dense_b(const s16x4 (&w)[NG], const s16x4 (&x)[4], const f32x4 bias) {
    f32x4 e = bias, o = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const f32x4 g0 = quad_mfma_b<0>(w[0], x);
    f32x4 g1 = g0;
    if constexpr (NG > 1) g1 = quad_mfma_b<1>(w[1], x);
    quad_dense_b_step<NG>(w, x, e, o, g0, g1);
    return e + o;
}

/* ELU(trunk) of the four games -> LDS as bf16, back as the B operand of the next layer.  The
 * bf16 trunk rows (64 halves each) share the memory of the fp32 kernel's trunk buffer, at offsets
 * 64 c + 8 (c >> 1) halves: no uniform stride keeps the sixteen pieces of a ds_read_b128 (four games
 * x four lane groups, 32 bytes apart) in disjoint banks, these offsets do. */
__device__ __forceinline__ int quad_trunk_off_b(int c) { return 64 * c + 8 * (c >> 1); }
template <bool kPacked = true>
__device__ __forceinline__ void quad_trunk_to_xb(const QuadLds& L, const QuadLane& q, const f32x4 acc, s16x4 (&x)[4]) {
    const f32x4 t = elu4<kPacked>(acc);
    unsigned short* tr = reinterpret_cast<unsigned short*>(L.trunk);
    *reinterpret_cast<uint2*>(tr + quad_trunk_off_b(q.c) + q.n0) = make_uint2(pack_bf16(t[0], t[1]), pack_bf16(t[2], t[3]));
    wave_sync();
    quad_load_xb(tr + quad_trunk_off_b(q.c), q.grp, x);
    wave_sync();
}

template <int SD, int NA, int H, bool kMid>
__device__ __forceinline__ float quad_prediction_b(const QuadWeightsB<SD, NA, H, kMid>& W, const QuadLds& L,
                                                   const QuadLane& q, const float* h, f32x4* prior, int n_mid) {
    s16x4 x[4];
    quad_load_xb_f32(h, q.grp & 1, x);
    f32x4 acc = quad_dense_b<quad_groups(SD)>(W.p_in, x, W.b_p_in);
    quad_trunk_to_xb<!kMid>(L, q, acc, x);
    if constexpr (kMid) {
        for (int l = 0; l < n_mid; ++l) {
            acc = quad_dense_b<quad_groups(H)>(W.p_mid, x, W.b_p_mid);
            quad_trunk_to_xb<!kMid>(L, q, acc, x);
        }
    }
    acc = quad_dense_b<quad_groups(H)>(W.p_h, x, W.b_p_h);
    const int i0 = 4 * q.eb, lim = q.half ? SD : NA;
    const bool valid[4] = {i0 < lim, i0 + 1 < lim, i0 + 2 < lim, i0 + 3 < lim};
    const f32x4 p = quad_softmax<!kMid>(acc, valid);
    *prior = p;
    const bool vv[4] = {valid[0] && q.half, valid[1] && q.half, valid[2] && q.half, valid[3] && q.half};
    return quad_support_mean<SD>(p, vv, i0);
}

template <int SD, int NA, int H, bool kMid>
__device__ __forceinline__ float quad_dynamics_b(const QuadWeightsB<SD, NA, H, kMid>& W, const QuadLds& L,
                                                 const QuadLane& q, const float* h, int action, float* hn,
                                                 int n_mid) {
    s16x4 x[4];
    quad_load_xb_f32(h, q.grp & 1, x);
    const float4 oh = *reinterpret_cast<const float4*>(L.onehot + action * 64 + q.n0);
    f32x4 acc = quad_dense_b<quad_groups(SD)>(W.d_in, x, W.b_d_in);
    acc = f32x4{acc[0] + oh.x, acc[1] + oh.y, acc[2] + oh.z, acc[3] + oh.w};
    quad_trunk_to_xb<!kMid>(L, q, acc, x);
    if constexpr (kMid) {
        for (int l = 0; l < n_mid; ++l) {
            acc = quad_dense_b<quad_groups(H)>(W.d_mid, x, W.b_d_mid);
            quad_trunk_to_xb<!kMid>(L, q, acc, x);
        }
    }
    acc = quad_dense_b<quad_groups(H)>(W.d_h, x, W.b_d_h);
    const int i0 = 4 * q.eb;
    const bool valid[4] = {i0 < SD, i0 + 1 < SD, i0 + 2 < SD, i0 + 3 < SD};
    const f32x4 p = quad_softmax<!kMid>(acc, valid);
    const float reward = quad_support_mean<SD>(p, valid, i0);
    const f32x4 y = quad_scale_to_bound<!kMid>(acc, valid);
    if (q.half) store4(hn + i0, y);
    wave_sync();
    return reward;
}

/* Representation_function on the observations staged in L.trunk ([game][64] floats, zero padded
 * to a multiple of 16); weights streamed from the rounded store (once per search) */
template <int SD, int NA, int H>
__device__ __forceinline__ void quad_representation_b(const Model& m, const QuadLds& L, const QuadLane& q,
                                                      float* h0) {
    const float* __restrict__ w = m.w;
    const int n = quad_neuron_of_lane(q.lane);
    const auto b4 = [&](int off) {
        const float4 v = *reinterpret_cast<const float4*>(w + off + q.n0);
        return f32x4{v.x, v.y, v.z, v.w};
    };
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    /* input layer: obs_dim <= 64 values per game, groups of four k, lane group g holds k 16g .. 16g+15 */
    s16x4 x[4];
    quad_load_xb_f32(L.trunk + q.c * kTrunkStride, q.grp, x);
    wave_sync();
    f32x4 e = b4(m.r_in.b_off), o = zero;
    const int ng = (m.obs_dim + 3) / 4;
    for (int g = 0; g < ng; ++g) {
        float v[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = 4 * g + t < m.obs_dim ? w[m.r_in.w_off + (4 * g + t) * 64 + n] : 0.0f;
        const s16x4 wg = as_s16x4(pack_hi(v[0], v[1]), pack_hi(v[2], v[3]));
        f32x4 d;
        switch (g >> 2) {
            case 0: d = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wg, x[g & 3], zero, 0, 0, 4); break;
            case 1: d = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wg, x[g & 3], zero, 0, 0, 5); break;
            case 2: d = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wg, x[g & 3], zero, 0, 0, 6); break;
            default: d = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wg, x[g & 3], zero, 0, 0, 7); break;
        }
        if (g & 1)
            o = g == 1 ? d : o + d;
        else
            e = e + d;
    }
    f32x4 acc = e + o;
    quad_trunk_to_xb(L, q, acc, x);
    s16x4 wk[quad_groups(H)];
    for (int l = 0; l < m.L; ++l) { /* tied mid layer, streamed like the rest of this net */
        quad_pack_rows<H>(w + m.r_mid.w_off, n, wk);
        acc = quad_dense_b<quad_groups(H)>(wk, x, b4(m.r_mid.b_off));
        quad_trunk_to_xb(L, q, acc, x);
    }
    quad_pack_rows<H>(w + m.r_out.w_off, n, wk);
    acc = quad_dense_b<quad_groups(H)>(wk, x, b4(m.r_out.b_off));
    const int i0 = 4 * q.eb;
    const bool valid[4] = {i0 < SD, i0 + 1 < SD, i0 + 2 < SD, i0 + 3 < SD};
    const f32x4 y = quad_scale_to_bound(acc, valid);
    if (!q.half) store4(h0 + i0, y);
    wave_sync();
}

/* ==================================================================================================
 * contract "dot_bf16x3" (include/mzs_numeric.h, MZS_PRECISION_BF16X3): weights and layer inputs as hi + lo
 * bfloat16 pairs, three v_mfma_f32_4x4x4_16B_bf16 per group of four k (hi hi, hi lo, lo hi), each with a zero
 * accumulator, four chains of fp32 adds.  Weights are split once at kernel start from the fp32 store (the same
 * register count as the fp32 kernel: one packed hi + one packed lo per two weights), activations when they
 * become a B operand.  Everything after the dot is the fp32 code of mzs_quad.h.
 * ================================================================================================ */

/* two floats -> their packed hi halves and packed lo halves (mzs_bf16_split on each) */
__device__ __forceinline__ void split_pack(float a, float b, unsigned& hi, unsigned& lo) {
    hi = pack_bf16(a, b);
    const float ah = __uint_as_float(hi << 16), bh = __uint_as_float(hi & 0xffff0000u);
    lo = pack_bf16(a - ah, b - bh);
}

template <int NG>
struct SplitRows {
    s16x4 h[NG], l[NG];
};
struct SplitX {
    s16x4 h[4], l[4];
};

template <int SD, int NA, int H, bool kMid = false>
struct QuadWeightsX3 {
    SplitRows<quad_groups(SD)> d_in, p_in;
    SplitRows<quad_groups(H)> d_h, p_h;
    SplitRows<kMid ? quad_groups(H) : 1> d_mid, p_mid;
    f32x4 b_d_in, b_d_h, b_p_in, b_p_h, b_d_mid, b_p_mid;
};

/* K weights of neuron n from the fp32 store W^T [k][64], split, packed in groups of four, zero past K */
template <int K>
__device__ __forceinline__ void quad_split_rows(const float* __restrict__ wt, int n, SplitRows<quad_groups(K)>& out) {
#pragma unroll
    for (int g = 0; g < quad_groups(K); ++g) {
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = 4 * g + q < K ? wt[(4 * g + q) * 64 + n] : 0.0f;
        unsigned h0, l0, h1, l1;
        split_pack(v[0], v[1], h0, l0);
        split_pack(v[2], v[3], h1, l1);
        out.h[g] = as_s16x4(h0, h1);
        out.l[g] = as_s16x4(l0, l1);
    }
}

template <int SD, int NA, int H, bool kMid>
__device__ __forceinline__ void quad_load_weights_x3(const Model& m, QuadWeightsX3<SD, NA, H, kMid>& W,
                                                     const QuadLane& q) {
    const float* __restrict__ w = m.w;
    const int n = quad_neuron_of_lane(q.lane);
    quad_split_rows<SD>(w + m.d_in.w_off, n, W.d_in);
    quad_split_rows<H>(w + m.d_heads.w_off, n, W.d_h);
    quad_split_rows<SD>(w + m.p_in.w_off, n, W.p_in);
    quad_split_rows<H>(w + m.p_heads.w_off, n, W.p_h);
    const auto b4 = [&](int off) {
        const float4 v = *reinterpret_cast<const float4*>(w + off + q.n0);
        return f32x4{v.x, v.y, v.z, v.w};
    };
    W.b_d_in = b4(m.d_in.b_off);
    W.b_d_h = b4(m.d_heads.b_off);
    W.b_p_in = b4(m.p_in.b_off);
    W.b_p_h = b4(m.p_heads.b_off);
    if constexpr (kMid) {
        quad_split_rows<H>(w + m.d_mid.w_off, n, W.d_mid);
        quad_split_rows<H>(w + m.p_mid.w_off, n, W.p_mid);
        W.b_d_mid = b4(m.d_mid.b_off);
        W.b_p_mid = b4(m.p_mid.b_off);
    }
}

/* B operands of lane group g from an fp32 row: 16 floats split into hi and lo halves */
__device__ __forceinline__ void quad_load_x3_f32(const float* row, int g, SplitX& x) {
    float t[16];
    quad_load_x(row, g, t);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned h0, l0, h1, l1;
        split_pack(t[4 * i], t[4 * i + 1], h0, l0);
        split_pack(t[4 * i + 2], t[4 * i + 3], h1, l1);
        x.h[i] = as_s16x4(h0, h1);
        x.l[i] = as_s16x4(l0, l1);
    }
}

template <int I>
__device__ __forceinline__ f32x4 quad_mfma_x(const s16x4 w, const s16x4 x) {
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    return __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(w, x, zero, 0, 0, 4 + (I >> 2));
}

template <int NG, int I = 0>
__device__ __forceinline__ void quad_dense_x3_step(const SplitRows<NG>& w, const SplitX& x, f32x4& e, f32x4& o,
                                                   f32x4& s, f32x4& t) {
    if constexpr (I < NG) {
        const f32x4 a = quad_mfma_x<I>(w.h[I], x.h[I & 3]);
        const f32x4 b = quad_mfma_x<I>(w.h[I], x.l[I & 3]);
        const f32x4 c = quad_mfma_x<I>(w.l[I], x.h[I & 3]);
        if constexpr (I & 1) {
            if constexpr (I == 1)
                o = a;
            else
                o = add4s(o, a);
        } else {
            e = add4s(e, a);
        }
        if constexpr (I == 0) {
            s = b;
            t = c;
        } else {
            s = add4s(s, b);
            t = add4s(t, c);
        }
        quad_dense_x3_step<NG, I + 1>(w, x, e, o, s, t);
    }
}

template <int NG>
__device__ __forceinline__ f32x4 quad_dense_x3(const SplitRows<NG>& w, const SplitX& x, const f32x4 bias) {
    f32x4 e = bias, o = f32x4{0.0f, 0.0f, 0.0f, 0.0f}, s = o, t = o;
    quad_dense_x3_step<NG>(w, x, e, o, s, t);
    return (e + o) + (s + t);
}

/* ELU(trunk) of the four games -> LDS as a row of hi halves and a row of lo halves (the lo rows 272 halves
 * behind the hi rows, inside the fp32 kernel's trunk buffer), back as the B operands of the next layer */
template <bool kPacked = true>
__device__ __forceinline__ void quad_trunk_to_x3(const QuadLds& L, const QuadLane& q, const f32x4 acc, SplitX& x) {
    const f32x4 t = elu4<kPacked>(acc);
    unsigned short* tr = reinterpret_cast<unsigned short*>(L.trunk);
    unsigned h0, l0, h1, l1;
    split_pack(t[0], t[1], h0, l0);
    split_pack(t[2], t[3], h1, l1);
    *reinterpret_cast<uint2*>(tr + quad_trunk_off_b(q.c) + q.n0) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(tr + 272 + quad_trunk_off_b(q.c) + q.n0) = make_uint2(l0, l1);
    wave_sync();
    quad_load_xb(tr + quad_trunk_off_b(q.c), q.grp, x.h);
    quad_load_xb(tr + 272 + quad_trunk_off_b(q.c), q.grp, x.l);
    wave_sync();
}

template <int SD, int NA, int H, bool kMid>
__device__ __forceinline__ float quad_prediction_x3(const QuadWeightsX3<SD, NA, H, kMid>& W, const QuadLds& L,
                                                    const QuadLane& q, const float* h, f32x4* prior, int n_mid) {
    SplitX x;
    quad_load_x3_f32(h, q.grp & 1, x);
    f32x4 acc = quad_dense_x3<quad_groups(SD)>(W.p_in, x, W.b_p_in);
    quad_trunk_to_x3<!kMid>(L, q, acc, x);
    if constexpr (kMid) {
        for (int l = 0; l < n_mid; ++l) {
            acc = quad_dense_x3<quad_groups(H)>(W.p_mid, x, W.b_p_mid);
            quad_trunk_to_x3<!kMid>(L, q, acc, x);
        }
    }
    acc = quad_dense_x3<quad_groups(H)>(W.p_h, x, W.b_p_h);
    const int i0 = 4 * q.eb, lim = q.half ? SD : NA;
    const bool valid[4] = {i0 < lim, i0 + 1 < lim, i0 + 2 < lim, i0 + 3 < lim};
    const f32x4 p = quad_softmax<!kMid>(acc, valid);
    *prior = p;
    const bool vv[4] = {valid[0] && q.half, valid[1] && q.half, valid[2] && q.half, valid[3] && q.half};
    return quad_support_mean<SD>(p, vv, i0);
}

template <int SD, int NA, int H, bool kMid>
__device__ __forceinline__ float quad_dynamics_x3(const QuadWeightsX3<SD, NA, H, kMid>& W, const QuadLds& L,
                                                  const QuadLane& q, const float* h, int action, float* hn,
                                                  int n_mid) {
    SplitX x;
    quad_load_x3_f32(h, q.grp & 1, x);
    const float4 oh = *reinterpret_cast<const float4*>(L.onehot + action * 64 + q.n0);
    f32x4 acc = quad_dense_x3<quad_groups(SD)>(W.d_in, x, W.b_d_in);
    acc = f32x4{acc[0] + oh.x, acc[1] + oh.y, acc[2] + oh.z, acc[3] + oh.w};
    quad_trunk_to_x3<!kMid>(L, q, acc, x);
    if constexpr (kMid) {
        for (int l = 0; l < n_mid; ++l) {
            acc = quad_dense_x3<quad_groups(H)>(W.d_mid, x, W.b_d_mid);
            quad_trunk_to_x3<!kMid>(L, q, acc, x);
        }
    }
    acc = quad_dense_x3<quad_groups(H)>(W.d_h, x, W.b_d_h);
    const int i0 = 4 * q.eb;
    const bool valid[4] = {i0 < SD, i0 + 1 < SD, i0 + 2 < SD, i0 + 3 < SD};
    const f32x4 p = quad_softmax<!kMid>(acc, valid);
    const float reward = quad_support_mean<SD>(p, valid, i0);
    const f32x4 y = quad_scale_to_bound<!kMid>(acc, valid);
    if (q.half) store4(hn + i0, y);
    wave_sync();
    return reward;
}

/* Representation_function, weights streamed from the fp32 store and split on the way (once per search) */
template <int SD, int NA, int H>
__device__ __forceinline__ void quad_representation_x3(const Model& m, const QuadLds& L, const QuadLane& q,
                                                       float* h0) {
    const float* __restrict__ w = m.w;
    const int n = quad_neuron_of_lane(q.lane);
    const auto b4 = [&](int off) {
        const float4 v = *reinterpret_cast<const float4*>(w + off + q.n0);
        return f32x4{v.x, v.y, v.z, v.w};
    };
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    SplitX x;
    quad_load_x3_f32(L.trunk + q.c * kTrunkStride, q.grp, x);
    wave_sync();
    f32x4 e = b4(m.r_in.b_off), o = zero, s = zero, t = zero;
    const int ng = (m.obs_dim + 3) / 4;
    for (int g = 0; g < ng; ++g) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = 4 * g + u < m.obs_dim ? w[m.r_in.w_off + (4 * g + u) * 64 + n] : 0.0f;
        unsigned h0u, l0u, h1u, l1u;
        split_pack(v[0], v[1], h0u, l0u);
        split_pack(v[2], v[3], h1u, l1u);
        const s16x4 wh = as_s16x4(h0u, h1u), wl = as_s16x4(l0u, l1u);
        f32x4 a, b, c;
        switch (g >> 2) {
            case 0:
                a = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wh, x.h[g & 3], zero, 0, 0, 4);
                b = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wh, x.l[g & 3], zero, 0, 0, 4);
                c = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wl, x.h[g & 3], zero, 0, 0, 4);
                break;
            case 1:
                a = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wh, x.h[g & 3], zero, 0, 0, 5);
                b = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wh, x.l[g & 3], zero, 0, 0, 5);
                c = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wl, x.h[g & 3], zero, 0, 0, 5);
                break;
            case 2:
                a = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wh, x.h[g & 3], zero, 0, 0, 6);
                b = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wh, x.l[g & 3], zero, 0, 0, 6);
                c = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wl, x.h[g & 3], zero, 0, 0, 6);
                break;
            default:
                a = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wh, x.h[g & 3], zero, 0, 0, 7);
                b = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wh, x.l[g & 3], zero, 0, 0, 7);
                c = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wl, x.h[g & 3], zero, 0, 0, 7);
                break;
        }
        if (g & 1)
            o = g == 1 ? a : o + a;
        else
            e = e + a;
        s = g == 0 ? b : s + b;
        t = g == 0 ? c : t + c;
    }
    f32x4 acc = (e + o) + (s + t);
    quad_trunk_to_x3(L, q, acc, x);
    SplitRows<quad_groups(H)> wk;
    for (int l = 0; l < m.L; ++l) {
        quad_split_rows<H>(w + m.r_mid.w_off, n, wk);
        acc = quad_dense_x3<quad_groups(H)>(wk, x, b4(m.r_mid.b_off));
        quad_trunk_to_x3(L, q, acc, x);
    }
    quad_split_rows<H>(w + m.r_out.w_off, n, wk);
    acc = quad_dense_x3<quad_groups(H)>(wk, x, b4(m.r_out.b_off));
    const int i0 = 4 * q.eb;
    const bool valid[4] = {i0 < SD, i0 + 1 < SD, i0 + 2 < SD, i0 + 3 < SD};
    const f32x4 y = quad_scale_to_bound(acc, valid);
    if (!q.half) store4(h0 + i0, y);
    wave_sync();
}

}  // namespace mzs
#endif
