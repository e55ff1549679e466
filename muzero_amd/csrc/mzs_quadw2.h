/*
 * mzs_quadw2.h — the wide-MLP search (experiments 127 / 128: S_dim 61, H 128) with the weight
 * stream SHARED by two wavefronts: a pair of waves owns eight games; each wave keeps the tree work
 * of its own four (mzs_quad.h, unchanged), but in the model half wave u computes output group u
 * (64 neurons) of every layer for all EIGHT games.  A weight chunk is then fetched by one wave of the
 * pair instead of both — half the L1 traffic that bound mzs_quadw.h's kernel (every wave fetching its
 * own copy of each 16 KB chunk: 1024 cycles of L1 time per chunk step next to 560 of matrix work) —
 * and multiplied against two sets of four games (two MFMAs per weight register).
 *
 *   activations   pair-shared LDS: xin [8][64] (hidden state / next hidden state), trunk [2][8][128]
 *                 (ping-pong: a layer reads one buffer while both waves write the other), exchanged
 *                 behind a block barrier per layer (seven per simulation at L = 1)
 *   heads         wave 0: reward (dynamics) and policy (prediction) of the eight games;
 *                 wave 1: next hidden state and value; scalars and priors return to the games'
 *                 owners through a small LDS table
 *   stream        k_build_stream lays the chunk images out per wave: wave u's 3 + 2 L units of the
 *                 dynamics net, then of the prediction net, after wave 0's (make_stream_plan)
 * Arithmetic: the same chains, epilogues and reductions per game as mzs_quadw.h — bit for bit the
 * contract of include/mzs_numeric.h (a game's result does not depend on which wave computed it).
 *
 * Every wave of the block reaches every barrier: waves whose games lie past n_games replay the last
 * game (as dead lanes do everywhere in the quad kernels) and write nothing.
 */
#ifndef MZS_QUADW2_H
#define MZS_QUADW2_H

#include "mzs_quadw.h"

namespace mzs {

/* row strides of the pair buffers: + 4 floats, so that the four game rows a ds_read_b128 of the B
 * operands touches (same columns, rows c = 0..3) fall into different LDS banks */
constexpr int kPairXStride = kWideXStride, kPairTStride = kWideTStride;
constexpr int kPairXin = 8 * kPairXStride;          /* floats */
constexpr int kPairTrunk = 8 * kPairTStride;        /* floats, one of the two ping-pong buffers */
constexpr int kPairXchg = 64;                       /* rew [8] | val [8] | act [8] | pad [8] | pri [8][4] */
constexpr int kPairFloats = kPairXin + 2 * kPairTrunk + kPairXchg;

/* block barrier between layers; the diagnostic build sums the cycles waited in slot 6 and the cycles
 * of the dense units in slot 7 */
#ifdef MZS_DIAG
#define PAIR_BARRIER()                                  \
    do {                                                \
        const unsigned long long b0_ = diag_stamp();    \
        __syncthreads();                                \
        dg[6] += diag_stamp() - b0_;                    \
    } while (0)
#define PAIR_UNIT_T0() const unsigned long long u0_ = diag_stamp()
#define PAIR_UNIT_T1() dg[7] += diag_stamp() - u0_
/* slot 0: waiting for the current chunk's weights and the x rows; slot 5: the matrix block */
#define PAIR_WAIT_T(n)                                                       \
    const unsigned long long w0_##n = diag_stamp();                          \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");              \
    unsigned long long w1_ = diag_stamp();                                   \
    dg[0] += w1_ - w0_##n
#define PAIR_MFMA_T() dg[5] += diag_stamp() - w1_
#else
#define PAIR_WAIT_T(n)
#define PAIR_MFMA_T()
#define PAIR_BARRIER() __syncthreads()
#define PAIR_UNIT_T0()
#define PAIR_UNIT_T1()
#endif

struct PairBufs {
    float* xin;      /* [8][kWideX]      */
    float* trunk0;   /* [2][8][kWideTrunk]: buffer i at trunk0 + i * kPairTrunk (plain pointer arithmetic on
                        purpose: an array of pointers indexed at run time costs the compiler the LDS address
                        space, and every read becomes a flat load that waits for the weight requests too) */
    float* rew;      /* [8] support expectation of the reward head  */
    float* val;      /* [8] support expectation of the value head   */
    int* act;        /* [8] action of the edge each game expands    */
    float* pri;      /* [8][4] priors                                */
};

__device__ __forceinline__ PairBufs pair_carve(float* base) {
    PairBufs P;
    P.xin = base;
    P.trunk0 = base + kPairXin;
    float* x = base + kPairXin + 2 * kPairTrunk;
    P.rew = x, P.val = x + 8, P.act = reinterpret_cast<int*>(x + 16), P.pri = x + 32;
    return P;
}

/* MFMA steps k = k0 + 16 kb + j of a chunk against both game sets */
template <int K, int k0, int kb>
__device__ __forceinline__ void pair_mfma16(const float (&wk)[64], const float (&x0)[16], const float (&x1)[16],
                                            f32x4 (&e)[2], f32x4 (&o)[2]) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if (k0 + 16 * kb + j < K) {
            if (j & 1) {
                o[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(wk[16 * kb + j], x0[j], o[0], 0, 0, 4 + kb);
                o[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wk[16 * kb + j], x1[j], o[1], 0, 0, 4 + kb);
            } else {
                e[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(wk[16 * kb + j], x0[j], e[0], 0, 0, 4 + kb);
                e[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wk[16 * kb + j], x1[j], e[1], 0, 0, 4 + kb);
            }
        }
    }
}

/* the matrix work of one chunk (rows k0 .. k0 + 63 of a layer with fan-in K, in cur) against both game
 * sets, with the sixteen requests of the stream's NEXT chunk (phase IN, KV rows) spread through it: four
 * ahead of each quarter.  With half the traffic of mzs_quadw.h the L1 is no longer saturated, so a
 * request between matrix blocks is accepted at once, where a burst of sixteen per wave (four waves at a
 * time) filled its queue and stalled the issuing wave ~1000 cycles per chunk. */
template <int K, int k0, int IN, int KV>
__device__ __forceinline__ void pair_block(WideStream& st, const float (&cur)[64], const float (&x0)[16],
                                           const float (&x1)[16], f32x4 (&e)[2], f32x4 (&o)[2]) {
    st.template load_part<IN, KV, 0>();
    __builtin_amdgcn_sched_barrier(0);
    pair_mfma16<K, k0, 0>(cur, x0, x1, e, o);
    __builtin_amdgcn_sched_barrier(0);
    st.template load_part<IN, KV, 1>();
    __builtin_amdgcn_sched_barrier(0);
    pair_mfma16<K, k0, 1>(cur, x0, x1, e, o);
    __builtin_amdgcn_sched_barrier(0);
    st.template load_part<IN, KV, 2>();
    __builtin_amdgcn_sched_barrier(0);
    pair_mfma16<K, k0, 2>(cur, x0, x1, e, o);
    __builtin_amdgcn_sched_barrier(0);
    st.template load_part<IN, KV, 3>();
    __builtin_amdgcn_sched_barrier(0);
    pair_mfma16<K, k0, 3>(cur, x0, x1, e, o);
    __builtin_amdgcn_sched_barrier(0);
}

/* one dense unit (this wave's 64-neuron group of a layer with fan-in K <= 128) for the pair's eight
 * games: x = the layer's input rows [8][stride] in pair LDS; out[s] = pre-activations of the lane's
 * four neurons for game row 4 s + c.  Stream phases as mzs_quadw.h wide_unit. */
template <int K, int I, int KVL = 64>
__device__ __forceinline__ void pair_unit(WideStream& st, const float* __restrict__ bias4, const float* x, int stride,
                                          const QuadLane& q, f32x4 (&out)[2], unsigned long long* dg = nullptr) {
    static_assert(K <= 128, "two chunks at most");
    const float4 bv = *reinterpret_cast<const float4*>(bias4);
    f32x4 e[2], o[2];
    e[0] = e[1] = f32x4{bv.x, bv.y, bv.z, bv.w};
    o[0] = o[1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const float* r0 = x + q.c * stride;
    const float* r1 = x + (4 + q.c) * stride;
    float x0[16], x1[16], y0[16], y1[16];
    quad_load_x(r0, q.grp, x0);
    quad_load_x(r1, q.grp, x1);
    if constexpr (K > 64) {
        quad_load_x(r0 + 64, q.grp, y0);
        quad_load_x(r1 + 64, q.grp, y1);
    }
    static_assert(kAhead == 1, "the requests of chunk c + 1 ride inside the matrix block of chunk c");
    __builtin_amdgcn_sched_barrier(0);
    PAIR_WAIT_T(0);
    pair_block<K, 0, I + 1, (K > 64 ? 64 : KVL)>(st, st.w[I % kBufs], x0, x1, e, o);
    PAIR_MFMA_T();
    if constexpr (K > 64) {
        PAIR_WAIT_T(1);
        pair_block<K, 64, I + 2, KVL>(st, st.w[(I + 1) % kBufs], y0, y1, e, o);
        PAIR_MFMA_T();
    }
    __builtin_amdgcn_sched_barrier(0);
    out[0] = e[0] + o[0];
    out[1] = e[1] + o[1];
}

/* this wave's half of a trunk layer + ELU for the eight games: reads `in` rows, writes columns
 * [64 u, 64 u + 64) of the `out` rows (a different buffer: the partner still reads `in`).
 * onehot: this wave's half of the [A][128] action rows, or null; act: the games' actions. */
template <int K, int I, class ST>
__device__ __forceinline__ void pair_trunk(ST& st, const float* bias_half, const float* onehot_half,
                                           const int* act, const float* in, int in_stride, float* out, int u,
                                           const QuadLane& q, unsigned long long* dg = nullptr) {
    f32x4 acc[2];
    PAIR_UNIT_T0();
    pair_unit<K, I>(st, bias_half + q.n0, in, in_stride, q, acc, dg);
    PAIR_UNIT_T1();
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        if (onehot_half) {
            const float4 c = *reinterpret_cast<const float4*>(onehot_half + act[4 * s + q.c] * 128 + q.n0);
            acc[s] = f32x4{acc[s][0] + c.x, acc[s][1] + c.y, acc[s][2] + c.z, acc[s][3] + c.w};
        }
        store4(out + (4 * s + q.c) * kPairTStride + 64 * u + q.n0, elu4<true>(acc[s]));
    }
}

/* Dynamics + prediction of one expansion for the pair's eight games.  In: the parents' hidden states
 * in P.xin, the actions in P.act (both written by the games' owners, barrier behind them).  Out: the
 * new hidden states in P.xin and in the global rows hdst[s] (wave 1 stores them), P.rew / P.val /
 * P.pri, barrier behind them.  kRoot: prediction only (the hidden states come from the
 * representation net).  Stream phase at entry: 0 (dynamics first) or 1 (kRoot). */
template <int SD, int H, int NA, bool kRoot, class ST>
__device__ __forceinline__ void pair_model(ST& st, const Model& m, const WideTabs& T, const PairBufs& P, int u,
                                           const QuadLane& q, float* const (&hdst)[2],
                                           unsigned long long* dg = nullptr) {
    static_assert(H > 64 && H <= 128 && SD <= 64, "two trunk groups, one chunk of state inputs");
    const WideValid vd = wide_valid(q, SD);
    int cur = 0; /* trunk buffer the next layer reads */
    if constexpr (!kRoot) {
        pair_trunk<SD, 0>(st, T.d_in_b + 64 * u, T.d_onehot + 64 * u, P.act, P.xin, kPairXStride, P.trunk0, u, q, dg);
        PAIR_BARRIER();
        for (int l = 0; l < m.L; ++l) {
            pair_trunk<H, 1>(st, T.d_mid_b + 64 * u, nullptr, nullptr, (P.trunk0 + cur * kPairTrunk), kPairTStride, (P.trunk0 + (cur ^ 1) * kPairTrunk), u, q, dg);
            cur ^= 1;
            PAIR_BARRIER();
        }
        f32x4 a[2];
        {
            PAIR_UNIT_T0();
            pair_unit<H, 1, SD>(st, T.d_heads_b + 64 * u + q.n0, (P.trunk0 + cur * kPairTrunk), kPairTStride, q, a, dg);
            PAIR_UNIT_T1();
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (u == 0) { /* reward head */
                const float yr = wide_support_mean(wide_softmax<true>(a[s], vd), vd, q, SD);
                if (q.b == 0) P.rew[4 * s + q.c] = yr;
            } else { /* next hidden state: P.xin is free (its last readers passed the first barrier) */
                const f32x4 y = wide_scale_to_bound(a[s], vd);
                store4(P.xin + (4 * s + q.c) * kPairXStride + q.n0, y);
                store4(hdst[s] + q.n0, y);
            }
        }
        PAIR_BARRIER();
    }
    /* prediction: stream phase 1 */
    pair_trunk<SD, 1>(st, T.p_in_b + 64 * u, nullptr, nullptr, P.xin, kPairXStride, P.trunk0, u, q, dg);
    cur = 0;
    PAIR_BARRIER();
    for (int l = 0; l < m.L; ++l) {
        pair_trunk<H, 0>(st, T.p_mid_b + 64 * u, nullptr, nullptr, (P.trunk0 + cur * kPairTrunk), kPairTStride, (P.trunk0 + (cur ^ 1) * kPairTrunk), u, q, dg);
        cur ^= 1;
        PAIR_BARRIER();
    }
    f32x4 a[2];
    {
        PAIR_UNIT_T0();
        pair_unit<H, 0, SD>(st, T.p_heads_b + 64 * u + q.n0, (P.trunk0 + cur * kPairTrunk), kPairTStride, q, a, dg);
        PAIR_UNIT_T1();
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        if (u == 0) { /* policy head */
            const f32x4 pr = wide_softmax<true>(a[s], wide_valid(q, NA));
            if (q.b == 0) store4(P.pri + 4 * (4 * s + q.c), pr);
        } else { /* value head */
            const float yv = wide_support_mean(wide_softmax<true>(a[s], vd), vd, q, SD);
            if (q.b == 0) P.val[4 * s + q.c] = yv;
        }
    }
    PAIR_BARRIER();
}

}  // namespace mzs
#endif
