/*
 * mzs_quadw.h — the four-games-per-wavefront search for MLPs too wide for register-resident
 * weights (experiments 127 / 128: S_dim 61, H 128, L 1 = 1268 weight rows per simulation).
 *
 * Same lane roles, tree code and arithmetic as mzs_quad.h (bit for bit: include/mzs_numeric.h);
 * what changes is where the operands live:
 *   A operand (weights)  streamed from L2: chunk images of the simulation's weights in consumption
 *                        order (k_build_stream; 320 KB for the shape above), four rows per
 *                        global_load_dwordx4, each row shared by the four games — a quarter of
 *                        the traffic of the one-game-per-wave generic kernel (WideStream below);
 *                        the root's representation net reads the store's W^T rows directly
 *   outputs              H_pad / 64 groups of 64 neurons, one f32x4 accumulator pair each
 *   hidden states        in HBM / L2 (the SoA output array the search fills anyway): 13 KB per
 *                        game would not fit LDS next to the trees of sixteen games
 *   heads                one 64-neuron group per head (reward | next state, policy | value at
 *                        column 64 of the fused head layers), reduced over all sixteen lanes of a
 *                        game: half_sum of each half, then the two halves — the contract's
 *                        adjacent-pair tree over a segment padded to 64
 * Compile-time S_dim (32 < S_dim <= 64), H (<= 128) and A, so that the fan-in tails generate no
 * guards and every unit knows its register buffer; obs_dim <= 64 and the number of tied mid
 * layers are runtime values.
 *
 * Reference code restated (file:line in /root/reference): as mzs_device.h.
 */
#ifndef MZS_QUADW_H
#define MZS_QUADW_H

#include "mzs_quad.h"

namespace mzs {

constexpr int kWideTrunk = 128; /* floats per game in the trunk buffer */
constexpr int kWideX = 64;      /* floats per game in the input (hidden state / observation) buffer */
/* row strides of those buffers: + 4 floats, so that the four games' rows of a ds_read_b128 of the B
 * operands fall into disjoint LDS banks (mzs_quad.h kTrunkStride) */
constexpr int kWideTStride = kWideTrunk + 4, kWideXStride = kWideX + 4;

struct WideBufs {
    float* xin;   /* [4][kWideXStride] */
    float* trunk; /* [4][kWideTStride] */
};

__device__ __forceinline__ float game_sum(float v) {
    v = half_sum(v);
    float lo, hi;
    halves_pair(v, lo, hi);
    return lo + hi;
}

/* MFMA steps k = k0 + 16 kb + j, j < 16, of a chunk whose weight rows sit in wk; K is the layer's
 * fan-in (compile time: rows past it do not exist and generate no code) */
template <int K, int k0, int kb>
__device__ __forceinline__ void wide_mfma16(const float (&wk)[64], const float (&x)[16], f32x4& e, f32x4& o) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if (k0 + 16 * kb + j < K) {
            if (j & 1)
                o = __builtin_amdgcn_mfma_f32_4x4x1f32(wk[16 * kb + j], x[j], o, 0, 0, 4 + kb);
            else
                e = __builtin_amdgcn_mfma_f32_4x4x1f32(wk[16 * kb + j], x[j], e, 0, 0, 4 + kb);
        }
    }
}

typedef const __attribute__((address_space(1))) float* WideGlobalPtr; /* keeps global_load (not flat) after the asm */

template <int K, int NPAD, int k0>
__device__ __forceinline__ void wide_load_chunk(WideGlobalPtr row0, unsigned n, float (&wk)[64]) {
#pragma unroll
    for (int j = 0; j < 64; ++j) {
        /* scalar base + zero-extended 32-bit lane offset: the saddr form of global_load */
        WideGlobalPtr rk = row0 + (size_t)(k0 + j) * NPAD;
        if (k0 + j < K) wk[j] = rk[n];
    }
}

/* ---- the weight stream of the simulation loop --------------------------------------------
 * The weights one simulation consumes are laid out in consumption order as 16 KB chunk images
 * (k_build_stream, before every search): 64 rows of one 64-neuron output group in the layout
 * [k / 4][neuron][4], so a lane fetches four consecutive k of its neuron with one
 * global_load_dwordx4 and the wave's 64 lanes cover one contiguous KiB — 16 vector-memory
 * instructions per chunk.  (From the store's W^T rows it takes 64 global_load_dword per chunk, and
 * their issue — not cache bandwidth — is then what the wave waits for: ~1100 stall cycles per
 * chunk next to 560 cycles of matrix work.  Tried and dropped as well: one copy per block in an
 * LDS ring filled by LDS-DMA)
 * Chunk c lives in register buffer c % kBufs; the step that multiplies chunk c first requests
 * chunk c + kAhead into the buffer chunk c - 1 has left — across groups, layers, nets and
 * simulations.  The chunks per net are 2 mod 4 (MLP) or 0 mod 4 (LSTM heads), so the buffer a unit
 * reads is a compile-time constant (its phase I).  kAhead = 3 (four buffers) measured 1 % faster
 * than 1: with 16 requests per chunk the wave no longer waits for latency but for the L1's
 * 64 B/clk — four waves x 16 KB per step = 1024 cycles next to 560 of matrix work. */
constexpr int kImageFloats = 64 * 64;
constexpr int kMaxImages = 128;
constexpr int kAhead = 1; /* chunks requested ahead of the one being multiplied (1 or 3) */
constexpr int kBufs = kAhead + 1;

struct WideStream {
    WideGlobalPtr img; /* image 0 + this lane's neuron */
    int n_img, next_img;
    float w[kBufs][64];
    /* images: the handle's image buffer; image0: index of the first image of this wave's share */
    __device__ __forceinline__ void init(const float* images, int image0, int n_images, int lane) {
        img = (WideGlobalPtr)images + (size_t)image0 * kImageFloats + 4 * quad_neuron_of_lane(lane);
        n_img = n_images;
        next_img = 0;
    }
    /* request the next image's first KV rows into buffer I % kBufs */
    template <int I, int KV = 64>
    __device__ __forceinline__ void load() {
        typedef const __attribute__((address_space(1))) f32x4* P4;
        P4 b = (P4)(img + (size_t)next_img * kImageFloats);
        float(&wk)[64] = w[I % kBufs];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (4 * j < KV) {
                const f32x4 v = b[j * 64];
                wk[4 * j] = v[0], wk[4 * j + 1] = v[1], wk[4 * j + 2] = v[2], wk[4 * j + 3] = v[3];
            }
        }
        next_img = next_img + 1 == n_img ? 0 : next_img + 1;
    }
    /* a quarter of load<I, KV>: four of the sixteen requests (k rows 16 part .. 16 part + 15); the
     * fourth quarter moves on to the next image */
    template <int I, int KV, int part>
    __device__ __forceinline__ void load_part() {
        typedef const __attribute__((address_space(1))) f32x4* P4;
        P4 b = (P4)(img + (size_t)next_img * kImageFloats);
        float(&wk)[64] = w[I % kBufs];
#pragma unroll
        for (int j = 4 * part; j < 4 * part + 4; ++j) {
            if (4 * j < KV) {
                const f32x4 v = b[j * 64];
                wk[4 * j] = v[0], wk[4 * j + 1] = v[1], wk[4 * j + 2] = v[2], wk[4 * j + 3] = v[3];
            }
        }
        if constexpr (part == 3) next_img = next_img + 1 == n_img ? 0 : next_img + 1;
    }
    /* the stream starts at image `first` (KV0 rows used) in phase I */
    template <int I, int KV0>
    __device__ __forceinline__ void start(int first) {
        next_img = first;
        load<I, KV0>();
        if constexpr (kAhead > 1) load<I + 1>();
        if constexpr (kAhead > 2) load<I + 2>();
    }
};

/* where the small per-unit vectors of the simulation loop live: the block's LDS copy made at
 * kernel start */
struct WideTabs {
    const float *d_in_b, *d_mid_b, *d_heads_b, *p_in_b, *p_mid_b, *p_heads_b; /* biases [128] */
    const float* d_onehot;                                                    /* [A][128]: the action rows of d_in */
};
constexpr int kWideTabFloats = 6 * 128; /* + A * 128 */

/* chunk k0 (its weight rows already requested into wk) and the chunks after it: the next chunk's
 * 64 rows are requested before this chunk's MFMAs, so their latency overlaps 512 cycles of matrix
 * work (one wave per SIMD hides nothing by itself) */
template <int K, int NPAD, int k0>
__device__ __forceinline__ void wide_chunks(WideGlobalPtr row0, unsigned n, const float* xrow,
                                            const QuadLane& q, float (&wk)[64], f32x4& e, f32x4& o) {
    if constexpr (k0 < K) {
        float wn[64];
        if constexpr (k0 + 64 < K) wide_load_chunk<K, NPAD, k0 + 64>(row0, n, wn);
        float x[16];
        quad_load_x(xrow + k0, q.grp, x);
        wide_mfma16<K, k0, 0>(wk, x, e, o);
        wide_mfma16<K, k0, 1>(wk, x, e, o);
        wide_mfma16<K, k0, 2>(wk, x, e, o);
        wide_mfma16<K, k0, 3>(wk, x, e, o);
        __builtin_amdgcn_sched_barrier(0);
        wide_chunks<K, NPAD, k0 + 64>(row0, n, xrow, q, wn, e, o);
    }
}

/* contract "dot" of output group `grp` of a layer (W^T at wl, row stride NPAD, bias at bl) for the
 * four games: K inputs from the game's LDS row xrow (a buffer of at least round_up(K, 64) floats) */
template <int K, int NPAD>
__device__ __forceinline__ f32x4 wide_dense(const float* __restrict__ wl, const float* __restrict__ bl, int grp,
                                            const float* xrow, const QuadLane& q) {
    const float4 bv = *reinterpret_cast<const float4*>(bl + 64 * grp + q.n0);
    f32x4 e = f32x4{bv.x, bv.y, bv.z, bv.w}, o = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    /* the layer base stays a scalar the optimiser cannot look through: otherwise it hoists the
     * address of every weight row of every layer out of the simulation loop (as loop invariants,
     * in vector register pairs) and spills them */
    WideGlobalPtr base = (WideGlobalPtr)(wl + 64 * grp);
    asm volatile("" : "+s"(base));
    const unsigned n = (unsigned)quad_neuron_of_lane(q.lane);
    float wk[64];
    wide_load_chunk<K, NPAD, 0>(base, n, wk);
    wide_chunks<K, NPAD, 0>(base, n, xrow, q, wk, e, o);
    return e + o;
}

/* trunk layer (all NPAD / 64 output groups) + ELU -> the game's trunk row.  extra_row >= 0: add
 * that row of W^T after the chains (the one-hot action column, per game).  The accumulators of
 * every group are complete before the row is overwritten (the mid layer reads and writes the
 * same buffer). */
template <int K, int NPAD>
__device__ __forceinline__ void wide_trunk(const float* __restrict__ w, const Layer ly, const float* xrow,
                                           int extra_row, float* trunk_row, const QuadLane& q) {
    constexpr int NG = NPAD / 64;
    f32x4 acc[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        acc[g] = wide_dense<K, NPAD>(w + ly.w_off, w + ly.b_off, g, xrow, q);
        if (extra_row >= 0) {
            const float4 c = *reinterpret_cast<const float4*>(w + ly.w_off + (size_t)extra_row * NPAD + 64 * g + q.n0);
            acc[g] = f32x4{acc[g][0] + c.x, acc[g][1] + c.y, acc[g][2] + c.z, acc[g][3] + c.w};
        }
    }
    wave_sync();
#pragma unroll
    for (int g = 0; g < NG; ++g) store4(trunk_row + 64 * g + q.n0, elu4<false>(acc[g]));
    wave_sync();
}

/* the matrix work of one chunk: rows k0 .. k0+63 of a layer with fan-in K, in cur */
template <int K, int k0>
__device__ __forceinline__ void wide_block(const float (&cur)[64], const float (&x)[16], f32x4& e, f32x4& o) {
    wide_mfma16<K, k0, 0>(cur, x, e, o);
    wide_mfma16<K, k0, 1>(cur, x, e, o);
    wide_mfma16<K, k0, 2>(cur, x, e, o);
    wide_mfma16<K, k0, 3>(cur, x, e, o);
    __builtin_amdgcn_sched_barrier(0);
}

/* wide_block with the stream's next chunk (phase IN, KV rows) requested inside it */
template <int K, int k0, int IN, int KV>
__device__ __forceinline__ void wide_block_ld(WideStream& st, const float (&cur)[64], const float (&x)[16], f32x4& e,
                                              f32x4& o) {
    st.template load_part<IN, KV, 0>();
    __builtin_amdgcn_sched_barrier(0);
    wide_mfma16<K, k0, 0>(cur, x, e, o);
    __builtin_amdgcn_sched_barrier(0);
    st.template load_part<IN, KV, 1>();
    __builtin_amdgcn_sched_barrier(0);
    wide_mfma16<K, k0, 1>(cur, x, e, o);
    __builtin_amdgcn_sched_barrier(0);
    st.template load_part<IN, KV, 2>();
    __builtin_amdgcn_sched_barrier(0);
    wide_mfma16<K, k0, 2>(cur, x, e, o);
    __builtin_amdgcn_sched_barrier(0);
    st.template load_part<IN, KV, 3>();
    __builtin_amdgcn_sched_barrier(0);
    wide_mfma16<K, k0, 3>(cur, x, e, o);
    __builtin_amdgcn_sched_barrier(0);
}

/* one dense unit (one 64-neuron output group of a layer with fan-in K <= 128) on the stream: its
 * first chunk is in buffer I % kBufs; KVL = the rows needed of the chunk its last step requests (the
 * next unit's first) */
template <int K, int I, int KVL = 64>
__device__ __forceinline__ f32x4 wide_unit(WideStream& st, const float* __restrict__ bias4, const float* xrow,
                                           const QuadLane& q) {
    static_assert(K <= 128, "two chunks at most");
    const float4 bv = *reinterpret_cast<const float4*>(bias4);
    f32x4 e = f32x4{bv.x, bv.y, bv.z, bv.w}, o = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    float x[16], x2[16];
    quad_load_x(xrow, q.grp, x);
    if constexpr (K > 64) quad_load_x(xrow + 64, q.grp, x2);
    /* the sixteen requests of the next chunk ride inside the matrix block, four ahead of each quarter */
    static_assert(kAhead == 1, "chunk c + 1 is requested inside the matrix block of chunk c");
    __builtin_amdgcn_sched_barrier(0);
    wide_block_ld<K, 0, I + 1, (K > 64 ? 64 : KVL)>(st, st.w[I % kBufs], x, e, o);
    if constexpr (K > 64) wide_block_ld<K, 64, I + 2, KVL>(st, st.w[(I + 1) % kBufs], x2, e, o);
    __builtin_amdgcn_sched_barrier(0);
    return e + o;
}

/* trunk layer with two output groups (H <= 128) + ELU -> the game's trunk row; extra: the one-hot
 * action row of this lane's game (added after the chains) or null.  The accumulators of both
 * groups are complete before the row is overwritten (the mid layer reads and writes it). */
template <int K, int I>
__device__ __forceinline__ void wide_trunk_pf(WideStream& st, const float* bias_row, const float* extra,
                                              const float* xrow, float* trunk_row, const QuadLane& q) {
    constexpr int I1 = K > 64 ? I + 2 : I + 1;
    const float* bias = bias_row + q.n0;
    f32x4 acc[2];
    float4 c[2];
    if (extra) {
        c[0] = *reinterpret_cast<const float4*>(extra + q.n0);
        c[1] = *reinterpret_cast<const float4*>(extra + 64 + q.n0);
    }
    acc[0] = wide_unit<K, I>(st, bias, xrow, q);
    acc[1] = wide_unit<K, I1>(st, bias + 64, xrow, q);
    if (extra) {
#pragma unroll
        for (int g = 0; g < 2; ++g)
            acc[g] = f32x4{acc[g][0] + c[g].x, acc[g][1] + c[g].y, acc[g][2] + c[g].z, acc[g][3] + c[g].w};
    }
    wave_sync();
#pragma unroll
    for (int g = 0; g < 2; ++g) store4(trunk_row + 64 * g + q.n0, elu4<false>(acc[g]));
    wave_sync();
}

struct WideValid {
    bool v[4];
};
__device__ __forceinline__ WideValid wide_valid(const QuadLane& q, int n) {
    WideValid r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.v[i] = q.n0 + i < n;
    return r;
}

/* softmax over the n valid logits of a 64-neuron group (contract exp + sum + shared-reciprocal quotient) */
template <bool kPacked = false>
__device__ __forceinline__ f32x4 wide_softmax(const f32x4 o, const WideValid& vd) {
    const float ninf = -__builtin_inff();
    const float m = game_max(max4(f32x4{vd.v[0] ? o[0] : ninf, vd.v[1] ? o[1] : ninf, vd.v[2] ? o[2] : ninf,
                                        vd.v[3] ? o[3] : ninf}));
    f32x4 e = exp4<kPacked>(o - splat4(m));
#pragma unroll
    for (int r = 0; r < 4; ++r) e[r] = vd.v[r] ? e[r] : 0.0f;
    const float Z = game_sum(sum4(e));
    return div_r2_4<kPacked>(e, Z, 1.0f / Z);
}

/* expectation over the support i - n/2 (before the inverse value transform) */
__device__ __forceinline__ float wide_support_mean(const f32x4 p, const WideValid& vd, const QuadLane& q, int n) {
    f32x4 t;
#pragma unroll
    for (int r = 0; r < 4; ++r) t[r] = vd.v[r] ? (float)(q.n0 + r - n / 2) * p[r] : 0.0f;
    return game_sum(sum4(t));
}

__device__ __forceinline__ f32x4 wide_scale_to_bound(const f32x4 o, const WideValid& vd) {
    const float ninf = -__builtin_inff();
    const float mx = game_max(max4(f32x4{vd.v[0] ? o[0] : ninf, vd.v[1] ? o[1] : ninf, vd.v[2] ? o[2] : ninf,
                                         vd.v[3] ? o[3] : ninf}));
    const float mn = -game_max(max4(f32x4{vd.v[0] ? -o[0] : ninf, vd.v[1] ? -o[1] : ninf, vd.v[2] ? -o[2] : ninf,
                                          vd.v[3] ? -o[3] : ninf}));
    const float sc = mzs_bound_scale(mn, mx), rs = 1.0f / sc;
    f32x4 y = div_r2_4<false>(o - splat4(mn), sc, rs);
#pragma unroll
    for (int r = 0; r < 4; ++r) y[r] = vd.v[r] ? y[r] : 0.0f;
    return y;
}

/* the tied mid layer, m.L times, in place on the trunk row */
template <int H>
__device__ __forceinline__ void wide_mid(const Model& m, const Layer ly, float* trunk_row, const QuadLane& q) {
    constexpr int HP = (H + 63) / 64 * 64;
    for (int l = 0; l < m.L; ++l) wide_trunk<H, HP>(m.w, ly, trunk_row, -1, trunk_row, q);
}

/* Representation_function: observation in the game's xin row -> hidden state in the xin row and in
 * the global row hdst.  Root only: the observation width stays a runtime value (padded to 64 zero
 * inputs, whose products add +0 to chains that start at a bias: no effect on any bit unless a bias
 * is -0, which weights_init and training never produce). */
template <int SD, int H>
__device__ __forceinline__ void wide_representation(const Model& m, const WideBufs& B, const QuadLane& q, float* hdst) {
    constexpr int HP = (H + 63) / 64 * 64;
    float* xrow = B.xin + q.c * kWideXStride;
    float* trow = B.trunk + q.c * kWideTStride;
    {
        constexpr int NG = HP / 64;
        f32x4 acc[NG];
        const int n = quad_neuron_of_lane(q.lane);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const float4 bv = *reinterpret_cast<const float4*>(m.w + m.r_in.b_off + 64 * g + q.n0);
            f32x4 e = f32x4{bv.x, bv.y, bv.z, bv.w}, o = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            for (int k = 0; k < m.obs_dim; ++k) {
                const float wk = m.w[m.r_in.w_off + (size_t)k * HP + 64 * g + n], xk = xrow[k];
                if (k & 1)
                    o = __builtin_amdgcn_mfma_f32_4x4x1f32(wk, xk, o, 0, 0, 0);
                else
                    e = __builtin_amdgcn_mfma_f32_4x4x1f32(wk, xk, e, 0, 0, 0);
            }
            acc[g] = e + o;
        }
        wave_sync();
#pragma unroll
        for (int g = 0; g < NG; ++g) store4(trow + 64 * g + q.n0, elu4<false>(acc[g]));
        wave_sync();
    }
    wide_mid<H>(m, m.r_mid, trow, q);
    const f32x4 acc = wide_dense<H, 64>(m.w + m.r_out.w_off, m.w + m.r_out.b_off, 0, trow, q);
    const f32x4 y = wide_scale_to_bound(acc, wide_valid(q, SD));
    wave_sync();
    store4(xrow + q.n0, y);
    store4(hdst + q.n0, y);
    wave_sync();
}

/* Dynamics_function on the hidden state in the game's xin row + action: the new hidden state
 * replaces it (and goes to the global row hdst); returns the reward's support expectation.
 * Stream phase I (an even number of chunks: 2 + 4 L + 4). */
template <int SD, int H, int I>
__device__ __forceinline__ float wide_dynamics(WideStream& st, const Model& m, const WideTabs& T, const WideBufs& Bf,
                                               const QuadLane& q, int action, float* hdst) {
    static_assert(H > 64 && H <= 128 && SD <= 64, "two trunk groups, one chunk of state inputs");
    float* xrow = Bf.xin + q.c * kWideXStride;
    float* trow = Bf.trunk + q.c * kWideTStride;
    wide_trunk_pf<SD, I>(st, T.d_in_b, T.d_onehot + action * 128, xrow, trow, q);
    for (int l = 0; l < m.L; ++l) wide_trunk_pf<H, I + 2>(st, T.d_mid_b, nullptr, trow, trow, q);
    const WideValid vd = wide_valid(q, SD);
    const float* hb = T.d_heads_b + q.n0;
    const f32x4 ar = wide_unit<H, I + 2>(st, hb, trow, q);
    const float reward = wide_support_mean(wide_softmax(ar, vd), vd, q, SD);
    const f32x4 as = wide_unit<H, I + 4>(st, hb + 64, trow, q);
    const f32x4 y = wide_scale_to_bound(as, vd);
    wave_sync();
    store4(xrow + q.n0, y);
    store4(hdst + q.n0, y);
    wave_sync();
    return reward;
}

/* Prediction_function on the hidden state in the game's xin row: priors of actions 0..3 in the
 * lanes with n0 == 0; returns the value's support expectation. */
template <int SD, int H, int I>
__device__ __forceinline__ float wide_prediction(WideStream& st, const Model& m, const WideTabs& T, const WideBufs& Bf,
                                                 const QuadLane& q, int n_actions, f32x4* prior) {
    float* xrow = Bf.xin + q.c * kWideXStride;
    float* trow = Bf.trunk + q.c * kWideTStride;
    wide_trunk_pf<SD, I>(st, T.p_in_b, nullptr, xrow, trow, q);
    for (int l = 0; l < m.L; ++l) wide_trunk_pf<H, I + 2>(st, T.p_mid_b, nullptr, trow, trow, q);
    const float* hb = T.p_heads_b + q.n0;
    const f32x4 ap = wide_unit<H, I + 2>(st, hb, trow, q);
    *prior = wide_softmax(ap, wide_valid(q, n_actions));
    const WideValid vd = wide_valid(q, SD);
    const f32x4 av = wide_unit<H, I + 4>(st, hb + 64, trow, q);
    return wide_support_mean(wide_softmax(av, vd), vd, q, SD);
}

}  // namespace mzs
#endif
