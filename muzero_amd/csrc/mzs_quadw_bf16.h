/*
 * mzs_quadw_bf16.h — the weight stream of mzs_quadw.h / mzs_quadw2.h / mzs_quadl.h under contract
 * "dot_bf16" (include/mzs_numeric.h, MZS_PRECISION_BF16): chunk images hold bfloat16 weights, the dense
 * units run on v_mfma_f32_4x4x4_16B_bf16 with a zero accumulator and add the group results in fp32.
 *
 * What that buys the kernels the stream binds (wide MLP, LSTM heads): a chunk (64 weight rows x 64 neurons)
 * is 8 KB instead of 16 — half the L1 / L2 traffic — fetched with 8 requests per lane instead of 16 into 32
 * registers instead of 64, and multiplied with 16 matrix instructions per set of four games instead of 64.
 *
 *   image layout   [k / 8][neuron][8] halves: a lane fetches eight consecutive k of its neuron with one
 *                  global_load_dwordx4 and the wave's 64 lanes cover one contiguous KiB (k_build_stream_b)
 *   A operand      two consecutive registers of the chunk buffer = four consecutive k, packed
 *   B operand      the layer's input row (fp32 in LDS, as in the fp32 kernels) rounded to bfloat16 when it is
 *                  loaded: lane group g holds x[16 g .. 16 g + 15], BLGP = 4 + g broadcasts it
 *   chains         group 4 i + j of a chunk: even groups into the chain that starts at the bias, odd groups
 *                  into the chain that starts AT group 1 of the layer; one add at the end
 * Everything behind the dot (bias order, one-hot action rows, ELU, softmax, scale_to_bound, the LSTM cell)
 * is the code of the fp32 kernels; a game's result equals the bf16 oracle's bit for bit.
 */
#ifndef MZS_QUADW_BF16_H
#define MZS_QUADW_BF16_H

#include "mzs_quad_bf16.h"
#include "mzs_quadw.h"

namespace mzs {

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
constexpr int kImageU4 = 64 * 64 * 2 / 16; /* 16-byte units of one bf16 chunk image (8 KB) */

struct WideStreamB {
    typedef const __attribute__((address_space(1))) u32x4_t* P4;
    P4 img; /* image 0 + this lane's neuron */
    int n_img, next_img;
    unsigned w[kBufs][32];
    /* images: the handle's image buffer; image0: index of the first image of this wave's share */
    __device__ __forceinline__ void init(const float* images, int image0, int n_images, int lane) {
        img = (P4)images + (size_t)image0 * kImageU4 + quad_neuron_of_lane(lane);
        n_img = n_images;
        next_img = 0;
    }
    /* request j of a chunk: k rows 8 j .. 8 j + 7 */
    template <int I, int j>
    __device__ __forceinline__ void request() {
        const u32x4_t v = (img + (size_t)next_img * kImageU4)[j * 64];
        unsigned(&wk)[32] = w[I % kBufs];
        wk[4 * j] = v[0], wk[4 * j + 1] = v[1], wk[4 * j + 2] = v[2], wk[4 * j + 3] = v[3];
    }
    __device__ __forceinline__ void advance() { next_img = next_img + 1 == n_img ? 0 : next_img + 1; }
    /* a quarter of the next image's requests (k rows 16 part .. 16 part + 15) into buffer I % kBufs; the
     * fourth quarter moves on to the image after it */
    template <int I, int KV, int part>
    __device__ __forceinline__ void load_part() {
        if constexpr (16 * part < KV) request<I, 2 * part>();
        if constexpr (16 * part + 8 < KV) request<I, 2 * part + 1>();
        if constexpr (part == 3) advance();
    }
    template <int I, int KV = 64>
    __device__ __forceinline__ void load() {
        load_part<I, KV, 0>();
        load_part<I, KV, 1>();
        load_part<I, KV, 2>();
        load_part<I, KV, 3>();
    }
    /* the stream starts at image `first` (KV0 rows used) in phase I */
    template <int I, int KV0>
    __device__ __forceinline__ void start(int first) {
        static_assert(kAhead == 1, "one chunk ahead");
        next_img = first;
        load<I, KV0>();
    }
};

/* fp32 row -> the B operand registers of this lane's group (16 values rounded to bfloat16) */
struct WideXB {
    s16x4 v[4];
};
__device__ __forceinline__ WideXB wide_load_xb(const float* row, int grp) {
    WideXB x;
    quad_load_xb_f32(row, grp, x.v);
    return x;
}

/* groups 4 part .. 4 part + 3 of the chunk at k0 against NS sets of four games */
template <int K, int k0, int part, int NS>
__device__ __forceinline__ void wide_quarter_b(const unsigned (&cur)[32], const WideXB (&x)[NS], f32x4 (&e)[NS],
                                               f32x4 (&o)[NS]) {
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        const int gi = 4 * part + gq; /* group within the chunk: k0 + 4 gi .. + 3 */
        if (k0 + 4 * gi < K) {
            const s16x4 wv = as_s16x4(cur[2 * gi], cur[2 * gi + 1]);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const f32x4 d = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wv, x[s].v[gq], zero, 0, 0, 4 + part);
                if (gi & 1) {
                    if (k0 == 0 && gi == 1)
                        o[s] = d;
                    else
                        o[s] = o[s] + d;
                } else {
                    e[s] = e[s] + d;
                }
            }
        }
    }
}

/* the matrix work of one chunk with the requests of the stream's next chunk (phase IN, KV rows) spread
 * through it, two ahead of each quarter */
constexpr bool kSpreadRequestsB = true;  /* true: two requests ahead of each quarter (tag 127: 0.603 ms); false: all eight ahead of the block (0.622) */
template <int K, int k0, int IN, int KV, int NS>
__device__ __forceinline__ void wide_block_b(WideStreamB& st, const unsigned (&cur)[32], const WideXB (&x)[NS],
                                             f32x4 (&e)[NS], f32x4 (&o)[NS]) {
    if constexpr (!kSpreadRequestsB) {
        st.template load<IN, KV>();
        __builtin_amdgcn_sched_barrier(0);
        wide_quarter_b<K, k0, 0, NS>(cur, x, e, o);
        wide_quarter_b<K, k0, 1, NS>(cur, x, e, o);
        wide_quarter_b<K, k0, 2, NS>(cur, x, e, o);
        wide_quarter_b<K, k0, 3, NS>(cur, x, e, o);
        __builtin_amdgcn_sched_barrier(0);
        return;
    }
    st.template load_part<IN, KV, 0>();
    __builtin_amdgcn_sched_barrier(0);
    wide_quarter_b<K, k0, 0, NS>(cur, x, e, o);
    __builtin_amdgcn_sched_barrier(0);
    st.template load_part<IN, KV, 1>();
    __builtin_amdgcn_sched_barrier(0);
    wide_quarter_b<K, k0, 1, NS>(cur, x, e, o);
    __builtin_amdgcn_sched_barrier(0);
    st.template load_part<IN, KV, 2>();
    __builtin_amdgcn_sched_barrier(0);
    wide_quarter_b<K, k0, 2, NS>(cur, x, e, o);
    __builtin_amdgcn_sched_barrier(0);
    st.template load_part<IN, KV, 3>();
    __builtin_amdgcn_sched_barrier(0);
    wide_quarter_b<K, k0, 3, NS>(cur, x, e, o);
    __builtin_amdgcn_sched_barrier(0);
}

/* diagnostic build: the slots of mzs_quadw2.h PAIR_WAIT_T / PAIR_MFMA_T (0: waiting for the chunk's weights and
 * the x rows, 5: the matrix block) */
#ifdef MZS_DIAG
#define X3_WAIT_T(n)                                            \
    const unsigned long long xw0_##n = diag_stamp();            \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
    const unsigned long long xw1_##n = diag_stamp();            \
    if (dg) dg[0] += xw1_##n - xw0_##n
#define X3_MFMA_T(n) \
    if (dg) dg[5] += diag_stamp() - xw1_##n
#else
#define X3_WAIT_T(n)
#define X3_MFMA_T(n)
#endif

/* one dense unit for NS sets of four games: rows[s] = the set's input rows in LDS (row of game c at
 * rows[s] + c * stride); out[s] = pre-activations of the lane's four neurons.  Stream phases as wide_unit. */
template <int K, int I, int KVL, int NS>
__device__ __forceinline__ void wide_unit_sets_b(WideStreamB& st, const float* __restrict__ bias4,
                                                 const float* const (&rows)[NS], int stride, const QuadLane& q,
                                                 f32x4 (&out)[NS], unsigned long long* dg = nullptr) {
    static_assert(K <= 128, "two chunks at most");
    const float4 bv = *reinterpret_cast<const float4*>(bias4);
    f32x4 e[NS], o[NS];
    WideXB x[NS], y[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        e[s] = f32x4{bv.x, bv.y, bv.z, bv.w};
        o[s] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        x[s] = wide_load_xb(rows[s] + q.c * stride, q.grp);
        if constexpr (K > 64) y[s] = wide_load_xb(rows[s] + q.c * stride + 64, q.grp);
    }
    __builtin_amdgcn_sched_barrier(0);
    X3_WAIT_T(0);
    wide_block_b<K, 0, I + 1, (K > 64 ? 64 : KVL), NS>(st, st.w[I % kBufs], x, e, o);
    X3_MFMA_T(0);
    if constexpr (K > 64) {
        X3_WAIT_T(1);
        wide_block_b<K, 64, I + 2, KVL, NS>(st, st.w[(I + 1) % kBufs], y, e, o);
        X3_MFMA_T(1);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < NS; ++s) out[s] = e[s] + o[s];
    /* not pinned as wide_unit_sets_x pins its result: here the sunk adds (64 products at most) fit the register
     * file, and the matrix instructions issuing back to back ahead of them measured faster (tag 127: 0.616 ms
     * against 0.677 pinned) */
}

/* the overloads the shared model code of mzs_quadw.h / mzs_quadw2.h / mzs_quadl.h resolves to when it is
 * handed a WideStreamB */
template <int K, int I, int KVL = 64>
__device__ __forceinline__ f32x4 wide_unit(WideStreamB& st, const float* __restrict__ bias4, const float* xrow,
                                           const QuadLane& q) {
    const float* const rows[1] = {xrow}; /* already this lane's game row: stride 0 */
    f32x4 out[1];
    wide_unit_sets_b<K, I, KVL, 1>(st, bias4, rows, 0, q, out);
    return out[0];
}

/* ... and the pair kernel's unit: two sets of four games (rows 0..3 and 4..7 of x) */
template <int K, int I, int KVL = 64>
__device__ __forceinline__ void pair_unit(WideStreamB& st, const float* __restrict__ bias4, const float* x, int stride,
                                          const QuadLane& q, f32x4 (&out)[2], unsigned long long* dg = nullptr) {
    const float* const rows[2] = {x, x + 4 * stride};
    wide_unit_sets_b<K, I, KVL, 2>(st, bias4, rows, stride, q, out, dg);
}

/* contract "dot_bf16" of a layer read from the ROUNDED store's W^T rows (root only: the representation net),
 * output group `grp`; K is a runtime fan-in <= 128 whose input row (padded with zeros to a multiple of 64)
 * sits at xrow */
__device__ __forceinline__ f32x4 wide_dense_store_b(const float* __restrict__ w, const Layer ly, int K, int grp,
                                                    const float* xrow, const QuadLane& q) {
    const int npad = ly.n_pad;
    const int n = quad_neuron_of_lane(q.lane);
    const float4 bv = *reinterpret_cast<const float4*>(w + ly.b_off + 64 * grp + q.n0);
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    f32x4 e = f32x4{bv.x, bv.y, bv.z, bv.w}, o = zero;
    const float* __restrict__ wt = w + ly.w_off + 64 * grp + n;
    for (int k0 = 0; k0 < K; k0 += 64) {
        const WideXB x = wide_load_xb(xrow + k0, q.grp);
        const int ng = (min(K - k0, 64) + 3) / 4;
        for (int gi = 0; gi < ng; ++gi) {
            float v[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] = k0 + 4 * gi + t < K ? wt[(size_t)(k0 + 4 * gi + t) * npad] : 0.0f;
            const s16x4 wg = as_s16x4(pack_hi(v[0], v[1]), pack_hi(v[2], v[3]));
            f32x4 d;
            switch (gi >> 2) {
                case 0: d = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wg, x.v[gi & 3], zero, 0, 0, 4); break;
                case 1: d = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wg, x.v[gi & 3], zero, 0, 0, 5); break;
                case 2: d = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wg, x.v[gi & 3], zero, 0, 0, 6); break;
                default: d = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wg, x.v[gi & 3], zero, 0, 0, 7); break;
            }
            if (gi & 1)
                o = (k0 == 0 && gi == 1) ? d : o + d;
            else
                e = e + d;
        }
    }
    return e + o;
}

/* Representation_function of the wide MLP (wide_representation under "dot_bf16") */
template <int SD, int H>
__device__ __forceinline__ void wide_representation_b(const Model& m, const WideBufs& B, const QuadLane& q,
                                                      float* hdst) {
    constexpr int HP = (H + 63) / 64 * 64, NG = HP / 64;
    float* xrow = B.xin + q.c * kWideXStride;
    float* trow = B.trunk + q.c * kWideTStride;
    f32x4 acc[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[g] = wide_dense_store_b(m.w, m.r_in, m.obs_dim, g, xrow, q);
    wave_sync();
#pragma unroll
    for (int g = 0; g < NG; ++g) store4(trow + 64 * g + q.n0, elu4<false>(acc[g]));
    wave_sync();
    for (int l = 0; l < m.L; ++l) {
#pragma unroll
        for (int g = 0; g < NG; ++g) acc[g] = wide_dense_store_b(m.w, m.r_mid, H, g, trow, q);
        wave_sync();
#pragma unroll
        for (int g = 0; g < NG; ++g) store4(trow + 64 * g + q.n0, elu4<false>(acc[g]));
        wave_sync();
    }
    const f32x4 a = wide_dense_store_b(m.w, m.r_out, H, 0, trow, q);
    const f32x4 y = wide_scale_to_bound(a, wide_valid(q, SD));
    wave_sync();
    store4(xrow + q.n0, y);
    store4(hdst + q.n0, y);
    wave_sync();
}

/* Representation_function of the 'lstm_model' (lstm_representation_quad under "dot_bf16"): one Linear */
template <int SD>
__device__ __forceinline__ void lstm_representation_quad_b(const Model& m, const WideBufs& Bf, const QuadLane& q,
                                                           float* hdst, int hid_stride) {
    float* xrow = Bf.xin + q.c * kWideXStride;
    f32x4 y = wide_dense_store_b(m.w, m.r_in, m.obs_dim, 0, xrow, q);
    const WideValid vd = wide_valid(q, SD);
#pragma unroll
    for (int r = 0; r < 4; ++r) y[r] = vd.v[r] ? y[r] : 0.0f;
    wave_sync();
    store4(xrow + q.n0, y);
    if (q.n0 < hid_stride) store4(hdst + q.n0, y);
    wave_sync();
}

/* ==================================================================================================
 * contract "dot_bf16x3" (MZS_PRECISION_BF16X3) on the weight stream: a chunk is a hi image followed by a lo
 * image (2 x 8 KB = the fp32 chunk's 16 KB and its 64 registers, k_build_stream_x splits the fp32 store),
 * the layer's input row is split when it is loaded, three matrix instructions per group of four k
 * (hi hi -> the even / odd chains, hi lo -> s, lo hi -> t; mzs_quad_bf16.h quad_dense_x3_step), result
 * (e + o) + (s + t).  48 matrix instructions per chunk and set of four games: the fp32 stream issues 64.
 * ================================================================================================ */
struct WideStreamX {
    typedef const __attribute__((address_space(1))) u32x4_t* P4;
    P4 img;
    int n_img, next_img;
    unsigned w[kBufs][64]; /* [0..31] hi halves, [32..63] lo halves, each as WideStreamB::w */
    __device__ __forceinline__ void init(const float* images, int image0, int n_images, int lane) {
        img = (P4)images + (size_t)image0 * 2 * kImageU4 + quad_neuron_of_lane(lane);
        n_img = n_images;
        next_img = 0;
    }
    template <int I, int j>
    __device__ __forceinline__ void request() {
        const P4 base = img + (size_t)next_img * 2 * kImageU4;
        const u32x4_t vh = base[j * 64], vl = base[kImageU4 + j * 64];
        unsigned(&wk)[64] = w[I % kBufs];
        wk[4 * j] = vh[0], wk[4 * j + 1] = vh[1], wk[4 * j + 2] = vh[2], wk[4 * j + 3] = vh[3];
        wk[32 + 4 * j] = vl[0], wk[32 + 4 * j + 1] = vl[1], wk[32 + 4 * j + 2] = vl[2], wk[32 + 4 * j + 3] = vl[3];
    }
    __device__ __forceinline__ void advance() { next_img = next_img + 1 == n_img ? 0 : next_img + 1; }
    template <int I, int KV, int part>
    __device__ __forceinline__ void load_part() {
        if constexpr (16 * part < KV) request<I, 2 * part>();
        if constexpr (16 * part + 8 < KV) request<I, 2 * part + 1>();
        if constexpr (part == 3) advance();
    }
    template <int I, int KV = 64>
    __device__ __forceinline__ void load() {
        load_part<I, KV, 0>();
        load_part<I, KV, 1>();
        load_part<I, KV, 2>();
        load_part<I, KV, 3>();
    }
    template <int I, int KV0>
    __device__ __forceinline__ void start(int first) {
        static_assert(kAhead == 1, "one chunk ahead");
        next_img = first;
        load<I, KV0>();
    }
};

__device__ __forceinline__ SplitX wide_load_x3(const float* row, int grp) {
    SplitX x;
    quad_load_x3_f32(row, grp, x);
    return x;
}

/* the three products of one group of four k for NS sets of games, waiting for their adds */
template <int NS>
struct X3Pipe {
    f32x4 a[NS], b[NS], c[NS];
};
/* the adds of group G of the layer (contract "dot_bf16x3": a -> the even / odd chain, b -> s, c -> t; o is
 * assigned at group 1, s and t at group 0) */
template <int NS>
__device__ __forceinline__ void x3_retire(int G, const X3Pipe<NS>& p, f32x4 (&e)[NS], f32x4 (&o)[NS], f32x4 (&s)[NS],
                                          f32x4 (&t)[NS]) {
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        if (G & 1)
            o[i] = G == 1 ? p.a[i] : add4s(o[i], p.a[i]);
        else
            e[i] = add4s(e[i], p.a[i]);
        s[i] = G == 0 ? p.b[i] : add4s(s[i], p.b[i]);
        t[i] = G == 0 ? p.c[i] : add4s(t[i], p.c[i]);
    }
}

/* groups 4 part .. 4 part + 3 of the chunk at k0.  Software pipeline: a group's 3 NS matrix instructions
 * issue, then the 12 NS adds of the group BEFORE it (pipe: v_add_f32, add4s), four behind each matrix instruction: no add
 * waits on the matrix pipe.  The fence keeps one group's products live at a time (left to itself the scheduler
 * hoists a quarter's 12 NS instructions ahead of their adds: 48 NS result registers on top of the 128 of the
 * stream, 528 spilled in the pair kernel). */
template <int K, int k0, int part, int NS, bool kPipe = true>
__device__ __forceinline__ void wide_quarter_x(const unsigned (&cur)[64], const SplitX (&x)[NS], f32x4 (&e)[NS],
                                               f32x4 (&o)[NS], f32x4 (&s)[NS], f32x4 (&t)[NS], X3Pipe<NS>& pipe) {
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        const int gi = 4 * part + gq;
        if (k0 + 4 * gi < K) {
            const int G = k0 / 4 + gi;
            const s16x4 wh = as_s16x4(cur[2 * gi], cur[2 * gi + 1]);
            const s16x4 wl = as_s16x4(cur[32 + 2 * gi], cur[32 + 2 * gi + 1]);
            X3Pipe<NS> n;
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                n.a[i] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wh, x[i].h[gq], zero, 0, 0, 4 + part);
                n.b[i] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wh, x[i].l[gq], zero, 0, 0, 4 + part);
                n.c[i] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wl, x[i].h[gq], zero, 0, 0, 4 + part);
            }
            if constexpr (kPipe) {
                if (G > 0) x3_retire<NS>(G - 1, pipe, e, o, s, t);
                pipe = n;
#pragma unroll
                for (int i = 0; i < 3 * NS; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                }
            } else {
                x3_retire<NS>(G, n, e, o, s, t);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int K, int k0, int IN, int KV, int NS>
__device__ __forceinline__ void wide_block_x(WideStreamX& st, const unsigned (&cur)[64], const SplitX (&x)[NS],
                                             f32x4 (&e)[NS], f32x4 (&o)[NS], f32x4 (&s)[NS], f32x4 (&t)[NS],
                                             X3Pipe<NS>& pipe) {
    st.template load_part<IN, KV, 0>();
    __builtin_amdgcn_sched_barrier(0);
    wide_quarter_x<K, k0, 0, NS>(cur, x, e, o, s, t, pipe);
    __builtin_amdgcn_sched_barrier(0);
    st.template load_part<IN, KV, 1>();
    __builtin_amdgcn_sched_barrier(0);
    wide_quarter_x<K, k0, 1, NS>(cur, x, e, o, s, t, pipe);
    __builtin_amdgcn_sched_barrier(0);
    st.template load_part<IN, KV, 2>();
    __builtin_amdgcn_sched_barrier(0);
    wide_quarter_x<K, k0, 2, NS>(cur, x, e, o, s, t, pipe);
    __builtin_amdgcn_sched_barrier(0);
    st.template load_part<IN, KV, 3>();
    __builtin_amdgcn_sched_barrier(0);
    wide_quarter_x<K, k0, 3, NS>(cur, x, e, o, s, t, pipe);
    __builtin_amdgcn_sched_barrier(0);
}

template <int K, int I, int KVL, int NS>
__device__ __forceinline__ void wide_unit_sets_x(WideStreamX& st, const float* __restrict__ bias4,
                                                 const float* const (&rows)[NS], int stride, const QuadLane& q,
                                                 f32x4 (&out)[NS], unsigned long long* dg = nullptr) {
    static_assert(K <= 128, "two chunks at most");
    const float4 bv = *reinterpret_cast<const float4*>(bias4);
    f32x4 e[NS], o[NS], s[NS], t[NS];
    SplitX x[NS];
    X3Pipe<NS> pipe;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        e[i] = f32x4{bv.x, bv.y, bv.z, bv.w};
        o[i] = s[i] = t[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        x[i] = wide_load_x3(rows[i] + q.c * stride, q.grp);
    }
    __builtin_amdgcn_sched_barrier(0);
    X3_WAIT_T(0);
    wide_block_x<K, 0, I + 1, (K > 64 ? 64 : KVL), NS>(st, st.w[I % kBufs], x, e, o, s, t, pipe);
    X3_MFMA_T(0);
    if constexpr (K > 64) {
        /* the second chunk's input halves replace the first's: 16 registers per set instead of 32 */
#pragma unroll
        for (int i = 0; i < NS; ++i) x[i] = wide_load_x3(rows[i] + q.c * stride + 64, q.grp);
        __builtin_amdgcn_sched_barrier(0);
        X3_WAIT_T(1);
        wide_block_x<K, 64, I + 2, KVL, NS>(st, st.w[(I + 1) % kBufs], x, e, o, s, t, pipe);
        X3_MFMA_T(1);
    }
    x3_retire<NS>((K + 3) / 4 - 1, pipe, e, o, s, t); /* the last group's adds */
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        out[i] = (e[i] + o[i]) + (s[i] + t[i]);
        /* the result is pinned HERE: where the model code consumes it under a branch (the pair kernel's two waves
         * finish a layer differently), the optimiser otherwise sinks the whole add chains into that branch, behind
         * the unit, and every matrix result waits for them in a register (768 of them: the accumulator file and
         * scratch) */
        asm volatile("" : "+v"(out[i]));
    }
}

template <int K, int I, int KVL = 64>
__device__ __forceinline__ f32x4 wide_unit(WideStreamX& st, const float* __restrict__ bias4, const float* xrow,
                                           const QuadLane& q) {
    const float* const rows[1] = {xrow};
    f32x4 out[1];
    wide_unit_sets_x<K, I, KVL, 1>(st, bias4, rows, 0, q, out);
    return out[0];
}

template <int K, int I, int KVL = 64>
__device__ __forceinline__ void pair_unit(WideStreamX& st, const float* __restrict__ bias4, const float* x, int stride,
                                          const QuadLane& q, f32x4 (&out)[2], unsigned long long* dg = nullptr) {
    const float* const rows[2] = {x, x + 4 * stride};
    wide_unit_sets_x<K, I, KVL, 2>(st, bias4, rows, stride, q, out, dg);
}

/* contract "dot_bf16x3" of a layer read from the fp32 store's W^T rows, split on the way (root only) */
__device__ __forceinline__ f32x4 wide_dense_store_x(const float* __restrict__ w, const Layer ly, int K, int grp,
                                                    const float* xrow, const QuadLane& q) {
    const int npad = ly.n_pad;
    const int n = quad_neuron_of_lane(q.lane);
    const float4 bv = *reinterpret_cast<const float4*>(w + ly.b_off + 64 * grp + q.n0);
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    f32x4 e = f32x4{bv.x, bv.y, bv.z, bv.w}, o = zero, s = zero, t = zero;
    const float* __restrict__ wt = w + ly.w_off + 64 * grp + n;
    for (int k0 = 0; k0 < K; k0 += 64) {
        const SplitX x = wide_load_x3(xrow + k0, q.grp);
        const int ng = (min(K - k0, 64) + 3) / 4;
        for (int gi = 0; gi < ng; ++gi) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = k0 + 4 * gi + u < K ? wt[(size_t)(k0 + 4 * gi + u) * npad] : 0.0f;
            unsigned h0, l0, h1, l1;
            split_pack(v[0], v[1], h0, l0);
            split_pack(v[2], v[3], h1, l1);
            const s16x4 wh = as_s16x4(h0, h1), wl = as_s16x4(l0, l1);
            const s16x4 xh = x.h[gi & 3], xl = x.l[gi & 3];
            f32x4 a, b, c;
            switch (gi >> 2) {
                case 0:
                    a = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wh, xh, zero, 0, 0, 4);
                    b = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wh, xl, zero, 0, 0, 4);
                    c = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wl, xh, zero, 0, 0, 4);
                    break;
                case 1:
                    a = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wh, xh, zero, 0, 0, 5);
                    b = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wh, xl, zero, 0, 0, 5);
                    c = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wl, xh, zero, 0, 0, 5);
                    break;
                case 2:
                    a = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wh, xh, zero, 0, 0, 6);
                    b = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wh, xl, zero, 0, 0, 6);
                    c = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wl, xh, zero, 0, 0, 6);
                    break;
                default:
                    a = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wh, xh, zero, 0, 0, 7);
                    b = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wh, xl, zero, 0, 0, 7);
                    c = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wl, xh, zero, 0, 0, 7);
                    break;
            }
            const bool first = k0 == 0 && gi == 0;
            if (gi & 1)
                o = (k0 == 0 && gi == 1) ? a : o + a;
            else
                e = e + a;
            s = first ? b : s + b;
            t = first ? c : t + c;
        }
    }
    return (e + o) + (s + t);
}

template <int SD, int H>
__device__ __forceinline__ void wide_representation_x(const Model& m, const WideBufs& B, const QuadLane& q,
                                                      float* hdst) {
    constexpr int HP = (H + 63) / 64 * 64, NG = HP / 64;
    float* xrow = B.xin + q.c * kWideXStride;
    float* trow = B.trunk + q.c * kWideTStride;
    f32x4 acc[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[g] = wide_dense_store_x(m.w, m.r_in, m.obs_dim, g, xrow, q);
    wave_sync();
#pragma unroll
    for (int g = 0; g < NG; ++g) store4(trow + 64 * g + q.n0, elu4<false>(acc[g]));
    wave_sync();
    for (int l = 0; l < m.L; ++l) {
#pragma unroll
        for (int g = 0; g < NG; ++g) acc[g] = wide_dense_store_x(m.w, m.r_mid, H, g, trow, q);
        wave_sync();
#pragma unroll
        for (int g = 0; g < NG; ++g) store4(trow + 64 * g + q.n0, elu4<false>(acc[g]));
        wave_sync();
    }
    const f32x4 a = wide_dense_store_x(m.w, m.r_out, H, 0, trow, q);
    const f32x4 y = wide_scale_to_bound(a, wide_valid(q, SD));
    wave_sync();
    store4(xrow + q.n0, y);
    store4(hdst + q.n0, y);
    wave_sync();
}

template <int SD>
__device__ __forceinline__ void lstm_representation_quad_x(const Model& m, const WideBufs& Bf, const QuadLane& q,
                                                           float* hdst, int hid_stride) {
    float* xrow = Bf.xin + q.c * kWideXStride;
    f32x4 y = wide_dense_store_x(m.w, m.r_in, m.obs_dim, 0, xrow, q);
    const WideValid vd = wide_valid(q, SD);
#pragma unroll
    for (int r = 0; r < 4; ++r) y[r] = vd.v[r] ? y[r] : 0.0f;
    wave_sync();
    store4(xrow + q.n0, y);
    if (q.n0 < hid_stride) store4(hdst + q.n0, y);
    wave_sync();
}

}  // namespace mzs
#endif
