/*
 * mzs_wide_kernels.h — the search kernels of the models whose weights ride the weight stream
 * (k_search_quadw2: wide MLP, mzs_quadw2.h; k_search_quadl: 'lstm_model', mzs_quadl.h) and the kernels that
 * build the stream's chunk images.  Templates, instantiated in mzsearch.hip (fp32: contract "dot") and in
 * mzsearch_bf16.hip (kB16 1: contract "dot_bf16" on the bf16 stream of mzs_quadw_bf16.h; 2: "dot_bf16x3" on its
 * hi + lo stream).
 */
#ifndef MZS_WIDE_KERNELS_H
#define MZS_WIDE_KERNELS_H

#include "mzs_search_kernels.h"
#include "mzs_quadw.h"
#include "mzs_quadw2.h"
#include "mzs_quadl.h"
#include "mzs_quadw_bf16.h"

namespace mzs {

/* block-cooperative copy of n floats of the weight store into LDS (before the block's barrier) */
__device__ __forceinline__ const float* tab_copy(float*& cursor, const float* src, int n) {
    float* dst = cursor;
    for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
    cursor += n;
    return dst;
}

/* ------------------------------------------------------------------------------
 * k_search_quadw2: the wide-MLP search with the weight stream shared by pairs of waves
 * (mzs_quadw2.h): each wave searches its own four games and computes one 64-neuron output group
 * of every layer for the pair's eight.  One block = four waves = two pairs; every wave reaches
 * every block barrier.
 * ---------------------------------------------------------------------------- */
template <int SD, int NA, int H, int kB16 = 0>
__global__ void __launch_bounds__(256, 1) k_search_quadw2(const SearchParams p, const QuadLayout ql) {
    extern __shared__ __align__(16) unsigned char smem[];
    const QuadLane q = quad_lane();
    const int wave = threadIdx.x >> 6, pair = wave >> 1, u = wave & 1;
    const Model& m = p.m;
    const int S = p.S;
    const int games_per_block = 4 * (blockDim.x >> 6); /* 16, or 8 when the trees of sixteen games do not fit */
    if (redo_skip_block(p, games_per_block)) return;

    quad_fill_tables(smem, ql, p);
    WideTabs T;
    {
        const float* w = m.w;
        float* cur = reinterpret_cast<float*>(smem + ql.off_tabs);
        T.d_in_b = tab_copy(cur, w + m.d_in.b_off, 128);
        T.d_mid_b = tab_copy(cur, w + m.d_mid.b_off, 128);
        T.d_heads_b = tab_copy(cur, w + m.d_heads.b_off, 128);
        T.p_in_b = tab_copy(cur, w + m.p_in.b_off, 128);
        T.p_mid_b = tab_copy(cur, w + m.p_mid.b_off, 128);
        T.p_heads_b = tab_copy(cur, w + m.p_heads.b_off, 128);
        T.d_onehot = tab_copy(cur, w + m.d_in.w_off + SD * 128, NA * 128);
    }
    __syncthreads();

    /* games past n_games replay the last one and write nothing (whole waves included) */
    const int gp0 = blockIdx.x * games_per_block + pair * 8, g0 = gp0 + 4 * u;
    const bool live = g0 + q.c < p.n_games;
    const int g = live ? g0 + q.c : p.n_games - 1;
    const QuadLds L = quad_carve(smem, ql, wave, q, false);
    WideBufs B;
    B.trunk = L.trunk;
    B.xin = L.trunk + 4 * kWideTStride;
    const PairBufs P = pair_carve(reinterpret_cast<float*>(smem + ql.shared_bytes) - (2 - pair) * kPairFloats);
    typename std::conditional<kB16 == 2, WideStreamX, typename std::conditional<kB16 == 1, WideStreamB, WideStream>::type>::type st;
    const int n_half = p.n_img / 2;
    st.init(p.img, u * n_half, n_half, q.lane);

    const double* __restrict__ noise = p.noise ? p.noise + (size_t)g * p.noise_stride : nullptr;
    const int noise_cap = p.noise ? (int)p.noise_avail : 0;
    const size_t game_hid = (size_t)(p.t.max_sims + 1) * p.t.hid_stride;
    float* hid_g = p.t.hidden + (size_t)g * game_hid;
    /* the global hidden rows of the pair's game rows 4 s + c (what wave 1 stores new states into) */
    float* hid_s[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        const int gs = gp0 + 4 * s2 + q.c;
        hid_s[s2] = p.t.hidden + (size_t)(gs < p.n_games ? gs : p.n_games - 1) * game_hid;
    }
    float* xrow = B.xin + q.c * kWideXStride;
    float* prow = P.xin + (4 * u + q.c) * kPairXStride;

    QuadState s = quad_state_init();
    quad_tree_init(L, q, p.n_nodes);
    double nz = S > 0 ? quad_noise_prefetch<NA>(s, q, noise, noise_cap) : 0.0;

    /* root: representation (each wave for its own games, from the store's rows) -> prediction (shared) */
    for (int k = q.b; k < kWideX; k += 16) xrow[k] = k < m.obs_dim ? p.obs[(size_t)g * m.obs_dim + k] : 0.0f;
    wave_sync();
    if constexpr (kB16 == 2)
        wide_representation_x<SD, H>(m, B, q, hid_g);
    else if constexpr (kB16 == 1)
        wide_representation_b<SD, H>(m, B, q, hid_g);
    else
        wide_representation<SD, H>(m, B, q, hid_g);
    *reinterpret_cast<float4*>(prow + 4 * q.b) = *reinterpret_cast<const float4*>(xrow + 4 * q.b);
    __syncthreads();
    st.template start<1, SD>(p.first_img);
    {
        float* const none[2] = {nullptr, nullptr};
#ifdef MZS_DIAG
        unsigned long long d0[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        pair_model<SD, H, NA, true>(st, m, T, P, u, q, none, d0);
#else
        pair_model<SD, H, NA, true>(st, m, T, P, u, q, none);
#endif
    }
    f32x4 pr;
    {
        const float4 v = *reinterpret_cast<const float4*>(P.pri + 4 * (4 * u + q.c));
        pr = f32x4{v.x, v.y, v.z, v.w};
    }
    quad_root_expand<NA>(L, s, q, p, g, pr);

#ifdef MZS_DIAG
    unsigned long long dsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    for (int sim = 0; sim < S; ++sim) {
        DIAG_T(t0);
        L.nwin[q.lane] = nz;
        wave_sync();
        const QuadSelect<NA> sel = quad_select<NA>(L, s, q, noise, noise_cap, sim);
        nz = quad_noise_prefetch<NA>(s, q, noise, noise_cap);
        const int e_new = sim + 1, e_parent = (sel.pbase - 1) / NA;
        const float4 hv = *reinterpret_cast<const float4*>(hid_g + (size_t)e_parent * p.t.hid_stride + 4 * q.b);
        *reinterpret_cast<float4*>(prow + 4 * q.b) = hv;
        if (q.b == 0) P.act[4 * u + q.c] = sel.action;
        DIAG_T(t1);
        __syncthreads();
        DIAG_T(t2);
        float* const hdst[2] = {hid_s[0] + (size_t)e_new * p.t.hid_stride, hid_s[1] + (size_t)e_new * p.t.hid_stride};
#ifdef MZS_DIAG
        pair_model<SD, H, NA, false>(st, m, T, P, u, q, hdst, dsum);
#else
        pair_model<SD, H, NA, false>(st, m, T, P, u, q, hdst);
#endif
        DIAG_T(t3);
        const float yr = P.rew[4 * u + q.c], yv = P.val[4 * u + q.c];
        {
            const float4 v = *reinterpret_cast<const float4*>(P.pri + 4 * (4 * u + q.c));
            pr = f32x4{v.x, v.y, v.z, v.w};
        }
        float reward, value;
        halves_pair(quad_inverse_value_transform(q.half ? yv : yr), reward, value);
        quad_expand<NA>(L, q, sel, sim, pr, reward);
        quad_backup(L, s, q, sel.depth, value, p.g32);
        DIAG_T(t4);
        DIAG_ADD(1, t0, t1);
        DIAG_ADD(2, t1, t2);
        DIAG_ADD(3, t2, t3);
        DIAG_ADD(4, t3, t4);
    }
#ifdef MZS_DIAG
    if (q.lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&g_diag[i], dsum[i]);
#endif
    if (live) quad_write_out<NA>(p, L, s, q, g);
}

/* ------------------------------------------------------------------------------
 * k_search_quadl: four games per wave for the 'lstm_model' (mzs_quadl.h): the structure of
 * k_search_quadw with the LSTM heads.
 * ---------------------------------------------------------------------------- */
template <int SD, int NA, int H, int kB16 = 0>
__global__ void __launch_bounds__(256, 1) k_search_quadl(const SearchParams p, const QuadLayout ql) {
    extern __shared__ __align__(16) unsigned char smem[];
    const QuadLane q = quad_lane();
    const int wave = threadIdx.x >> 6;
    const Model& m = p.m;
    const int S = p.S, hs = p.t.hid_stride;
    if (redo_skip_block(p, 4 * (blockDim.x >> 6))) return;

    quad_fill_tables(smem, ql, p);
    LstmTabs T;
    {
        const float* w = m.w;
        float* cur = reinterpret_cast<float*>(smem + ql.off_tabs);
        const auto head = [&](const LstmHead& hd, bool acts) {
            LstmHeadTabs t;
            t.lin_b = tab_copy(cur, w + hd.lin.b_off, 64);
            t.gate_b = tab_copy(cur, w + hd.gate[0].b_off, 192);
            t.bhh = tab_copy(cur, w + hd.bhh_off[0], 192);
            t.extra = acts ? tab_copy(cur, w + hd.lin.w_off + SD * 64, NA * 64) : nullptr;
            return t;
        };
        T.rew = head(m.l_rew, true), T.st = head(m.l_st, true), T.pol = head(m.l_pol, false), T.val = head(m.l_val, false);
    }
    __syncthreads();

    const int g0 = (blockIdx.x * (blockDim.x >> 6) + wave) * 4;
    if (g0 >= p.n_games || redo_skip_quad(p, g0)) return;
    const bool live = g0 + q.c < p.n_games;
    const int g = live ? g0 + q.c : p.n_games - 1;
    const QuadLds L = quad_carve(smem, ql, wave, q, false);
    WideBufs B;
    B.trunk = L.trunk;
    B.xin = L.trunk + 4 * kWideTStride;
    typename std::conditional<kB16 == 2, WideStreamX, typename std::conditional<kB16 == 1, WideStreamB, WideStream>::type>::type st;
    st.init(p.img, 0, p.n_img, q.lane);

    const double* __restrict__ noise = p.noise ? p.noise + (size_t)g * p.noise_stride : nullptr;
    const int noise_cap = p.noise ? (int)p.noise_avail : 0;
    float* hid_g = p.t.hidden + (size_t)g * (p.t.max_sims + 1) * hs;
    float* xrow = B.xin + q.c * kWideXStride;

    QuadState s = quad_state_init();
    quad_tree_init(L, q, p.n_nodes);
    double nz = S > 0 ? quad_noise_prefetch<NA>(s, q, noise, noise_cap) : 0.0;

    f32x4 pr;
    for (int k = q.b; k < kWideX; k += 16) xrow[k] = k < m.obs_dim ? p.obs[(size_t)g * m.obs_dim + k] : 0.0f;
    wave_sync();
    if constexpr (kB16 == 2)
        lstm_representation_quad_x<SD>(m, B, q, hid_g, hs);
    else if constexpr (kB16 == 1)
        lstm_representation_quad_b<SD>(m, B, q, hid_g, hs);
    else
        lstm_representation_quad<SD>(m, B, q, hid_g, hs);
    st.template start<0, SD>(p.first_img); /* the policy head's Linear: every head starts in phase 0 */
    (void)lstm_prediction_quad<SD, H>(st, T, B, q, NA, &pr);
    quad_root_expand<NA>(L, s, q, p, g, pr);

    for (int sim = 0; sim < S; ++sim) {
        L.nwin[q.lane] = nz;
        wave_sync();
        const QuadSelect<NA> sel = quad_select<NA>(L, s, q, noise, noise_cap, sim);
        nz = quad_noise_prefetch<NA>(s, q, noise, noise_cap);
        const int e_new = sim + 1, e_parent = (sel.pbase - 1) / NA;
        float4 hv = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (4 * q.b < hs) hv = *reinterpret_cast<const float4*>(hid_g + (size_t)e_parent * hs + 4 * q.b);
        *reinterpret_cast<float4*>(xrow + 4 * q.b) = hv;
        wave_sync();
        const float yr = lstm_dynamics_quad<SD, H>(st, T, B, q, sel.action, hid_g + (size_t)e_new * hs, hs);
        const float yv = lstm_prediction_quad<SD, H>(st, T, B, q, NA, &pr);
        float reward, value;
        halves_pair(quad_inverse_value_transform(q.half ? yv : yr), reward, value);
        quad_expand<NA>(L, q, sel, sim, pr, reward);
        quad_backup(L, s, q, sel.depth, value, p.g32);
    }
    if (live) quad_write_out<NA>(p, L, s, q, g);
}

/* ------------------------------------------------------------------------------
 * k_build_stream: the chunk images k_search_quadw / k_search_quadl consume (mzs_quadw.h
 * ImageStream), from the device weight store: image u = rows k0 .. k0+63 of one 64-neuron output
 * group in the layout [k / 4][neuron][4]; rows past the fan-in are zero (and never multiplied).
 * One block per image, before every such search: the store changes under training, 0.3 MB of
 * images cost less than tracking that.
 * ---------------------------------------------------------------------------- */
struct StreamUnit {
    int w_off, npad, K, k0; /* W^T (+ the group's column) in the store, its row stride, fan-in, first row */
};
struct StreamPlan {
    int n, first; /* images per simulation; the one the root's prediction starts at */
    StreamUnit u[kMaxImages];
};

static __global__ void __launch_bounds__(256) k_build_stream(const float* __restrict__ w, const StreamPlan plan,
                                                             float* __restrict__ img) {
    const StreamUnit u = plan.u[blockIdx.x];
    float* out = img + (size_t)blockIdx.x * kImageFloats;
    for (int i = threadIdx.x; i < 64 * 64; i += blockDim.x) {
        const int k = i >> 6, n = i & 63;
        const float v = u.k0 + k < u.K ? w[u.w_off + (size_t)(u.k0 + k) * u.npad + n] : 0.0f;
        out[((k >> 2) * 64 + n) * 4 + (k & 3)] = v;
    }
}

/* the same images for the bf16 stream (mzs_quadw_bf16.h): 8 KB each, layout [k / 8][neuron][8] halves, from
 * the ROUNDED store (every weight already a bfloat16 value held in a float: its high half) */
static __global__ void __launch_bounds__(256) k_build_stream_b(const float* __restrict__ w, const StreamPlan plan,
                                                               float* __restrict__ img) {
    const StreamUnit u = plan.u[blockIdx.x];
    unsigned short* out = reinterpret_cast<unsigned short*>(img) + (size_t)blockIdx.x * 64 * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += blockDim.x) {
        const int k = i >> 6, n = i & 63;
        const float v = u.k0 + k < u.K ? w[u.w_off + (size_t)(u.k0 + k) * u.npad + n] : 0.0f;
        out[((k >> 3) * 64 + n) * 8 + (k & 7)] = (unsigned short)((unsigned)mzs_f2bits(v) >> 16);
    }
}

/* ... and for the hi + lo stream (WideStreamX): per chunk a hi image then a lo image in the bf16 layout, split
 * from the fp32 store (mzs_bf16_split: hi = round(v), lo = round(v - hi)) */
static __global__ void __launch_bounds__(256) k_build_stream_x(const float* __restrict__ w, const StreamPlan plan,
                                                               float* __restrict__ img) {
    const StreamUnit u = plan.u[blockIdx.x];
    unsigned short* out = reinterpret_cast<unsigned short*>(img) + (size_t)blockIdx.x * 2 * 64 * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += blockDim.x) {
        const int k = i >> 6, n = i & 63;
        const float v = u.k0 + k < u.K ? w[u.w_off + (size_t)(u.k0 + k) * u.npad + n] : 0.0f;
        unsigned hi, lo;
        split_pack(v, 0.0f, hi, lo);
        out[((k >> 3) * 64 + n) * 8 + (k & 7)] = (unsigned short)(hi & 0xffffu); /* split_pack: first value in the low half */
        out[64 * 64 + ((k >> 3) * 64 + n) * 8 + (k & 7)] = (unsigned short)(lo & 0xffffu);
    }
}

}  // namespace mzs
#endif
