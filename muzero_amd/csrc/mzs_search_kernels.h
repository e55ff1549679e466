/*
 * mzs_search_kernels.h — what the translation units of libmzsearch.so share about the fused
 * search kernels: the parameter blocks, the per-game tree helpers of the four-games-per-wave
 * design and the k_search_quad kernel template itself.
 *
 * Two translation units instantiate the template: mzsearch.hip (fp32: MFMA accumulators chain in
 * AGPRs, the weights fill the VGPRs) and mzsearch_bf16.hip (contract "dot_bf16": every MFMA starts
 * from a zero accumulator and its result goes straight into fp32 adds, so that unit is compiled
 * with -mllvm -amdgpu-mfma-vgpr-form — with AGPR destinations each MFMA costs four
 * v_accvgpr_read_b32 and a serialising s_nop, measured 8 % SLOWER than the fp32 kernel).
 *
 * Reference code restated (file:line in /root/reference): monte_carlo_tree_search.py:88-123
 * (root), :134-145 (noise), :186-200 (expand), :215-222 (results), :224-265 (run).
 */
#ifndef MZS_SEARCH_KERNELS_H
#define MZS_SEARCH_KERNELS_H

#include <type_traits>

#include "../../include/mzsearch.h"
#include "mzs_device.h"
#include "mzs_quad.h"
#include "mzs_quad_bf16.h"

namespace mzs {

/* byte offsets inside one wave's LDS slice */
struct LdsLayout {
    int off_cb, off_path, off_xh, off_xa, off_xb, off_ob, off_vc, bytes;
    int shared_bytes; /* per workgroup, behind the waves' slices (vision search: the parameter cache) */
};

struct TreeArrays {
    int* visit;
    float* vsum;
    float* prior;
    float* reward;
    int* child_base;
    int* sim_leaf;
    float* minmax;
    float* hidden;
    int* status;
    int stride;      /* elements per game in the node arrays */
    int hid_stride;  /* floats per hidden state              */
    int max_sims;
};

struct Results {
    int* visits;
    float* root_value;
    int* visits_sink;       /* mzs_set_result_sink: a second destination (pinned host memory) or null */
    float* root_value_sink;
    double* root_prior;
    long long* consumed;
    long long* path_len;
};

struct SearchParams {
    Model m;
    LdsLayout lds;
    TreeArrays t;
    Results r;
    int n_games, S, train, n_nodes;
    float g32, omf32;
    double frac;
    const double* pbc;
    const float* obs;
    const double* dirichlet;
    const double* noise;
    long long noise_stride;
    long long noise_avail; /* draws of every game's row that hold data (<= noise_stride): a search that asks
                              for more sets status[g] and *overflow (mzs_search_streams redoes it) */
    int* overflow;         /* one flag per handle, or null */
    int redo;              /* 1: search only the games whose status is set (after a prefix search) */
    const float* t_root_policy;
    const float* t_reward;
    const float* t_value;
    const float* t_prior;
    const float* img; /* chunk images of one simulation in consumption order (k_build_stream) */
    int n_img, first_img;
};


typedef void (*quad_kernel_t)(const SearchParams, const QuadLayout);

/* redo launches (SearchParams.redo): true if nothing in this block's games_per_block games asked for
 * more draws.  Block-uniform, contains a block barrier: call it before any other barrier or return. */
__device__ __forceinline__ bool redo_skip_block(const SearchParams& p, int games_per_block) {
    if (!p.redo) return false;
    if (p.overflow && *p.overflow == 0) return true;
    const int gb = blockIdx.x * games_per_block;
    int any = 0;
    for (int i = threadIdx.x; i < games_per_block; i += blockDim.x)
        if (gb + i < p.n_games && p.t.status[gb + i]) any = 1;
    return !__syncthreads_or(any);
}
/* the four games of a quad wave: redo them if any of them asked for more (the others replay to the
 * same results: they never read past the prefix) */
__device__ __forceinline__ bool redo_skip_quad(const SearchParams& p, int g0) {
    if (!p.redo) return false;
    int any = 0;
    for (int c = 0; c < 4; ++c)
        if (g0 + c < p.n_games && p.t.status[g0 + c]) any = 1;
    return !any;
}

/* bf16 instantiations of k_search_quad (mzsearch_bf16.hip); nullptr: the shape has none */
quad_kernel_t pick_quad_kernel_bf16(const mzs_dims& d, bool hid_global, bool two_per_simd = false);
/* "dot_bf16" instantiations of k_search_quadw2 / k_search_quadl (mzsearch_bf16.hip) for the shapes the fp32
 * pickers of mzsearch.hip accept */
quad_kernel_t pick_wide_kernel_bf16(const mzs_dims& d);
quad_kernel_t pick_wide_kernel_x3(const mzs_dims& d); /* mzsearch_x3w.hip */
/* mzsearch_f32v.hip: the fp32 kernels that gain from matrix results in VGPRs */
quad_kernel_t pick_quad_kernel_f32v(const mzs_dims& d, bool hid_global);
quad_kernel_t pick_wide_kernel_f32v(const mzs_dims& d);

/* ------------------------------------------------------------------------------
 * Quad fused engine (mzs_quad.h): four games per wavefront, dense layers on
 * v_mfma_f32_4x4x1_16b_f32 with the weights in registers, trees and hidden states in LDS,
 * compile-time S_dim / A / H.  One block = up to four waves = sixteen games.
 * ---------------------------------------------------------------------------- */
#ifdef MZS_DIAG
/* diagnostic build only: per-phase cycle sums (s_memtime), never read by product code */
static __device__ unsigned long long g_diag[8]; /* one per translation unit */
/* add this unit's counters to out8 and clear them */
static inline bool diag_drain(unsigned long long* out8) {
    unsigned long long v[8], z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyFromSymbol(v, HIP_SYMBOL(g_diag), sizeof v) != hipSuccess) return false;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_diag), z, sizeof z) != hipSuccess) return false;
    for (int i = 0; i < 8; ++i) out8[i] += v[i];
    return true;
}
bool diag_drain_bf16(unsigned long long* out8); /* mzsearch_bf16.hip */
bool diag_drain_x3w(unsigned long long* out8);  /* mzsearch_x3w.hip */
bool diag_drain_f32v(unsigned long long* out8); /* mzsearch_f32v.hip */
__device__ __forceinline__ unsigned long long diag_stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define DIAG_T(var) const unsigned long long var = diag_stamp()
#define DIAG_ADD(i, a, b) dsum[i] += (b) - (a)
#else
#define DIAG_T(var)
#define DIAG_ADD(i, a, b)
#endif

/* ---- pieces shared by the four-games-per-wave kernels (k_search_quad, k_search_quadw) ---- */

/* block-shared tables at the start of dynamic LDS: sqrt(n) pb_c(n), 1/n in f64 and f32
 * (correctly rounded reciprocals for div_r*; the selection reads the f64 ones only in the
 * register-starved kMid kernels, elsewhere it uses quad_rcp_count); the caller synchronises the
 * block afterwards */
__device__ __forceinline__ void quad_fill_tables(unsigned char* smem, const QuadLayout& ql, const SearchParams& p) {
    double* pbc_s = reinterpret_cast<double*>(smem);
    double* inv64_s = reinterpret_cast<double*>(smem + ql.off_inv64);
    float* inv32_s = reinterpret_cast<float*>(smem + ql.off_inv32);
    for (int i = threadIdx.x; i < p.S + 2; i += blockDim.x) {
        pbc_s[i] = p.pbc[i];
        inv64_s[i] = 1.0 / (double)(i > 0 ? i : 1);
        inv32_s[i] = 1.0f / (float)(i > 0 ? i : 1);
    }
}

/* this lane's view of the block's LDS: shared tables, its wave's buffers, its game's tree */
__device__ __forceinline__ QuadLds quad_carve(unsigned char* smem, const QuadLayout& ql, int wave, const QuadLane& q,
                                              bool with_hidden) {
    unsigned char* wbase = smem + ql.shared_bytes + (size_t)wave * ql.wave_bytes;
    unsigned char* gbase = wbase + ql.off_games + (size_t)q.c * ql.g_bytes;
    QuadLds L;
    L.trunk = reinterpret_cast<float*>(wbase);
    L.nwin = reinterpret_cast<double*>(wbase + ql.off_nwin);
    L.onehot = reinterpret_cast<const float*>(smem + ql.off_onehot);
    L.pbc = reinterpret_cast<const double*>(smem);
    L.inv64 = reinterpret_cast<const double*>(smem + ql.off_inv64);
    L.inv32 = reinterpret_cast<const float*>(smem + ql.off_inv32);
    L.nodes = reinterpret_cast<QNode*>(gbase);
    L.vsum = reinterpret_cast<float*>(gbase + ql.g_vsum);
    L.path = reinterpret_cast<uint2*>(gbase + ql.g_path);
    L.hid = with_hidden ? reinterpret_cast<float*>(gbase + ql.g_hid) : nullptr;
    L.leafs = reinterpret_cast<unsigned short*>(gbase + ql.g_leafs);
    return L;
}

__device__ __forceinline__ QuadState quad_state_init() {
    QuadState s;
    s.mn = __builtin_inff();
    s.mx = -__builtin_inff();
    s.rinv = 0.0f;
    s.rootp = 0.0;
    s.consumed = 0;
    s.path_total = 0;
    s.status = 0;
    return s;
}

/* empty tree: every node unvisited, the root (reward 0) on the path */
__device__ __forceinline__ void quad_tree_init(const QuadLds& L, const QuadLane& q, int n_nodes) {
    for (int n = q.b; n < n_nodes; n += 16) {
        QNode z;
        z.vc = 0u;
        z.x0 = 0.0f;
        z.prior = 0.0f;
        z.reward = 0.0f;
        L.nodes[n] = z;
        L.vsum[n] = 0.0f;
    }
    if (q.b == 0) L.path[0] = make_uint2(0u, 0u);
}

/* expand the root with the priors pr (mcts.py:113-123) and mix the exploration noise into them
 * (add_exploration_noise_at_the_root, mcts.py:134-145: f32 product widened, f64 sum) */
template <int NA>
__device__ __forceinline__ void quad_root_expand(const QuadLds& L, QuadState& s, const QuadLane& q,
                                                 const SearchParams& p, int g, const f32x4 pr) {
    if (q.b == 0) {
#pragma unroll
        for (int r = 0; r < NA; ++r) L.nodes[1 + r].prior = pr[r];
        L.nodes[0].vc = 1u;
    }
    wave_sync();
    constexpr int NAL = NA <= 2 ? 2 : 4;
    const int a = q.b & (NAL - 1);
    if (a < NA) {
        const float pa = L.nodes[1 + a].prior;
        if (p.train && p.S > 0)
            s.rootp = (double)(pa * p.omf32) + p.dirichlet[(size_t)g * NA + a] * p.frac;
        else
            s.rootp = (double)pa;
    }
}

/* expand the selected leaf (mcts.py:186-200) */
template <int NA>
__device__ __forceinline__ void quad_expand(const QuadLds& L, const QuadLane& q, const QuadSelect<NA>& sel, int sim,
                                            const f32x4 pr, float reward) {
    const int cbase = 1 + (sim + 1) * NA;
    if (q.b == 0) {
#pragma unroll
        for (int r = 0; r < NA; ++r) L.nodes[cbase + r].prior = pr[r];
        L.nodes[sel.leaf].reward = reward;
        L.nodes[sel.leaf].vc = (unsigned)cbase; /* never visited before its expansion */
        L.leafs[sim] = (unsigned short)sel.leaf;
        L.path[sel.depth].y = __float_as_uint(reward);
    }
    wave_sync();
}

/* tree -> HBM structure of arrays; results (mcts.py:215-222).  Hidden states are the caller's. */
template <int NA>
__device__ __forceinline__ void quad_write_out(const SearchParams& p, const QuadLds& L, const QuadState& s,
                                               const QuadLane& q, int g) {
    const size_t row = (size_t)g * p.t.stride;
    for (int n = q.b; n < p.n_nodes; n += 16) {
        const QNode nd = L.nodes[n];
        p.t.visit[row + n] = (int)(nd.vc >> 16);
        p.t.vsum[row + n] = L.vsum[n];
        p.t.prior[row + n] = nd.prior;
        p.t.reward[row + n] = nd.reward;
        p.t.child_base[row + n] = (int)(nd.vc & 0xffffu);
    }
    for (int i = q.b; i < p.S; i += 16) p.t.sim_leaf[(size_t)g * p.t.max_sims + i] = L.leafs[i];
    constexpr int NAL = NA <= 2 ? 2 : 4;
    if (q.b < NAL && q.b < NA) {
        p.r.visits[(size_t)g * NA + q.b] = (int)(L.nodes[1 + q.b].vc >> 16);
        if (p.r.visits_sink) p.r.visits_sink[(size_t)g * NA + q.b] = (int)(L.nodes[1 + q.b].vc >> 16);
        p.r.root_prior[(size_t)g * NA + q.b] = s.rootp;
    }
    if (q.b == 0) {
        const int rv = (int)(L.nodes[0].vc >> 16);
        const float rootv = rv == 0 ? 0.0f : L.vsum[0] / (float)rv;
        p.r.root_value[g] = rootv;
        if (p.r.root_value_sink) p.r.root_value_sink[g] = rootv;
        p.r.consumed[g] = s.consumed;
        p.r.path_len[g] = s.path_total;
        p.t.minmax[2 * (size_t)g] = s.mn;
        p.t.minmax[2 * (size_t)g + 1] = s.mx;
        p.t.status[g] = s.status;
        if (s.status && p.overflow) *p.overflow = 1;
    }
}

/* kB16: contract "dot_bf16" (mzs_quad_bf16.h): packed bf16 weights in half the registers, dense
 * layers on v_mfma_f32_4x4x4_16B_bf16; the tree code and every epilogue are the fp32 ones */
/* kOcc: wavefronts per SIMD the register budget is compiled for.  2 (bf16 + kHidG only: packed weights
 * take half the registers and the trees of sixteen games 50 KB of LDS) lets two blocks share a CU when
 * a batch has more than sixteen games per CU: each wave's dependent chains hide behind the other's. */
/* kB16 == 2: contract "dot_bf16x3" (hi + lo pairs, three MFMAs per group; MZS_PRECISION_BF16X3) */
template <int SD, int NA, int H, bool kInjected, bool kMid = false, bool kHidG = false, int kB16 = 0, int kOcc = 1>
__global__ void __launch_bounds__(256, kOcc) k_search_quad(const SearchParams p, const QuadLayout ql) {
    extern __shared__ __align__(16) unsigned char smem[];
    const QuadLane q = quad_lane();
    const int wave = threadIdx.x >> 6;
    const Model& m = p.m;
    const int S = p.S;

    if (redo_skip_block(p, 4 * (blockDim.x >> 6))) return;
    /* block-shared tables, register-resident weights, then the only other block-level barrier */
    quad_fill_tables(smem, ql, p);
    typename std::conditional<kB16 == 2, QuadWeightsX3<SD, NA, H, kMid>,
                              typename std::conditional<kB16 == 1, QuadWeightsB<SD, NA, H, kMid>,
                                                        QuadWeights<SD, NA, H, kMid>>::type>::type W;
    if (!kInjected) {
        float* onehot_s = reinterpret_cast<float*>(smem + ql.off_onehot);
        for (int i = threadIdx.x; i < NA * 64; i += blockDim.x)
            onehot_s[i] = m.w[m.d_in.w_off + (SD + (i >> 6)) * 64 + (i & 63)];
        if constexpr (kB16 == 2)
            quad_load_weights_x3<SD, NA, H, kMid>(m, W, q);
        else if constexpr (kB16 == 1)
            quad_load_weights_b<SD, NA, H, kMid>(m, W, q);
        else
            quad_load_weights<SD, NA, H, kMid>(m, W, q);
    }
    __syncthreads();

    const int g0 = (blockIdx.x * (blockDim.x >> 6) + wave) * 4;
    if (g0 >= p.n_games || redo_skip_quad(p, g0)) return;
    const bool live = g0 + q.c < p.n_games;          /* dead slots replay the last game, write nothing */
    const int g = live ? g0 + q.c : p.n_games - 1;
    /* kHidG: the hidden states of the expansions live in the global SoA array (L2) and LDS keeps
     * one 32-float row per game — for trees whose sixteen games would not fit the CU's LDS next
     * to 6.5 KB of hidden states each (three or four actions, or more than ~55 simulations) */
    QuadLds L = quad_carve(smem, ql, wave, q, !kHidG);
    float* hid_g = p.t.hidden + (size_t)g * (p.t.max_sims + 1) * p.t.hid_stride;
    if constexpr (kHidG) L.hid = L.trunk + 4 * kTrunkStride + q.c * kHidRowStride;

    const double* __restrict__ noise = p.noise ? p.noise + (size_t)g * p.noise_stride : nullptr;
    const int noise_cap = p.noise ? (int)p.noise_avail : 0;

    QuadState s = quad_state_init();
    quad_tree_init(L, q, p.n_nodes);
    double nz = S > 0 ? quad_noise_prefetch<NA>(s, q, noise, noise_cap) : 0.0;

    /* root: representation -> prediction -> expand (mcts.py:88-123) */
    f32x4 pr;
    if (kInjected) {
#pragma unroll
        for (int r = 0; r < 4; ++r) pr[r] = r < NA ? p.t_root_policy[(size_t)g * NA + r] : 0.0f;
    } else {
        if constexpr (kB16) { /* the bf16 input layer reads whole groups of four k: zero the row's tail */
            for (int k = q.b; k < 64; k += 16) L.trunk[q.c * kTrunkStride + k] = k < m.obs_dim ? p.obs[(size_t)g * m.obs_dim + k] : 0.0f;
        } else {
            for (int k = q.b; k < m.obs_dim; k += 16) L.trunk[q.c * kTrunkStride + k] = p.obs[(size_t)g * m.obs_dim + k];
        }
        wave_sync();
        if constexpr (kB16 == 2)
            quad_representation_x3<SD, NA, H>(m, L, q, L.hid);
        else if constexpr (kB16 == 1)
            quad_representation_b<SD, NA, H>(m, L, q, L.hid);
        else
            quad_representation<SD, NA, H>(m, L, q, L.hid);
        if constexpr (kHidG) {
            if (q.b < 8) *reinterpret_cast<float4*>(hid_g + 4 * q.b) = *reinterpret_cast<const float4*>(L.hid + 4 * q.b);
        }
        if constexpr (kB16 == 2)
            (void)quad_prediction_x3<SD, NA, H, kMid>(W, L, q, L.hid, &pr, m.L);
        else if constexpr (kB16 == 1)
            (void)quad_prediction_b<SD, NA, H, kMid>(W, L, q, L.hid, &pr, m.L);
        else
            (void)quad_prediction<SD, NA, H, kMid>(W, L, q, L.hid, &pr, m.L);   /* the root's value is not used */
    }
    quad_root_expand<NA>(L, s, q, p, g, pr);

#ifdef MZS_DIAG
    unsigned long long dsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    for (int sim = 0; sim < S; ++sim) {
        DIAG_T(t0);
        L.nwin[q.lane] = nz;
        wave_sync();
        const QuadSelect<NA> sel = quad_select<NA, kMid>(L, s, q, noise, noise_cap, sim);
        nz = quad_noise_prefetch<NA>(s, q, noise, noise_cap);
        DIAG_T(t1);
        const int e_new = sim + 1;
        float reward, value;
        if (kInjected) {
            reward = p.t_reward[(size_t)g * S + sim];
            value = p.t_value[(size_t)g * S + sim];
#pragma unroll
            for (int r = 0; r < 4; ++r) pr[r] = r < NA ? p.t_prior[((size_t)g * S + sim) * NA + r] : 0.0f;
        } else {
            /* dynamics on the parent's hidden state, prediction on the new one (mcts.py:180-192) */
            const int e_parent = (sel.pbase - 1) / NA;
            float* hn = kHidG ? L.hid : L.hid + e_new * 32;
            const float* hp = kHidG ? L.hid : L.hid + e_parent * 32;
            if constexpr (kHidG) { /* the parent's hidden state: 32 floats per game, four per lane */
                if (q.b < 8)
                    *reinterpret_cast<float4*>(L.hid + 4 * q.b) =
                        *reinterpret_cast<const float4*>(hid_g + (size_t)e_parent * p.t.hid_stride + 4 * q.b);
                wave_sync();
            }
            float yr;
            if constexpr (kB16 == 2)
                yr = quad_dynamics_x3<SD, NA, H, kMid>(W, L, q, hp, sel.action, hn, m.L);
            else if constexpr (kB16 == 1)
                yr = quad_dynamics_b<SD, NA, H, kMid>(W, L, q, hp, sel.action, hn, m.L);
            else
                yr = quad_dynamics<SD, NA, H, kMid>(W, L, q, hp, sel.action, hn, m.L);
            if constexpr (kHidG) {
                if (q.b < 8)
                    *reinterpret_cast<float4*>(hid_g + (size_t)e_new * p.t.hid_stride + 4 * q.b) =
                        *reinterpret_cast<const float4*>(L.hid + 4 * q.b);
            }
            DIAG_T(t2);
            DIAG_ADD(2, t1, t2);
            float yv;
            if constexpr (kB16 == 2)
                yv = quad_prediction_x3<SD, NA, H, kMid>(W, L, q, hn, &pr, m.L);
            else if constexpr (kB16 == 1)
                yv = quad_prediction_b<SD, NA, H, kMid>(W, L, q, hn, &pr, m.L);
            else
                yv = quad_prediction<SD, NA, H, kMid>(W, L, q, hn, &pr, m.L);
            /* one inverse transform for both scalars: reward in the low half, value in the high */
            halves_pair(quad_inverse_value_transform(q.half ? yv : yr), reward, value);
        }
        DIAG_T(t3);
        quad_expand<NA>(L, q, sel, sim, pr, reward);
        quad_backup(L, s, q, sel.depth, value, p.g32);
        DIAG_T(t4);
        DIAG_ADD(1, t0, t1);
        DIAG_ADD(3, t1, t3);
        DIAG_ADD(4, t3, t4);
    }

    if (live) {
        quad_write_out<NA>(p, L, s, q, g);
        if (!kInjected && !kHidG) { /* hidden states of the expansions */
            for (int i = q.b; i < (S + 1) * 32; i += 16) {
                const int e = i >> 5, k = i & 31;
                if (k < SD) hid_g[(size_t)e * p.t.hid_stride + k] = L.hid[i];
            }
        }
    }
#ifdef MZS_DIAG
    if (q.lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&g_diag[i], dsum[i]);
#endif
}


}  // namespace mzs
#endif
