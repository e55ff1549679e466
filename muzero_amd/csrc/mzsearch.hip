/*
 * mzsearch.hip — kernels and C-ABI of libmzsearch.so (gfx950 only).
 * Entry points and the reference code each replaces: include/mzsearch.h.
 *
 * Build: muzero_amd/csrc/Makefile
 *   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
 *         -fhip-fp32-correctly-rounded-divide-sqrt -shared -fPIC
 */
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/mzsearch.h"
#include "../../include/mzs_crmath.h"
#include "mzs_device.h"
#include "mzs_quad.h"
#include "mzs_vision_mm.h"
#include "mzs_search_kernels.h"
#include "mzs_quadw.h"
#include "mzs_quadw2.h"
#include "mzs_quadl.h"
#include "mzs_wide_kernels.h"
#include "mzs_train.h"
#include "mzs_train_lstm.h"
#include "mzs_selfplay.h"
#include "mzs_replay.h"
#include "mzs_vision_train.h"

using namespace mzs;

/* Scratch of the no-allocation entry point (mzs_train_grad_static, include/mzsearch.h): one copy
 * per GPU the library runs on, held until the process ends.  In host code this name is only a
 * stand-in for the symbol: the device address comes from hipGetSymbolAddress, once per device
 * (scratch_for_device below), and only that pointer is ever handed to kernels. */
constexpr size_t kStaticScratchBytes = (size_t)MZS_STATIC_SCRATCH_BYTES;
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(MZS_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));     \
    } while (0)

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

/* device address of this GPU's copy of `scratch`; `device` must be the current device */
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mu;
void* g_scratch_ptr[kMaxDevices] = {};

int scratch_for_device(int device, void** out) {
    if (device < 0 || device >= kMaxDevices) return fail(MZS_E_ARG, "device ordinal out of range");
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[device]) {
        void* p = nullptr;
        HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) return fail(MZS_E_HIP, "hipGetSymbolAddress(scratch) returned a null address");
        g_scratch_ptr[device] = p;
    }
    *out = g_scratch_ptr[device];
    return MZS_OK;
}

__device__ __forceinline__ WaveLds carve(unsigned char* base, const LdsLayout& l) {
    WaveLds L;
    L.nodes = reinterpret_cast<Node*>(base);
    L.cb = reinterpret_cast<unsigned short*>(base + l.off_cb);
    L.path = reinterpret_cast<unsigned short*>(base + l.off_path);
    L.xh = reinterpret_cast<float*>(base + l.off_xh);
    L.xa = reinterpret_cast<float*>(base + l.off_xa);
    L.xb = reinterpret_cast<float*>(base + l.off_xb);
    L.ob = reinterpret_cast<float*>(base + l.off_ob);
    L.vc = reinterpret_cast<float*>(base + l.off_vc);
    return L;
}

/* lane pair (2i, 2i+1) of a global row of length n */
__device__ __forceinline__ Pair pair_from_global(const float* row, int n, int lane) {
    Pair p;
    const int i = 2 * lane;
    p.va = i < n;
    p.vb = i + 1 < n;
    p.a = p.va ? row[i] : 0.0f;
    p.b = p.vb ? row[i + 1] : 0.0f;
    return p;
}

__device__ __forceinline__ void pair_to_global(float* row, const Pair p, int lane) {
    const int i = 2 * lane;
    if (p.va) row[i] = p.a;
    if (p.vb) row[i + 1] = p.b;
}

/* ------------------------------------------------------------------------------
 * Fused engine: Monte_carlo_tree_search.run (monte_carlo_tree_search.py:224-265) for
 * one game per wavefront, all simulations in one launch, tree resident in LDS and
 * written to the HBM structure-of-arrays once at the end.
 * ---------------------------------------------------------------------------- */
/* kVisWaves (vision model): waves per SIMD the kernel is compiled for — 2: up to 256 registers, the Linear
 * layers request 64 rows per batch; 4: 128 registers, 32 rows per batch (search_launch picks by game count) */
template <bool kInjected, bool kVision = false, int kVisWaves = 4>
__global__ void __launch_bounds__(256, kVision ? kVisWaves : 4) k_search_fused(const SearchParams p) {
    constexpr int kVisBatch = kVisWaves >= 4 ? 32 : 64;
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    const int g = blockIdx.x * (blockDim.x >> 6) + wave;
    if (redo_skip_block(p, blockDim.x >> 6)) return;
    WaveLds L = carve(smem + (size_t)wave * p.lds.bytes, p.lds);
    const Model& m = p.m;
    if constexpr (kVision && !kInjected) { /* the block's copy of the convolution / batchnorm parameters */
        L.vc = reinterpret_cast<float*>(smem + (size_t)(blockDim.x >> 6) * p.lds.bytes);
        vis_cache_load_block(m.blob, vis_offsets(m), L.vc);
    }
    if (g >= p.n_games || (p.redo && p.t.status[g] == 0)) return; /* no block-level barrier anywhere below */
    const int A = m.A, S = p.S;

    SearchConst c;
    c.A = A;
    c.S = S;
    c.n_nodes = p.n_nodes;
    c.g32 = p.g32;
    c.pbc = p.pbc;
    c.noise = p.noise ? p.noise + (size_t)g * p.noise_stride : nullptr;
    c.noise_cap = p.noise ? p.noise_avail : 0;

    SearchState s;
    s.mn = __builtin_inff();
    s.mx = -__builtin_inff();
    s.consumed = 0;
    s.path_total = 0;
    s.status = 0;
    s.root_prior = 0.0;
    if (c.noise) {
        noise_refill(c, s, lane);
    } else {
        s.win_base = 0;
        s.win = 0.0;
    }

    for (int n = lane; n < p.n_nodes; n += kWave) {
        Node z;
        z.visit = 0;
        z.vsum = 0.0f;
        z.prior = 0.0f;
        z.reward = 0.0f;
        L.nodes[n] = z;
        L.cb[n] = 0;
    }
    wave_sync();

    float* hid_g = p.t.hidden + (size_t)g * (p.t.max_sims + 1) * p.t.hid_stride;

    /* root: representation -> prediction -> expand (mcts.py:88-123) */
    Pair pol;
    if (kInjected) {
        pol = pair_from_global(p.t_root_policy + (size_t)g * A, A, lane);
    } else if constexpr (kVision) {
        /* vision model: the root was evaluated by k_inference<initial, vision> (search_launch): its hidden
         * state is row 0 of the game's hidden rows, its policy in t_root_policy.  The representation needs
         * three 49 x 49 maps of LDS per wave — one wave per SIMD; the simulations need ~6 KB per wave
         * and run here at up to four waves per SIMD */
        pol = pair_from_global(p.t_root_policy + (size_t)g * A, A, lane);
    } else {
        for (int k = lane; k < m.obs_dim; k += kWave) L.xh[k] = p.obs[(size_t)g * m.obs_dim + k];
        wave_sync();
        model_representation<false>(m, L, p.obs + (size_t)g * m.obs_dim, hid_g, lane);
        (void)model_prediction<false>(m, L, &pol, lane);
    }
    wave_expand(c, L, 0, 0, 0.0f, pol, lane);
    /* add_exploration_noise_at_the_root (mcts.py:134-145): f32 product widened, f64 sum */
    if (lane < A) {
        const float pr = L.nodes[1 + lane].prior;
        if (p.train && S > 0)
            s.root_prior = (double)(pr * p.omf32) + p.dirichlet[(size_t)g * A + lane] * p.frac;
        else
            s.root_prior = (double)pr;
    }

    for (int sim = 0; sim < S; ++sim) {
        int depth, action, pbase;
        const int leaf = wave_select(c, s, L, &depth, &action, &pbase, lane);
        const int e_new = sim + 1;
        float reward, value;
        if (kInjected) {
            reward = p.t_reward[(size_t)g * S + sim];
            value = p.t_value[(size_t)g * S + sim];
            pol = pair_from_global(p.t_prior + ((size_t)g * S + sim) * A, A, lane);
        } else {
            /* dynamics on the parent's hidden state, prediction on the new one (mcts.py:180-192) */
            const int e_parent = (pbase - 1) / A;
            const float* hp = hid_g + (size_t)e_parent * p.t.hid_stride;
            for (int k = lane; k < m.hid; k += kWave) L.xh[k] = hp[k];
            wave_sync();
            reward = model_dynamics<kVision, kVisBatch>(m, L, action, hid_g + (size_t)e_new * p.t.hid_stride, lane);
            value = model_prediction<kVision, kVisBatch>(m, L, &pol, lane);
        }
        wave_expand(c, L, leaf, e_new, reward, pol, lane);
        if (lane == 0) p.t.sim_leaf[(size_t)g * p.t.max_sims + sim] = leaf;
        wave_backup(c, s, L, depth, value, lane);
    }

    /* tree -> HBM structure of arrays; results (mcts.py:215-222) */
    const size_t row = (size_t)g * p.t.stride;
    for (int n = lane; n < p.n_nodes; n += kWave) {
        const Node nd = L.nodes[n];
        p.t.visit[row + n] = nd.visit;
        p.t.vsum[row + n] = nd.vsum;
        p.t.prior[row + n] = nd.prior;
        p.t.reward[row + n] = nd.reward;
        p.t.child_base[row + n] = L.cb[n];
    }
    if (lane < A) {
        p.r.visits[(size_t)g * A + lane] = L.nodes[1 + lane].visit;
        if (p.r.visits_sink) p.r.visits_sink[(size_t)g * A + lane] = L.nodes[1 + lane].visit;
        p.r.root_prior[(size_t)g * A + lane] = s.root_prior;
    }
    if (lane == 0) {
        const Node r = L.nodes[0];
        const float rv = r.visit == 0 ? 0.0f : r.vsum / (float)r.visit;
        p.r.root_value[g] = rv;
        if (p.r.root_value_sink) p.r.root_value_sink[g] = rv;
        p.r.consumed[g] = s.consumed;
        p.r.path_len[g] = s.path_total;
        p.t.minmax[2 * (size_t)g] = s.mn;
        p.t.minmax[2 * (size_t)g + 1] = s.mx;
        p.t.status[g] = s.status;
        if (s.status && p.overflow) *p.overflow = 1;
    }
}

/* ------------------------------------------------------------------------------
 * k_search_vis4: the simulations of a 'vision_model' search, four games per workgroup: the tree work and
 * the convolutions one game per wavefront exactly as in k_search_fused (mzs_vision.h), the three head
 * MLPs of a simulation on the matrix core for the block's four games at once, one head per wavefront
 * (mzs_vision_mm.h).  Every wave of the block reaches both barriers of every simulation: waves whose
 * game lies past n_games replay the last game and write nothing; a redo pass replays whole blocks.
 * The roots come from k_inference<initial, vision> as for k_search_fused.
 * ---------------------------------------------------------------------------- */
template <int kWavesPerSimd>
__global__ void __launch_bounds__(256, kWavesPerSimd) k_search_vis4(const SearchParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    if (redo_skip_block(p, 4)) return;
    WaveLds L = carve(smem + (size_t)wave * p.lds.bytes, p.lds);
    const Model& m = p.m;
    float* shared = reinterpret_cast<float*>(smem + (size_t)4 * p.lds.bytes);
    L.vc = shared;
    const VisMmLds M = vis_mm_carve(shared + kVisCache, m.H, m.Sd, m.A);
    vis_cache_load_block(m.blob, vis_offsets(m), L.vc); /* ends in the block's barrier */
    const int g_raw = blockIdx.x * 4 + wave;
    const bool live = g_raw < p.n_games;
    const int g = live ? g_raw : p.n_games - 1;
    const int A = m.A, S = p.S;

    SearchConst c;
    c.A = A;
    c.S = S;
    c.n_nodes = p.n_nodes;
    c.g32 = p.g32;
    c.pbc = p.pbc;
    c.noise = p.noise ? p.noise + (size_t)g * p.noise_stride : nullptr;
    c.noise_cap = p.noise ? p.noise_avail : 0;

    SearchState s;
    s.mn = __builtin_inff();
    s.mx = -__builtin_inff();
    s.consumed = 0;
    s.path_total = 0;
    s.status = 0;
    s.root_prior = 0.0;
    if (c.noise) {
        noise_refill(c, s, lane);
    } else {
        s.win_base = 0;
        s.win = 0.0;
    }
    for (int n = lane; n < p.n_nodes; n += kWave) {
        Node z;
        z.visit = 0;
        z.vsum = 0.0f;
        z.prior = 0.0f;
        z.reward = 0.0f;
        L.nodes[n] = z;
        L.cb[n] = 0;
    }
    wave_sync();

    float* hid_g = p.t.hidden + (size_t)g * (p.t.max_sims + 1) * p.t.hid_stride;
    Pair pol = pair_from_global(p.t_root_policy + (size_t)g * A, A, lane);
    wave_expand(c, L, 0, 0, 0.0f, pol, lane);
    if (lane < A) {
        const float pr = L.nodes[1 + lane].prior;
        if (p.train && S > 0)
            s.root_prior = (double)(pr * p.omf32) + p.dirichlet[(size_t)g * A + lane] * p.frac;
        else
            s.root_prior = (double)pr;
    }
    float* hx_rew = M.hx + (size_t)(0 * 4 + wave) * M.hxs;
    float* hx_val = M.hx + (size_t)(1 * 4 + wave) * M.hxs;
    float* hx_pol = M.hx + (size_t)(2 * 4 + wave) * M.hxs;

    for (int sim = 0; sim < S; ++sim) {
        int depth, action, pbase;
        const int leaf = wave_select(c, s, L, &depth, &action, &pbase, lane);
        const int e_new = sim + 1;
        const int e_parent = (pbase - 1) / A;
        const float* hp = hid_g + (size_t)e_parent * p.t.hid_stride;
        for (int k = lane; k < m.hid; k += kWave) L.xh[k] = hp[k];
        wave_sync();
        /* dead waves replay the last game: they must not store into its rows while its owner reads them */
        vis_dynamics_conv(m, L, action, live ? hid_g + (size_t)e_new * p.t.hid_stride : nullptr, hx_rew, lane);
        vis_prediction_conv(m, L, hx_val, hx_pol, lane);
        __syncthreads();
        if (wave < 3) vis_mm_head<(kWavesPerSimd >= 4 ? 2 : 4)>(m, M, wave, lane);
        __syncthreads();
        const float reward = seg_support_scalar(M.lg + (size_t)(0 * 4 + wave) * M.ls, m.Sd, lane);
        wave_sync();
        const float value = seg_support_scalar(M.lg + (size_t)(1 * 4 + wave) * M.ls, m.Sd, lane);
        wave_sync();
        pol = seg_softmax(seg_load(M.lg + (size_t)(2 * 4 + wave) * M.ls, A, lane));
        wave_sync();
        wave_expand(c, L, leaf, e_new, reward, pol, lane);
        if (lane == 0 && live) p.t.sim_leaf[(size_t)g * p.t.max_sims + sim] = leaf;
        wave_backup(c, s, L, depth, value, lane);
    }
    if (!live) return;

    const size_t row = (size_t)g * p.t.stride;
    for (int n = lane; n < p.n_nodes; n += kWave) {
        const Node nd = L.nodes[n];
        p.t.visit[row + n] = nd.visit;
        p.t.vsum[row + n] = nd.vsum;
        p.t.prior[row + n] = nd.prior;
        p.t.reward[row + n] = nd.reward;
        p.t.child_base[row + n] = L.cb[n];
    }
    if (lane < A) {
        p.r.visits[(size_t)g * A + lane] = L.nodes[1 + lane].visit;
        if (p.r.visits_sink) p.r.visits_sink[(size_t)g * A + lane] = L.nodes[1 + lane].visit;
        p.r.root_prior[(size_t)g * A + lane] = s.root_prior;
    }
    if (lane == 0) {
        const Node r = L.nodes[0];
        const float rv = r.visit == 0 ? 0.0f : r.vsum / (float)r.visit;
        p.r.root_value[g] = rv;
        if (p.r.root_value_sink) p.r.root_value_sink[g] = rv;
        p.r.consumed[g] = s.consumed;
        p.r.path_len[g] = s.path_total;
        p.t.minmax[2 * (size_t)g] = s.mn;
        p.t.minmax[2 * (size_t)g + 1] = s.mx;
        p.t.status[g] = s.status;
        if (s.status && p.overflow) *p.overflow = 1;
    }
}

/* ------------------------------------------------------------------------------
 * Standalone batched inference (wave per sample): the Muzero.*_inference wrappers
 * (muzero_model.py:828-905).
 * ---------------------------------------------------------------------------- */
struct InferParams {
    Model m;
    LdsLayout lds;
    int batch;
    const float* in;       /* obs [B][obs_dim] (vision: [B][3][98][98]) or hidden [B][hid] */
    const int* action;     /* [B]                                */
    float* reward;         /* [B]                                */
    float* hidden_out;     /* [B][hid]                           */
    float* policy;         /* [B][A]                             */
    float* value;          /* [B]                                */
    long long hidden_stride; /* floats between the rows of hidden_out (0: hid) */
};

enum { kInferInitial = 0, kInferRecurrent = 1, kInferPrediction = 2 };

template <int kMode, bool kVision = false>
__global__ void __launch_bounds__(256) k_inference(const InferParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    const int b = blockIdx.x * (blockDim.x >> 6) + wave;
    if (b >= p.batch) return;
    const WaveLds L = carve(smem + (size_t)wave * p.lds.bytes, p.lds);
    const Model& m = p.m;
    constexpr bool vision = kVision;
    const size_t obs_floats = vision ? (size_t)MZS_VIS_C * m.obs_dim * m.obs_dim : (size_t)m.obs_dim;
    const int n_in = kMode == kInferInitial ? (vision ? 0 : m.obs_dim) : m.hid;
    for (int k = lane; k < n_in; k += kWave) L.xh[k] = p.in[(size_t)b * n_in + k];
    wave_sync();
    if constexpr (kVision) vis_cache_load(m.blob, vis_offsets(m), L.vc, lane);
    if (kMode == kInferInitial) {
        model_representation<kVision>(m, L, p.in + (size_t)b * obs_floats, nullptr, lane);
    } else if (kMode == kInferRecurrent) {
        const float reward = model_dynamics<kVision>(m, L, p.action[b], nullptr, lane);
        if (lane == 0) p.reward[b] = reward;
    }
    if (kMode != kInferPrediction) {
        const size_t hs = p.hidden_stride ? (size_t)p.hidden_stride : (size_t)m.hid;
        for (int k = lane; k < m.hid; k += kWave) p.hidden_out[(size_t)b * hs + k] = L.xh[k];
    }
    if (p.policy || p.value) {
        Pair pol;
        const float value = model_prediction<kVision>(m, L, &pol, lane);
        if (p.policy) pair_to_global(p.policy + (size_t)b * m.A, pol, lane);
        if (p.value && lane == 0) p.value[b] = value;
    }
}

/* ------------------------------------------------------------------------------
 * Per-game random streams on the device: what numpy.random.RandomState(seed) yields for
 * dirichlet([alpha] * A) followed by uniform(lo, hi, size = n) — the numbers the reference
 * draws from NumPy's global legacy stream after np.random.seed(seed)
 * (monte_carlo_tree_search.py:140, :163).  MT19937 (init_genrand seeding, 53-bit doubles from
 * two tempered words), legacy standard_gamma for shape <= 1, its log and pow from include/mzs_crmath.h
 * (double-double, rounded once: the correctly rounded double, which is what glibc's libm under NumPy
 * returns in all but ~0.07 % of the calls — oracle/mz_stream_oracle.c is this sampler on the CPU, equal
 * to the device bit for bit and pinned to NumPy by tests/test_crmath.py).
 * One workgroup = sixteen games.  init_genrand is a serial chain of 624 multiply-xorshift steps per
 * game, so the sixteen chains run side by side in sixteen lanes of wave 0 (states in LDS); then one
 * wavefront per game: the 624-word state refresh, the gamma draws and the uniform fill are spread
 * over its lanes.
 * ---------------------------------------------------------------------------- */
struct FillParams {
    const long long* seeds;
    const int* redo_status; /* null: every game; else only the games whose entry is set ...        */
    int* overflow_flag;     /* ... and none at all if this flag (may be null) is zero; a launch for every
                               game clears it instead                                                  */
    int n_games, A, n_uniform, train; /* n_uniform: draws written per game (a prefix of the stream) */
    long long stride;
    double alpha, lo, range;
    double* dirichlet;
    double* noise;
};

constexpr int kMtN = 624, kMtM = 397;
constexpr int kFillGames = 16;       /* games per workgroup of k_fill_streams          */
constexpr int kMtStride = kMtN + 1;  /* odd stride: the seeding lanes hit distinct banks */

__device__ __forceinline__ unsigned mt_temper(unsigned y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

/* regenerate all 624 words (mt19937_gen of NumPy's legacy generator), 64 at a time.  Word k needs the
 * old words k, k+1 and word k+397 (old) or k-227 (new, written at least three chunks earlier): in
 * ascending chunks every read of a chunk precedes its writes in program order, and LDS operations of
 * one wave execute in order.  Unrolled: the chunk index is a constant, only chunk 3 (which straddles
 * 227) and the last one (which ends at 623 and wraps to word 0) keep per-lane selects. */
__device__ __forceinline__ void mt_twist(unsigned* mt, int lane) {
#pragma unroll
    for (int c = 0; c < (kMtN + kWave - 1) / kWave; ++c) {
        const int kk = c * kWave + lane;
        const bool in = c * kWave + kWave <= kMtN || kk < kMtN;
        const int k1 = c * kWave + kWave < kMtN ? kk + 1 : (kk + 1 >= kMtN ? 0 : kk + 1);
        const int ks = c * kWave + kWave <= kMtN - kMtM ? kk + kMtM
                       : (c * kWave >= kMtN - kMtM ? kk - (kMtN - kMtM)
                                                   : (kk < kMtN - kMtM ? kk + kMtM : kk - (kMtN - kMtM)));
        unsigned v = 0;
        if (in) {
            const unsigned y = (mt[kk] & 0x80000000u) | (mt[k1] & 0x7fffffffu);
            v = mt[ks] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        wave_sync();
        if (in) mt[kk] = v;
        wave_sync();
    }
}

struct MtStream {
    unsigned* mt;
    int pos, lane;
    __device__ __forceinline__ unsigned word() {
        if (pos == kMtN) {
            mt_twist(mt, lane);
            pos = 0;
        }
        return mt_temper(mt[pos++]);
    }
    /* random_double of the legacy generator */
    __device__ __forceinline__ double next_double() {
        const unsigned a = word() >> 5, b = word() >> 6;
        return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
    }
};

/* legacy_standard_gamma, shape <= 1 (numpy/random/src/legacy/legacy-distributions.c) */
__device__ __forceinline__ double mt_standard_gamma(MtStream& r, double shape) {
    if (shape == 1.0) return -mzs_cr_log(1.0 - r.next_double());
    if (shape == 0.0) return 0.0;
    for (;;) {
        const double U = r.next_double();
        const double V = -mzs_cr_log(1.0 - r.next_double());
        if (U <= 1.0 - shape) {
            const double X = mzs_cr_pow(U, 1.0 / shape);
            if (X <= V) return X;
        } else {
            const double Y = -mzs_cr_log((1.0 - U) / shape);
            const double X = mzs_cr_pow(1.0 - shape + shape * Y, 1.0 / shape);
            if (X <= V + Y) return X;
        }
    }
}

/* One rejection attempt of legacy_standard_gamma (0 < shape < 1) on the two doubles U, V' it
 * draws: the value it would return and whether it accepts.  An attempt always consumes exactly
 * two doubles, so attempt number m of a stream reads words [4m, 4m + 4) whatever happened before. */
__device__ __forceinline__ bool mt_gamma_attempt(double U, double Vd, double shape, double* X) {
    /* both arms of the sampler end in pow(., 1 / shape): select the base, call pow once */
    const double V = -mzs_cr_log(1.0 - Vd);
    const bool low = U <= 1.0 - shape;
    const double Y = low ? 0.0 : -mzs_cr_log((1.0 - U) / shape);
    *X = mzs_cr_pow(low ? U : 1.0 - shape + shape * Y, 1.0 / shape);
    return *X <= V + Y; /* low arm: X <= V */
}

__global__ void __launch_bounds__(64 * kFillGames) k_fill_streams(const FillParams p) {
    __shared__ unsigned mt_s[kFillGames][kMtStride];
    __shared__ double gam_s[kFillGames][64];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int g = blockIdx.x * kFillGames + wave;
    bool mine = g < p.n_games;
    if (p.redo_status) { /* the games of a prefix search that asked for more; usually none */
        if (p.overflow_flag && *p.overflow_flag == 0) return;
        mine = mine && p.redo_status[g] != 0;
        if (!__syncthreads_or(mine ? 1 : 0)) return;
    }
    else if (p.overflow_flag && blockIdx.x == 0 && threadIdx.x == 0) {
        *p.overflow_flag = 0; /* the search that follows raises it */
    }
    /* mt19937_seed (init_genrand) of the block's sixteen games, one per lane */
    if (wave == 0 && lane < kFillGames && blockIdx.x * kFillGames + lane < p.n_games) {
        unsigned s = (unsigned)p.seeds[blockIdx.x * kFillGames + lane];
        unsigned* mt = mt_s[lane];
#pragma unroll 8
        for (int i = 0; i < kMtN; ++i) {
            mt[i] = s;
            s = 1812433253u * (s ^ (s >> 30)) + (unsigned)i + 1u;
        }
    }
    __syncthreads();
    if (!mine) return; /* no block-level barrier below */
    MtStream r;
    r.mt = mt_s[wave];
    r.lane = lane;
    mt_twist(r.mt, lane);
    r.pos = 0;
    if (p.train) {
        /* RandomState.dirichlet: gamma draws, then normalise by the reciprocal of their sum */
        double* val = gam_s[wave];
        int a_done = 0;
        if (p.alpha < 1.0) {
            /* lane m evaluates attempt m of the stream (words 4m .. 4m+3 of the first 624): the A gamma
             * values are the first A accepted attempts, in order */
            const unsigned w0 = mt_temper(r.mt[4 * lane]) >> 5, w1 = mt_temper(r.mt[4 * lane + 1]) >> 6;
            const unsigned w2 = mt_temper(r.mt[4 * lane + 2]) >> 5, w3 = mt_temper(r.mt[4 * lane + 3]) >> 6;
            const double U = ((double)w0 * 67108864.0 + (double)w1) / 9007199254740992.0;
            const double Vd = ((double)w2 * 67108864.0 + (double)w3) / 9007199254740992.0;
            double X;
            const bool ok = mt_gamma_attempt(U, Vd, p.alpha, &X);
            unsigned long long mask = __builtin_amdgcn_ballot_w64(ok);
            while (a_done < p.A && mask) {
                const int mth = __ffsll((long long)mask) - 1;
                const double v = readlane_f64(X, mth);
                if (lane == 0) val[a_done] = v;
                ++a_done;
                r.pos = 4 * (mth + 1);
                mask &= mth == 63 ? 0ull : ~((2ull << mth) - 1ull);
            }
        }
        /* shape 1, or (never seen) fewer than A acceptances among 64 attempts: the scalar sampler */
        for (int a = a_done; a < p.A; ++a) {
            const double v = mt_standard_gamma(r, p.alpha);
            if (lane == 0) val[a] = v;
        }
        wave_sync();
        double acc = 0.0;
        for (int a = 0; a < p.A; ++a) acc += val[a];
        const double inv = 1.0 / acc;
        for (int a = lane; a < p.A; a += kWave) p.dirichlet[(size_t)g * p.A + a] = val[a] * inv;
    }
    /* uniform(lo, hi): lo + (hi - lo) * random_double, two words per draw, lanes in parallel */
    double* out = p.noise + (size_t)g * p.stride;
    int done = 0;
    while (done < p.n_uniform) {
        if (r.pos == kMtN) {
            mt_twist(r.mt, lane);
            r.pos = 0;
        }
        const int avail = (kMtN - r.pos) / 2;
        const int n = min(avail, p.n_uniform - done);
        for (int i = lane; i < n; i += kWave) {
            const unsigned a = mt_temper(r.mt[r.pos + 2 * i]) >> 5;
            const unsigned b = mt_temper(r.mt[r.pos + 2 * i + 1]) >> 6;
            const double d = ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
            const double sc = p.range * d;
            out[done + i] = p.lo + sc;
        }
        r.pos += 2 * n;
        done += n;
        wave_sync();
    }
}

}  // namespace

/* ================================================================== host side */

struct mzs_handle {
    mzs_config cfg;
    Model model;
    LdsLayout lds;
    TreeArrays tree;
    Results res;
    float* w_dev = nullptr;        /* weight store, fp32 (the training kernels' forward pass reads this one) */
    float* w_search_dev = nullptr; /* what search / inference kernels read: w_dev, or for MZS_PRECISION_BF16 a
                                      copy with every weight (not bias) rounded to bfloat16          */
    float* blob_dev = nullptr;     /* master weights in blob order (include/mzs_layout.h)      */
    int* store_index_dev = nullptr; /* blob element -> offset in the device weight store        */
    double* pbc_dev = nullptr;
    int* overflow_dev = nullptr;   /* set by a search that ran out of tie-break draws (mzs_search_streams) */
    float* root_policy_dev = nullptr; /* vision model: [max_games][A] root policies (k_inference -> k_search_fused) */
    float* stream_dev = nullptr;   /* chunk images (k_build_stream) of the models that have a quadw / quadl kernel */
    StreamPlan stream_plan;
    /* training (mzs_train_*): allocated by mzs_train_create */
    int train_batch = 0, train_K = 0, train_step = 0;
    TapeLayout tape_layout;
    float* tape_dev = nullptr;
    float* loss_buf_dev = nullptr;
    float* adam_m_dev = nullptr;
    float* adam_v_dev = nullptr;
    double* wgrad_partial_dev = nullptr;
    LstmTrainParams lstm_train;    /* 'lstm_model': tape layout and descriptor tables of mzs_train_create */
    VisTrain* vis_train = nullptr; /* 'vision_model': mzs_vision_train.hip */
    int* vis_store_src_dev = nullptr; /* vision: element of the Linear stores (w_dev) -> blob index, -1 = padding */
    const unsigned char* dropout_masks = nullptr; /* mzs_train_set_dropout: injected masks for the NEXT gradient call */
    int64_t dropout_n = 0;
    uint64_t dropout_seed = 0x5DEECE66Dull, train_calls = 0;
    int loss_value = MZS_LOSS_CE, loss_reward = MZS_LOSS_CE; /* mzs_train_set_loss */
    double momentum = 0.9;                                   /* mzs_train_set_momentum */
    size_t w_floats = 0;
    bool have_weights = false;
    int n_nodes_max = 0;
    int n_cus = 256;
    std::vector<void*> allocs;
};

namespace {

template <typename T>
int dev_alloc(mzs_handle* h, T** out, size_t n) {
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, n * sizeof(T) + 256));
    HIP_TRY(hipMemset(p, 0, n * sizeof(T) + 256));
    h->allocs.push_back(p);
    *out = static_cast<T*>(p);
    return MZS_OK;
}

/* Device weight store: per layer W^T [rows][n_pad] + bias [n_pad] (see mzs_device.h Layer) */
constexpr int kStoreBiasFlag = 1 << 30; /* store_index entry of a bias element (never rounded to bf16) */

struct StoreBuilder {
    std::vector<float> data;
    std::vector<std::pair<size_t, size_t>> bias_ranges; /* [begin, end) of the bias vectors in data */
    Layer add(int rows, int k, int n_pad) {
        Layer ly;
        ly.k = k;
        ly.n_pad = n_pad;
        ly.w_off = (int)data.size();
        data.resize(data.size() + (size_t)rows * n_pad, 0.0f);
        ly.b_off = (int)data.size();
        data.resize(data.size() + (size_t)n_pad, 0.0f);
        bias_ranges.emplace_back((size_t)ly.b_off, data.size());
        return ly;
    }
    /* the store search / inference kernels read under MZS_PRECISION_BF16: weights rounded, biases kept */
    std::vector<float> rounded_bf16() const {
        std::vector<float> r(data.size());
        for (size_t i = 0; i < data.size(); ++i) r[i] = mzs_bf16_round(data[i]);
        for (const auto& br : bias_ranges)
            for (size_t i = br.first; i < br.second; ++i) r[i] = data[i];
        return r;
    }
    std::vector<int> index; /* blob element -> store offset (sized by the caller) */
    size_t w0 = 0, b0 = 0;  /* blob offsets of the tensor being filled             */
    /* W is torch layout [n_out][ld]; copy rows [0,rows) of W^T into columns [col0, col0+n_out) */
    void fill(const Layer& ly, const float* W, const float* b, int n_out, int rows, int ld, int col0) {
        for (int j = 0; j < n_out; ++j) {
            for (int k = 0; k < rows; ++k) {
                const size_t so = (size_t)ly.w_off + (size_t)k * ly.n_pad + col0 + j;
                data[so] = W[(size_t)j * ld + k];
                if (!index.empty()) index[w0 + (size_t)j * ld + k] = (int)so;
            }
            data[(size_t)ly.b_off + col0 + j] = b[j];
            if (!index.empty()) index[b0 + j] = (int)((size_t)ly.b_off + col0 + j) | kStoreBiasFlag;
        }
    }
    /* tensor at blob offsets (ow, ob) */
    void fill_at(const Layer& ly, const float* B, size_t ow, size_t ob, int n_out, int rows, int ld, int col0) {
        w0 = ow, b0 = ob;
        fill(ly, B + ow, B + ob, n_out, rows, ld, col0);
    }
};

/* 'lstm_model' head: Linear(in_dim -> H) + per layer the fused (i | g | o) gate projection */
LstmHead describe_lstm_head(StoreBuilder* sb, int lin_rows, int lin_k, int H, int n_out, int n_layers) {
    LstmHead hd;
    hd.n_out = n_out;
    hd.out_pad = round_up(n_out, 64);
    hd.lin = sb->add(lin_rows, lin_k, round_up(H, 64));
    for (int l = 0; l < kLstmMaxLayers; ++l) {
        hd.gate[l] = Layer{0, 0, 0, 0};
        hd.bhh_off[l] = 0;
        if (l >= n_layers) continue;
        const int in_l = l == 0 ? H : n_out;
        hd.gate[l] = sb->add(in_l, in_l, 3 * hd.out_pad);
        hd.bhh_off[l] = (int)sb->data.size();
        sb->data.resize(sb->data.size() + (size_t)3 * hd.out_pad, 0.0f);
        sb->bias_ranges.emplace_back((size_t)hd.bhh_off[l], sb->data.size());
    }
    return hd;
}

void fill_lstm_head(StoreBuilder* sb, const LstmHead& hd, const float* B, const mzs_lstm_head_offsets& o, int H,
                    int n_layers) {
    const int n = hd.n_out;
    sb->fill_at(hd.lin, B, o.lin_w, o.lin_b, H, o.in_dim, o.in_dim, 0);
    for (int l = 0; l < n_layers; ++l) {
        const int in_l = l == 0 ? H : n;
        const int gate_row[3] = {0, 2, 3}; /* torch rows i, f, g, o: f never contributes (c0 = 0) */
        for (int q = 0; q < 3; ++q) {
            const size_t r0 = (size_t)gate_row[q] * n;
            sb->fill_at(hd.gate[l], B, o.w_ih[l] + r0 * in_l, o.b_ih[l] + r0, n, in_l, in_l, q * hd.out_pad);
            for (int j = 0; j < n; ++j) {
                const size_t so = (size_t)hd.bhh_off[l] + q * hd.out_pad + j;
                sb->data[so] = B[o.b_hh[l] + r0 + j];
                if (!sb->index.empty()) sb->index[o.b_hh[l] + r0 + j] = (int)so | kStoreBiasFlag;
            }
        }
    }
}

void describe_model(const mzs_dims& d, Model* m, StoreBuilder* sb) {
    const int H = d.hidden_dim, Sd = d.state_dim, A = d.n_actions, HP = round_up(H, 64);
    m->obs_dim = d.obs_dim;
    m->A = A;
    m->Sd = Sd;
    m->H = H;
    m->L = d.n_mid_layers;
    m->kind = d.model_kind;
    m->bf16 = d.precision; /* 0, MZS_PRECISION_BF16 or MZS_PRECISION_BF16X3 */
    m->seg_state = round_up(Sd, 32);
    m->seg_value = Sd > 32 ? 64 : round_up(A, 32); /* wide heads: one 64-neuron group each (mzs_quadw.h) */
    m->hid = mzs_hidden_floats(&d);
    m->blob = nullptr;
    if (d.model_kind == MZS_MODEL_VISION) { /* reads the blob itself: no transposed store */
        const Layer none{0, 0, 0, 0};
        m->r_in = m->r_mid = m->r_out = m->d_in = m->d_mid = m->d_heads = m->p_in = m->p_mid = m->p_heads = none;
        return;
    }
    if (d.model_kind == MZS_MODEL_LSTM) {
        const Layer none{0, 0, 0, 0};
        m->r_mid = m->r_out = m->d_in = m->d_mid = m->d_heads = m->p_in = m->p_mid = m->p_heads = none;
        m->r_in = sb->add(d.obs_dim, d.obs_dim, round_up(Sd, 64));
        m->l_rew = describe_lstm_head(sb, Sd + A, Sd, H, Sd, d.n_mid_layers);
        m->l_st = describe_lstm_head(sb, Sd + A, Sd, H, Sd, d.n_mid_layers);
        m->l_pol = describe_lstm_head(sb, Sd, Sd, H, A, d.n_mid_layers);
        m->l_val = describe_lstm_head(sb, Sd, Sd, H, Sd, d.n_mid_layers);
        return;
    }
    m->r_in = sb->add(d.obs_dim, d.obs_dim, HP);
    m->r_mid = sb->add(H, H, HP);
    m->r_out = sb->add(H, H, round_up(Sd, 64));
    m->d_in = sb->add(Sd + A, Sd, HP);
    m->d_mid = sb->add(H, H, HP);
    m->d_heads = sb->add(H, H, round_up(m->seg_state + Sd, 64));
    m->p_in = sb->add(Sd, Sd, HP);
    m->p_mid = sb->add(H, H, HP);
    m->p_heads = sb->add(H, H, round_up(m->seg_value + Sd, 64));
}

/* vis_sims_only (vision model): 0 = every map of the representation, 1 = the simulations of k_search_fused,
 * 2 = the simulations of k_search_vis4 */
LdsLayout make_lds_layout(const Model& m, int n_nodes, int S, int vis_sims_only = 0) {
    LdsLayout l;
    int off = round_up(n_nodes * (int)sizeof(Node), 16);
    l.off_cb = off;
    off += round_up(n_nodes * 2, 16);
    l.off_path = off;
    off += round_up((S + 2) * 2, 16);
    l.off_xh = off;
    off += round_up((m.obs_dim > m.Sd ? m.obs_dim : m.Sd) * 4, 16);
    int act = round_up(m.H, 64); /* trunk activations; the LSTM heads also keep h (<= out_pad) there */
    int ob = m.r_out.n_pad;
    ob = m.d_heads.n_pad > ob ? m.d_heads.n_pad : ob;
    ob = m.p_heads.n_pad > ob ? m.p_heads.n_pad : ob;
    if (m.kind == MZS_MODEL_LSTM) {
        const int out_pad = round_up(m.Sd > m.A ? m.Sd : m.A, 64);
        act = act > out_pad ? act : out_pad;
        ob = 3 * out_pad;
    }
    if (m.kind == MZS_MODEL_VISION) { /* mzs_vision.h: three maps of up to 49 x 49 floats, logits behind the third's head */
        off = l.off_xh + kVisXh * 4;
        if (vis_sims_only == 2) { /* k_search_vis4: the 7 x 7 maps only (the head MLPs live in block-shared LDS) */
            act = 256;
            ob = 160;
        } else if (vis_sims_only) { /* dynamics / prediction on 7 x 7 maps: 4 x 49 floats, H activations, logits at ob + 256 */
            act = act > 256 ? act : 256;
            ob = 256 + 128;
        } else {
            act = kVisPadBuf > act ? kVisPadBuf : act; /* xa holds the 49 x 49 map with its border */
            ob = kVisBuf;
        }
    }
    l.off_xa = off;
    off += act * 4;
    l.off_xb = off;
    off += act * 4;
    l.off_ob = off;
    off += ob * 4;
    l.off_vc = off;
    l.shared_bytes = 0;
    if (m.kind == MZS_MODEL_VISION) {
        if (vis_sims_only)
            l.shared_bytes = kVisCache * 4; /* one copy per workgroup (k_search_fused) */
        else
            off += kVisCache * 4;
    }
    l.bytes = round_up(off, 16);
    return l;
}

constexpr int kMaxLdsPerBlock = 160 * 1024;

/* waves per block so that the block's LDS fits; 0 if even one wave does not */
int waves_per_block(const LdsLayout& l) {
    int w = 4;
    while (w > 0 && w * l.bytes + l.shared_bytes > kMaxLdsPerBlock) --w;
    return w;
}

template <typename K, typename P>
int launch(K kernel, const P& params, int units, const LdsLayout& lds, hipStream_t st) {
    const int wpb = waves_per_block(lds);
    if (wpb == 0) return fail(MZS_E_ARG, "problem too large for one wave's LDS slice");
    const int blocks = (units + wpb - 1) / wpb;
    const size_t shmem = (size_t)wpb * lds.bytes + lds.shared_bytes;
    if (shmem > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(wpb * kWave), shmem, st, params);
    HIP_TRY(hipGetLastError());
    return MZS_OK;
}

/* LDS plan of k_search_quad (mzs_quad.h QuadLayout) */
QuadLayout make_quad_layout(int A, int n_nodes, int S, bool with_model, bool hid_lds = true) {
    QuadLayout l;
    l.off_inv64 = round_up((S + 2) * 8, 16);
    l.off_inv32 = 2 * l.off_inv64;
    l.off_onehot = l.off_inv32 + round_up((S + 2) * 4, 16);
    l.off_tabs = 0;
    l.shared_bytes = l.off_onehot + (with_model ? A * 64 * 4 : 0);
    l.off_nwin = 4 * kTrunkStride * 4 + (with_model && !hid_lds ? 4 * kHidRowStride * 4 : 0); /* + one hidden-state row per game */
    l.off_games = l.off_nwin + 64 * 8;
    int off = n_nodes * (int)sizeof(QNode);
    l.g_vsum = off;
    off += round_up(n_nodes * 4, 16);
    l.g_path = off;
    off += round_up((S + 2) * 8, 16);
    l.g_hid = off;
    off += with_model && hid_lds ? (S + 1) * 32 * 4 : 0;
    l.g_leafs = off;
    off += round_up((S + 1) * 2, 16);
    /* game records 16 bytes past a multiple of 32: the hidden-state rows (128 bytes each) of the four
     * games then sit 4 banks apart (mod 8), and a ds_read_b128 of the B operands — four games x two
     * lane groups, whatever expansions the games read — touches disjoint banks */
    while (off % 32 != 16) off += 16;
    l.g_bytes = off;
    l.wave_bytes = l.off_games + 4 * l.g_bytes;
    return l;
}

/* LDS plan of k_search_quadw / k_search_quadl: block-shared tables (incl. the small vectors of the
 * simulation loop), trunk [4][128] and xin [4][64] per wave, trees per game, no hidden states (they
 * live in the global SoA array) */
QuadLayout make_quadw_layout(int n_nodes, int S, int tab_floats, int pair_floats = 0) {
    QuadLayout l;
    l.off_inv64 = round_up((S + 2) * 8, 16);
    l.off_inv32 = 2 * l.off_inv64;
    l.off_onehot = l.off_inv32 + round_up((S + 2) * 4, 16);
    l.off_tabs = l.off_onehot;
    l.shared_bytes = l.off_tabs + round_up(tab_floats * 4, 16) + 2 * pair_floats * 4; /* two pairs of waves per block */
    l.off_nwin = 4 * kWideTStride * 4 + 4 * kWideXStride * 4;
    l.off_games = l.off_nwin + 64 * 8;
    int off = n_nodes * (int)sizeof(QNode);
    l.g_vsum = off;
    off += round_up(n_nodes * 4, 16);
    l.g_path = off;
    off += round_up((S + 2) * 8, 16);
    l.g_hid = off;
    l.g_leafs = off;
    off += round_up((S + 1) * 2, 16);
    l.g_bytes = off;
    l.wave_bytes = l.off_games + 4 * l.g_bytes;
    return l;
}

/* the (S_dim, A, H) shapes the four-games-per-wave kernel is instantiated for */
template <bool kHidG>
quad_kernel_t pick_quad_kernel_t(const mzs_dims& d) {
    if (d.obs_dim > 64 || d.model_kind != MZS_MODEL_MLP) return nullptr;
    const bool mid = d.n_mid_layers != 0; /* tied mid layers (any count): H more register-resident rows per net */
    if (d.state_dim == 31 && d.hidden_dim == 64) {
        /* without mid layers: the instantiations of mzsearch_f32v.hip (matrix results in VGPRs: -1.6 % on the
         * benchmark shape; the tied-mid kernels, whose weights fill the accumulator file, lose 10 % that way) */
        if (!mid) return mzs::pick_quad_kernel_f32v(d, kHidG);
        switch (d.n_actions) {
            case 2: return k_search_quad<31, 2, 64, false, true, kHidG>;
            case 3: return k_search_quad<31, 3, 64, false, true, kHidG>;
            case 4: return k_search_quad<31, 4, 64, false, true, kHidG>;
            default: return nullptr;
        }
    }
    if (d.state_dim == 11 && d.n_actions == 2 && d.hidden_dim == 16)
        return mid ? k_search_quad<11, 2, 16, false, true, kHidG> : k_search_quad<11, 2, 16, false, false, kHidG>;
    return nullptr;
}
/* hid_global: the variant that keeps the expansions' hidden states in the global SoA array */
quad_kernel_t pick_quad_kernel(const mzs_dims& d, bool hid_global = false) {
    if (d.precision != MZS_PRECISION_F32) return mzs::pick_quad_kernel_bf16(d, hid_global); /* mzsearch_bf16.hip */
    return hid_global ? pick_quad_kernel_t<true>(d) : pick_quad_kernel_t<false>(d);
}

/* the kernels compiled for two wavefronts per SIMD (hidden states in the global array, no tied mid
 * layers); fp32: ~35 register reloads from scratch per simulation, paid back by the second wave */
quad_kernel_t pick_quad_kernel_occ2(const mzs_dims& d) {
    if (d.precision != MZS_PRECISION_F32) return mzs::pick_quad_kernel_bf16(d, true, true);
    if (d.obs_dim > 64 || d.model_kind != MZS_MODEL_MLP || d.n_mid_layers != 0) return nullptr;
    if (d.state_dim == 31 && d.hidden_dim == 64) {
        if (d.n_actions == 2) return k_search_quad<31, 2, 64, false, false, true, 0, 2>;
        if (d.n_actions == 4) return k_search_quad<31, 4, 64, false, false, true, 0, 2>;
    }
    return nullptr;
}

/* wide MLPs (mzs_quadw.h): the shape of experiments 127 / 128 */
quad_kernel_t pick_quadw_kernel(const mzs_dims& d) {
    if (d.model_kind != MZS_MODEL_MLP || d.obs_dim > 64) return nullptr;
    if (d.state_dim == 61 && d.hidden_dim == 128 && d.n_actions == 2)
        return d.precision != MZS_PRECISION_F32 ? mzs::pick_wide_kernel_bf16(d) : mzs::pick_wide_kernel_f32v(d);
    return nullptr;
}

/* 'lstm_model' on the weight stream (mzs_quadl.h): one LSTM layer, the shape of the LSTM fixture */
quad_kernel_t pick_quadl_kernel(const mzs_dims& d) {
    if (d.model_kind != MZS_MODEL_LSTM || d.n_mid_layers != 1 || d.obs_dim > 64) return nullptr;
    if (d.state_dim == 31 && d.hidden_dim == 64 && d.n_actions == 2)
        return d.precision != MZS_PRECISION_F32 ? mzs::pick_wide_kernel_bf16(d) : mzs::pick_wide_kernel_f32v(d);
    return nullptr;
}

/* the chunk images of one simulation in the order the kernels consume them (n = 0: the model has
 * no such kernel, or more images than kMaxImages) */
StreamPlan make_stream_plan(const mzs_dims& d, const Model& m) {
    StreamPlan pl;
    pl.n = pl.first = 0;
    std::vector<StreamUnit> v;
    const auto unit = [&](const Layer& ly, int col, int K) {
        for (int k0 = 0; k0 < K; k0 += 64) v.push_back(StreamUnit{ly.w_off + col, ly.n_pad, K, k0});
    };
    if (pick_quadw_kernel(d)) {
        /* k_search_quadw2: wave u of a pair consumes output group u of every layer — its images first
         * for u = 0, then for u = 1; `first` = where the prediction net starts within a wave's half */
        for (int u = 0; u < 2; ++u) {
            const auto net = [&](const Layer& in, const Layer& mid, const Layer& heads) {
                unit(in, 64 * u, m.Sd);
                for (int l = 0; l < m.L; ++l) unit(mid, 64 * u, m.H);
                unit(heads, 64 * u, m.H);
            };
            const int base = (int)v.size();
            net(m.d_in, m.d_mid, m.d_heads);
            pl.first = (int)v.size() - base;
            net(m.p_in, m.p_mid, m.p_heads);
        }
    } else if (pick_quadl_kernel(d)) {
        const auto head = [&](const LstmHead& hd) {
            unit(hd.lin, 0, m.Sd);
            unit(hd.gate[0], 0, m.H), unit(hd.gate[0], 64, m.H), unit(hd.gate[0], 128, m.H);
        };
        head(m.l_rew), head(m.l_st);
        pl.first = (int)v.size();
        head(m.l_pol), head(m.l_val);
    }
    if (v.empty() || (int)v.size() > kMaxImages) return pl;
    pl.n = (int)v.size();
    for (int i = 0; i < pl.n; ++i) pl.u[i] = v[i];
    return pl;
}

quad_kernel_t pick_quad_injected(int A) {
    if (A == 2) return k_search_quad<31, 2, 64, true>;
    if (A == 3) return k_search_quad<31, 3, 64, true>;
    if (A == 4) return k_search_quad<31, 4, 64, true>;
    return nullptr;
}

/* waves per block that fit the LDS; 0 = the trees of four games do not fit at all */
int quad_waves_per_block(const QuadLayout& l) {
    int w = 4;
    while (w > 0 && l.shared_bytes + w * l.wave_bytes > kMaxLdsPerBlock) --w;
    return w;
}

int launch_quad(quad_kernel_t kernel, const SearchParams& p, const QuadLayout& ql, int wpb, hipStream_t st) {
    const size_t shmem = (size_t)ql.shared_bytes + (size_t)wpb * ql.wave_bytes;
    if (shmem > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    const int per_block = 4 * wpb;
    const int blocks = (p.n_games + per_block - 1) / per_block;
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(wpb * kWave), shmem, st, p, ql);
    HIP_TRY(hipGetLastError());
    return MZS_OK;
}

/* what mzs_search launches for a search of S simulations on this handle: the one place the
 * dispatch lives (mzs_search and mzs_search_kernel_name both read it) */
enum { kPlanGeneric = 0, kPlanQuad = 1, kPlanStream = 2 };

/* vision model: the simulations run in k_search_vis4 (head MLPs on the matrix core, four games per block) when
 * the head buffers fit its LDS plan; MZS_ENGINE_FUSED_GENERIC keeps k_search_fused (the cross-check) */
bool vis4_ok(const mzs_handle* h, int n_games) {
    const mzs_dims& d = h->cfg.dims;
    if (!(d.model_kind == MZS_MODEL_VISION && h->cfg.engine == MZS_ENGINE_FUSED && d.hidden_dim <= 256 &&
          d.state_dim <= 128 && d.n_actions <= 64))
        return false;
    /* Measured (tools/bench_vision.py, visA, 50 simulations; profiles/r3_vision_summary.md): up to two blocks
     * per CU k_search_vis4 is the faster kernel (1024 games 1.25 ms against 1.43, 2048 1.80 against 1.89);
     * beyond, k_search_fused at four waves per SIMD is (4096: 2.99 against 3.43 at any occupancy of vis4: its
     * barriers hold a block's waves in step and a quarter of them idle in the head phase, which costs more
     * than the shared weight rows save once the SIMDs are full).  MZS_VIS4_OCC (measurement only) lifts this. */
    return (n_games + 3) / 4 <= 2 * h->n_cus || std::getenv("MZS_VIS4_OCC") != nullptr;
}
struct SearchPlan {
    int kind;
    quad_kernel_t kernel;
    QuadLayout ql;
    int wpb;
    char name[96];
};

SearchPlan plan_search(const mzs_handle* h, int S, int n_games) {
    SearchPlan pl;
    std::memset(&pl, 0, sizeof pl);
    const mzs_dims& d = h->cfg.dims;
    const int A = d.n_actions, n_nodes = mzs_nodes_per_game(A, S);
    pl.kind = kPlanGeneric;
    std::snprintf(pl.name, sizeof pl.name,
                  d.model_kind == MZS_MODEL_VISION ? (vis4_ok(h, n_games) ? "k_search_vis4" : "k_search_fused<false,vision>")
                                                   : "k_search_fused<false>");
    if (h->cfg.engine != MZS_ENGINE_FUSED || mzs_max_draws(A, S) >= (1ll << 31)) return pl;
    static const bool no_occ2 = std::getenv("MZS_NO_OCC2") != nullptr; /* measurement knob: same results either way */
    if ((n_games + 15) / 16 > h->n_cus && !no_occ2) {
        /* more blocks than CUs: the kernel compiled for two wavefronts per SIMD, two blocks per CU */
        const QuadLayout qg = make_quad_layout(A, n_nodes, S, true, false);
        quad_kernel_t k2 = pick_quad_kernel_occ2(d);
        if (k2 && quad_waves_per_block(qg) == 4 && qg.shared_bytes + 4 * qg.wave_bytes <= kMaxLdsPerBlock / 2) {
            pl.kind = kPlanQuad;
            pl.kernel = k2;
            pl.ql = qg;
            pl.wpb = 4;
            std::snprintf(pl.name, sizeof pl.name, "k_search_quad<%d,%d,%d,false,%s,true,%s2 waves/SIMD>", d.state_dim, A,
                          d.hidden_dim, d.n_mid_layers ? "true" : "false",
                          d.precision == MZS_PRECISION_BF16 ? "bf16," : (d.precision == MZS_PRECISION_BF16X3 ? "bf16x3," : ""));
            return pl;
        }
    }
    if (quad_kernel_t qk = pick_quad_kernel(d)) {
        /* sixteen games per CU with their hidden states in LDS if that fits; else the variant
         * with the hidden states in the global array, if it seats more waves per block */
        const QuadLayout ql = make_quad_layout(A, n_nodes, S, true);
        const int wpb = quad_waves_per_block(ql);
        const QuadLayout qg = make_quad_layout(A, n_nodes, S, true, false);
        const int wpg = wpb < 4 ? quad_waves_per_block(qg) : 0;
        const bool hid_global = wpg > wpb;
        if (hid_global || wpb > 0) {
            pl.kind = kPlanQuad;
            pl.kernel = hid_global ? pick_quad_kernel(d, true) : qk;
            pl.ql = hid_global ? qg : ql;
            pl.wpb = hid_global ? wpg : wpb;
            std::snprintf(pl.name, sizeof pl.name, "k_search_quad<%d,%d,%d,false,%s,%s%s>", d.state_dim, A,
                          d.hidden_dim, d.n_mid_layers ? "true" : "false", hid_global ? "true" : "false",
                          d.precision == MZS_PRECISION_BF16 ? ",bf16" : (d.precision == MZS_PRECISION_BF16X3 ? ",bf16x3" : ""));
            return pl;
        }
    }
    quad_kernel_t wide = pick_quadw_kernel(d);
    const bool lstm = !wide;
    if (!wide) wide = pick_quadl_kernel(d);
    if (wide && h->stream_plan.n > 0) { /* n == 0: more chunks per simulation than kMaxImages */
        const int tab_floats = d.model_kind == MZS_MODEL_LSTM ? 4 * kLstmHeadTabFloats + 2 * A * 64
                                                              : kWideTabFloats + A * 128;
        const QuadLayout ql = make_quadw_layout(n_nodes, S, tab_floats, lstm ? 0 : kPairFloats);
        int wpb = quad_waves_per_block(ql);
        if (!lstm) wpb &= ~1; /* k_search_quadw2 runs whole pairs of waves */
        if (wpb > 0) {
            pl.kind = kPlanStream;
            pl.kernel = wide;
            pl.ql = ql;
            pl.wpb = wpb;
            std::snprintf(pl.name, sizeof pl.name, "k_search_quad%s<%d,%d,%d%s>", lstm ? "l" : "w2", d.state_dim, A,
                          d.hidden_dim,
                          d.precision == MZS_PRECISION_BF16 ? ",bf16" : (d.precision == MZS_PRECISION_BF16X3 ? ",bf16x3" : ""));
        }
    }
    return pl;
}

int check_search_args(mzs_handle* h, int n_games, int S, int train, const double* dirichlet,
                      const double* noise, int64_t stride) {
    if (!h) return fail(MZS_E_ARG, "null handle");
    if (n_games < 0 || n_games > h->cfg.max_games)
        return fail(MZS_E_ARG, "n_games exceeds max_games of the handle");
    if (S < 0 || S > h->cfg.max_simulations)
        return fail(MZS_E_ARG, "num_simulations exceeds max_simulations of the handle");
    if (S > 0 && (!noise || stride < mzs_max_draws(h->cfg.dims.n_actions, S)))
        return fail(MZS_E_ARG, "noise stream missing or shorter than mzs_max_draws(A, S)");
    if (S > 0 && train && !dirichlet) return fail(MZS_E_ARG, "train=1 needs dirichlet_dev");
    return MZS_OK;
}

void fill_search_params(mzs_handle* h, SearchParams* p, int n_games, int S, int train,
                        const double* dirichlet, const double* noise, int64_t stride) {
    std::memset(p, 0, sizeof(*p));
    p->m = h->model;
    p->n_nodes = mzs_nodes_per_game(h->cfg.dims.n_actions, S);
    p->lds = make_lds_layout(h->model, p->n_nodes, S);
    p->t = h->tree;
    p->r = h->res;
    p->n_games = n_games;
    p->S = S;
    p->train = train;
    p->g32 = (float)h->cfg.discount;
    p->omf32 = (float)(1.0 - h->cfg.root_exploration_fraction);
    p->frac = h->cfg.root_exploration_fraction;
    p->pbc = h->pbc_dev;
    p->dirichlet = dirichlet;
    p->noise = S > 0 ? noise : nullptr;
    p->noise_stride = stride;
    p->noise_avail = stride;
    p->overflow = nullptr;
    p->redo = 0;
}

}  // namespace

extern "C" {

const char* mzs_last_error(void) { return g_err.c_str(); }

#ifdef MZS_DIAG
/* diagnostic build only (not declared in mzsearch.h): read and clear the phase counters */
int mzs_debug_counters(unsigned long long* out8) {
    if (hipDeviceSynchronize() != hipSuccess) return MZS_E_HIP;
    for (int i = 0; i < 8; ++i) out8[i] = 0;
    return diag_drain(out8) && mzs::diag_drain_bf16(out8) && mzs::diag_drain_x3w(out8) && mzs::diag_drain_f32v(out8) ? MZS_OK
                                                                                                                  : MZS_E_HIP;
}
#endif
namespace {
__global__ void k_selftest_reciprocals(int n_max, double* out) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x + 1;
    if (d <= n_max) out[d - 1] = quad_rcp_count(d);
}
}  // namespace

int mzs_selftest_reciprocals(int32_t device, int32_t n_max, int32_t* n_bad) {
    if (n_max < 1 || !n_bad) return fail(MZS_E_ARG, "bad arguments");
    HIP_TRY(hipSetDevice(device));
    double* dev = nullptr;
    HIP_TRY(hipMalloc(&dev, (size_t)n_max * sizeof(double)));
    hipLaunchKernelGGL(k_selftest_reciprocals, dim3((n_max + 255) / 256), dim3(256), 0, nullptr, n_max, dev);
    std::vector<double> host((size_t)n_max);
    hipError_t e = hipMemcpy(host.data(), dev, (size_t)n_max * sizeof(double), hipMemcpyDeviceToHost);
    (void)hipFree(dev);
    if (e != hipSuccess) return fail(MZS_E_HIP, std::string("selftest copy: ") + hipGetErrorString(e));
    int bad = 0;
    for (int d = 1; d <= n_max; ++d) bad += host[(size_t)d - 1] != 1.0 / (double)d;
    *n_bad = bad;
    return MZS_OK;
}

const char* mzs_version(void) { return "mzsearch 0.1 (gfx950)"; }

size_t mzs_weight_count(const mzs_dims* dims) {
    if (!dims) return 0;
    return mzs_blob_total(dims);
}

int mzs_create(const mzs_config* cfg, mzs_handle** out) {
    if (!cfg || !out) return fail(MZS_E_ARG, "null argument");
    const mzs_dims& d = cfg->dims;
    if (d.obs_dim < 1 || d.n_actions < 1 || d.state_dim < 1 || d.hidden_dim < 1 ||
        d.n_mid_layers < 0)
        return fail(MZS_E_ARG, "dims must be positive");
    if (d.model_kind != MZS_MODEL_MLP && d.model_kind != MZS_MODEL_LSTM && d.model_kind != MZS_MODEL_VISION)
        return fail(MZS_E_ARG, "model_kind must be MZS_MODEL_MLP, MZS_MODEL_LSTM or MZS_MODEL_VISION");
    if (d.model_kind == MZS_MODEL_VISION &&
        (d.obs_dim != MZS_VIS_SIDE || d.n_mid_layers < 1 || d.precision != MZS_PRECISION_F32 || d.hidden_dim > 2048))
        return fail(MZS_E_ARG, "vision_model: obs_dim = 98 (the image side, muzero_model.py:314), "
                               "number_of_hidden_layer >= 1, fp32, hidden_layer_dimensions <= 2048");
    if (d.model_kind == MZS_MODEL_VISION) {
        const mzs_vision_offsets vo = mzs_vision_layout(&d);
        if (vo.d_fc.in_w + (vo.p_v.in_w - vo.p_res.conv_a) > (size_t)kVisCache)
            return fail(MZS_E_ARG, "vision_model: convolution / batchnorm parameters exceed the LDS cache (kVisCache)");
    }
    if (d.precision != MZS_PRECISION_F32 && d.precision != MZS_PRECISION_BF16 && d.precision != MZS_PRECISION_BF16X3)
        return fail(MZS_E_ARG, "precision must be MZS_PRECISION_F32, MZS_PRECISION_BF16 or MZS_PRECISION_BF16X3");
    if (d.model_kind == MZS_MODEL_LSTM && (d.n_mid_layers < 1 || d.n_mid_layers > MZS_LSTM_MAX_LAYERS))
        return fail(MZS_E_ARG, "lstm_model: 1..4 LSTM layers (n_mid_layers)");
    if (d.n_actions > 64) return fail(MZS_E_ARG, "n_actions > 64 not supported (one lane per child)");
    if (d.state_dim > 128) return fail(MZS_E_ARG, "state_dim > 128 not supported (two logits per lane)");
    if (cfg->max_games < 1 || cfg->max_simulations < 0) return fail(MZS_E_ARG, "bad sizes");
    if (mzs_nodes_per_game(d.n_actions, cfg->max_simulations) > 65535)
        return fail(MZS_E_ARG, "tree exceeds 65535 nodes per game");
    if (cfg->max_simulations > 0 && !cfg->pbc_table) return fail(MZS_E_ARG, "pbc_table is null");
    if (cfg->engine != MZS_ENGINE_FUSED && cfg->engine != MZS_ENGINE_FUSED_GENERIC)
        return fail(MZS_E_ARG, "unknown engine");
    HIP_TRY(hipSetDevice(cfg->device));
    {
        /* set-up for mzs_train_grad_static: this GPU's address of the static scratch array */
        void* sp = nullptr;
        const int rc = scratch_for_device(cfg->device, &sp);
        if (rc) return rc;
    }
    mzs_handle* h = new (std::nothrow) mzs_handle();
    if (!h) return fail(MZS_E_ARG, "out of host memory");
    h->cfg = *cfg;
    h->cfg.pbc_table = nullptr;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess && prop.multiProcessorCount > 0)
            h->n_cus = prop.multiProcessorCount;
    }
    StoreBuilder sb;
    describe_model(d, &h->model, &sb);
    h->w_floats = sb.data.size();
    if (d.model_kind == MZS_MODEL_VISION) /* the Linear layers' quad store (mzs_vision.h) */
        h->w_floats = vis_lin_total(d.hidden_dim, d.n_mid_layers, d.state_dim, d.n_actions) +
                      vis_mm_total(d.hidden_dim, d.n_mid_layers, d.state_dim, d.n_actions); /* + the MFMA store */
    h->n_nodes_max = mzs_nodes_per_game(d.n_actions, cfg->max_simulations);
    h->lds = make_lds_layout(h->model, h->n_nodes_max, cfg->max_simulations, d.model_kind == MZS_MODEL_VISION ? 1 : 0);
    if (waves_per_block(h->lds) == 0) {
        delete h;
        return fail(MZS_E_ARG, "tree + activations of one game exceed 160 KiB of LDS");
    }
    const size_t G = (size_t)cfg->max_games, Smax = (size_t)cfg->max_simulations;
    const int A = d.n_actions;
    h->tree.stride = round_up(h->n_nodes_max, 32);
    h->tree.hid_stride = round_up(mzs_hidden_floats(&d), 32);
    h->tree.max_sims = cfg->max_simulations;
    int rc = MZS_OK;
    const size_t ts = (size_t)h->tree.stride;
    const size_t n_blob = mzs_blob_total(&d);
    if ((rc = dev_alloc(h, &h->w_dev, h->w_floats)) ||
        (rc = dev_alloc(h, &h->blob_dev, n_blob)) ||
        (rc = dev_alloc(h, &h->store_index_dev, n_blob)) ||
        (rc = dev_alloc(h, &h->pbc_dev, Smax + 2)) ||
        (rc = dev_alloc(h, &h->tree.visit, G * ts)) ||
        (rc = dev_alloc(h, &h->tree.vsum, G * ts)) ||
        (rc = dev_alloc(h, &h->tree.prior, G * ts)) ||
        (rc = dev_alloc(h, &h->tree.reward, G * ts)) ||
        (rc = dev_alloc(h, &h->tree.child_base, G * ts)) ||
        (rc = dev_alloc(h, &h->tree.sim_leaf, G * (Smax + 1))) ||
        (rc = dev_alloc(h, &h->tree.minmax, G * 2)) ||
        (rc = dev_alloc(h, &h->tree.hidden, G * (Smax + 1) * h->tree.hid_stride)) ||
        (rc = dev_alloc(h, &h->tree.status, G)) ||
        (rc = dev_alloc(h, &h->res.visits, G * A + G)) || /* root values right behind: one read-back */
        (rc = dev_alloc(h, &h->res.root_prior, G * A)) ||
        (rc = dev_alloc(h, &h->res.consumed, G)) ||
        (rc = dev_alloc(h, &h->res.path_len, G))) {
        mzs_destroy(h);
        return rc;
    }
    h->res.root_value = reinterpret_cast<float*>(h->res.visits + G * A);
    if (cfg->max_simulations > 0) {
        hipError_t e = hipMemcpy(h->pbc_dev, cfg->pbc_table, (Smax + 1) * sizeof(double),
                                 hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            mzs_destroy(h);
            return fail(MZS_E_HIP, std::string("pbc_table upload: ") + hipGetErrorString(e));
        }
    }
    if ((rc = dev_alloc(h, &h->overflow_dev, 64))) {
        mzs_destroy(h);
        return rc;
    }
    if (d.model_kind == MZS_MODEL_VISION &&
        (rc = dev_alloc(h, &h->root_policy_dev, (size_t)cfg->max_games * d.n_actions))) {
        mzs_destroy(h);
        return rc;
    }
    h->w_search_dev = h->w_dev;
    if (d.precision == MZS_PRECISION_BF16 && (rc = dev_alloc(h, &h->w_search_dev, h->w_floats))) {
        mzs_destroy(h);
        return rc;
    }
    h->model.w = h->w_search_dev;
    h->model.blob = h->blob_dev;
    h->stream_plan = make_stream_plan(d, h->model);
    if (h->stream_plan.n > 0 &&
        (rc = dev_alloc(h, &h->stream_dev, (size_t)h->stream_plan.n * kImageFloats))) {
        mzs_destroy(h);
        return rc;
    }
    *out = h;
    return MZS_OK;
}

void mzs_destroy(mzs_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->cfg.device);
    for (void* p : h->allocs) (void)hipFree(p);
    vis_train_destroy(h->vis_train);
    delete h;
}

int mzs_set_weights(mzs_handle* h, const float* blob, size_t n_floats) {
    if (!h || !blob) return fail(MZS_E_ARG, "null argument");
    const mzs_dims& d = h->cfg.dims;
    const mzs_blob_offsets o = mzs_blob_layout(&d);
    if (n_floats != mzs_blob_total(&d)) {
        char msg[128];
        std::snprintf(msg, sizeof msg, "weight blob has %zu floats, expected %zu", n_floats, mzs_blob_total(&d));
        return fail(MZS_E_ARG, msg);
    }
    HIP_TRY(hipSetDevice(h->cfg.device));
    std::vector<float> host(n_floats);
    HIP_TRY(hipMemcpy(host.data(), blob, n_floats * sizeof(float), hipMemcpyDefault));
    const float* B = host.data();
    const int H = d.hidden_dim, Sd = d.state_dim, A = d.n_actions, ob = d.obs_dim;
    if (d.model_kind == MZS_MODEL_VISION) {
        /* convolutions, batchnorms and biases are read from the blob in place; the Linear layers of the three
         * head MLPs from a store of their own, laid out for 16-byte loads (mzs_vision.h, vis_lin_offsets) */
        HIP_TRY(hipMemcpy(h->blob_dev, B, n_floats * sizeof(float), hipMemcpyHostToDevice));
        const size_t n_store = vis_lin_total(H, d.n_mid_layers, Sd, A) + vis_mm_total(H, d.n_mid_layers, Sd, A);
        const mzs_vision_offsets vo = mzs_vision_layout(&d);
        /* pass 0 fills the stores with the weights; pass 1 runs the same permutation over the blob INDICES
         * (exact in a float below 2^24), which is what refreshes the stores after a training step */
        std::vector<float> ids;
        if (!h->vis_store_src_dev) {
            if (n_floats >= (size_t)1 << 24) return fail(MZS_E_ARG, "vision blob too large for the store index");
            ids.resize(n_floats);
            for (size_t i = 0; i < n_floats; ++i) ids[i] = (float)(i + 1);
        }
        for (int pass = 0; pass < (ids.empty() ? 1 : 2); ++pass) {
        const float* B = pass == 0 ? host.data() : ids.data();
        /* the quad store of mzs_vision.h (vis_lin_offsets) */
        std::vector<float> t(n_store, 0.0f);
        const auto quads = [&](size_t dst, size_t src, int n_out, int n_in) {
            const int Q = n_in / 4;
            for (int j = 0; j < n_out; ++j)
                for (int k = 0; k < n_in; ++k) {
                    const float v = B[src + (size_t)j * n_in + k];
                    if (k < 4 * Q)
                        t[dst + ((size_t)(k / 4) * n_out + j) * 4 + k % 4] = v;
                    else
                        t[dst + (size_t)4 * Q * n_out + (size_t)(k - 4 * Q) * n_out + j] = v;
                }
        };
        const mzs_vis_mlp* heads[3] = {&vo.d_fc, &vo.p_v, &vo.p_p};
        for (int hd = 0; hd < 3; ++hd) {
            const mzs_vis_mlp* fc = heads[hd];
            const VisLin lw = vis_lin_offsets(hd, H, d.n_mid_layers, Sd, A);
            quads(lw.in_w, fc->in_w, H, MZS_VIS_HIDDEN);
            if (d.n_mid_layers > 0) quads(lw.mid_w, fc->mid_w, H, H);
            quads(lw.out_w, fc->out_w, fc->n_out, H);
            /* the same matrices for the matrix core (mzs_vision_mm.h): [Kpad / 4][NPAD][4], zero padded */
            const auto mm = [&](size_t dst, size_t src, int n_out, int n_in) {
                const size_t NP = vis_mm_r(n_out, 64);
                for (int j = 0; j < n_out; ++j)
                    for (int k = 0; k < n_in; ++k)
                        t[dst + ((size_t)(k / 4) * NP + j) * 4 + k % 4] = B[src + (size_t)j * n_in + k];
            };
            const VisLin mw = vis_mm_offsets(hd, H, d.n_mid_layers, Sd, A);
            mm(mw.in_w, fc->in_w, H, MZS_VIS_HIDDEN);
            if (d.n_mid_layers > 0) mm(mw.mid_w, fc->mid_w, H, H);
            mm(mw.out_w, fc->out_w, fc->n_out, H);
        }
        if (pass == 0) {
            HIP_TRY(hipMemcpy(h->w_dev, t.data(), t.size() * sizeof(float), hipMemcpyHostToDevice));
        } else {
            std::vector<int> src(n_store);
            for (size_t i = 0; i < n_store; ++i) src[i] = (int)t[i] - 1;
            int rc = dev_alloc(h, &h->vis_store_src_dev, n_store);
            if (rc) return rc;
            HIP_TRY(hipMemcpy(h->vis_store_src_dev, src.data(), n_store * sizeof(int), hipMemcpyHostToDevice));
        }
        }
        h->have_weights = true;
        return MZS_OK;
    }
    StoreBuilder sb;
    Model m;
    describe_model(d, &m, &sb);
    if (d.model_kind == MZS_MODEL_LSTM) {
        const mzs_lstm_offsets lo = mzs_lstm_layout(&d);
        /* W_hh and the forget-gate rows are not in the store (a search never reads them): index -1 */
        sb.index.assign(n_floats, -1);
        sb.fill_at(m.r_in, B, lo.r_w, lo.r_b, Sd, ob, ob, 0);
        fill_lstm_head(&sb, m.l_rew, B, lo.rew, H, d.n_mid_layers);
        fill_lstm_head(&sb, m.l_st, B, lo.st, H, d.n_mid_layers);
        fill_lstm_head(&sb, m.l_pol, B, lo.pol, H, d.n_mid_layers);
        fill_lstm_head(&sb, m.l_val, B, lo.val, H, d.n_mid_layers);
        HIP_TRY(hipMemcpy(h->blob_dev, B, n_floats * sizeof(float), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(h->store_index_dev, sb.index.data(), n_floats * sizeof(int), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(h->w_dev, sb.data.data(), sb.data.size() * sizeof(float), hipMemcpyHostToDevice));
        if (h->w_search_dev != h->w_dev) {
            const std::vector<float> r = sb.rounded_bf16();
            HIP_TRY(hipMemcpy(h->w_search_dev, r.data(), r.size() * sizeof(float), hipMemcpyHostToDevice));
        }
        h->have_weights = true;
        return MZS_OK;
    }
    sb.index.assign(n_floats, 0);
    sb.fill_at(m.r_in, B, o.r_in_w, o.r_in_b, H, ob, ob, 0);
    sb.fill_at(m.r_out, B, o.r_out_w, o.r_out_b, Sd, H, H, 0);
    sb.fill_at(m.d_in, B, o.d_in_w, o.d_in_b, H, Sd + A, Sd + A, 0);
    sb.fill_at(m.d_heads, B, o.d_rew_w, o.d_rew_b, Sd, H, H, 0);
    sb.fill_at(m.d_heads, B, o.d_st_w, o.d_st_b, Sd, H, H, m.seg_state);
    sb.fill_at(m.p_in, B, o.p_in_w, o.p_in_b, H, Sd, Sd, 0);
    sb.fill_at(m.p_heads, B, o.p_pol_w, o.p_pol_b, A, H, H, 0);
    sb.fill_at(m.p_heads, B, o.p_val_w, o.p_val_b, Sd, H, H, m.seg_value);
    if (d.n_mid_layers > 0) {
        sb.fill_at(m.r_mid, B, o.r_mid_w, o.r_mid_b, H, H, H, 0);
        sb.fill_at(m.d_mid, B, o.d_mid_w, o.d_mid_b, H, H, H, 0);
        sb.fill_at(m.p_mid, B, o.p_mid_w, o.p_mid_b, H, H, H, 0);
    }
    HIP_TRY(hipMemcpy(h->blob_dev, B, n_floats * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->store_index_dev, sb.index.data(), n_floats * sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->w_dev, sb.data.data(), sb.data.size() * sizeof(float),
                      hipMemcpyHostToDevice));
    if (h->w_search_dev != h->w_dev) {
        const std::vector<float> r = sb.rounded_bf16();
        HIP_TRY(hipMemcpy(h->w_search_dev, r.data(), r.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    h->have_weights = true;
    return MZS_OK;
}

}  // extern "C" (reopened below)

/* the search launch of mzs_search / mzs_search_streams: noise_avail draws of every row hold data;
 * redo = only the games whose status a previous launch set */
static int search_launch(mzs_handle* h, int32_t n_games, int32_t S, int32_t train, const float* obs_dev,
                         const double* dirichlet_dev, const double* noise_dev, int64_t noise_stride,
                         int64_t noise_avail, int redo, bool build_stream, hipStream_t st) {
    SearchParams p;
    fill_search_params(h, &p, n_games, S, train, dirichlet_dev, noise_dev, noise_stride);
    p.obs = obs_dev;
    p.noise_avail = noise_avail;
    p.overflow = h->overflow_dev;
    p.redo = redo;
    const SearchPlan plan = plan_search(h, S, n_games);
    if (plan.kind == kPlanGeneric && h->cfg.dims.model_kind == MZS_MODEL_VISION) {
        /* two launches: the roots (representation + prediction, one wave per SIMD for its LDS), then the
         * simulations on a small LDS slice.  A redo pass finds the roots where the first pass left them. */
        if (!redo) {
            InferParams ip;
            std::memset(&ip, 0, sizeof ip);
            ip.m = h->model;
            ip.lds = make_lds_layout(h->model, 1, 0);
            ip.batch = n_games;
            ip.in = obs_dev;
            ip.hidden_out = h->tree.hidden;
            ip.hidden_stride = (long long)(h->tree.max_sims + 1) * h->tree.hid_stride;
            ip.policy = h->root_policy_dev;
            const int rc = launch(k_inference<kInferInitial, true>, ip, n_games, ip.lds, st);
            if (rc) return rc;
        }
        p.t_root_policy = h->root_policy_dev;
        if (vis4_ok(h, n_games)) {
            /* four games per block, the head MLPs on the matrix core (k_search_vis4): the per-wave slices, then
             * the block's parameter cache and the head phase's buffers.  As many blocks per CU as the LDS seats
             * (each block is one wave per SIMD), at most four, and no more than the batch can use: fewer waves
             * per SIMD leave each of them more registers. */
            const mzs_dims& d = h->cfg.dims;
            p.lds = make_lds_layout(h->model, p.n_nodes, S, 2);
            const size_t bytes = (size_t)4 * p.lds.bytes +
                                 (kVisCache + vis_mm_lds_floats(d.hidden_dim, d.state_dim, d.n_actions)) * sizeof(float);
            const int blocks = (n_games + 3) / 4;
            int occ = (int)((size_t)kMaxLdsPerBlock / bytes);
            occ = occ > 4 ? 4 : occ;
            while (occ > 2 && blocks <= (occ - 1) * h->n_cus) --occ;
            if (const char* ev = std::getenv("MZS_VIS4_OCC")) { /* measurement only (tools/bench_vision.py) */
                const int want = std::atoi(ev);
                if (want >= 2 && want <= occ) occ = want;
                if (want == 1) occ = 0; /* 1: fall back to k_search_fused */
            }
            if (occ >= 2) {
                auto kern = occ == 4 ? k_search_vis4<4> : (occ == 3 ? k_search_vis4<3> : k_search_vis4<2>);
                if (bytes > 64 * 1024)
                    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
                hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), bytes, st, p);
                HIP_TRY(hipGetLastError());
                return MZS_OK;
            }
        }
        p.lds = make_lds_layout(h->model, p.n_nodes, S, 1);
        /* up to two waves per SIMD's worth of games: the build with 256 registers; more: four per SIMD */
        return n_games <= 8 * h->n_cus ? launch(k_search_fused<false, true, 2>, p, n_games, p.lds, st)
                                       : launch(k_search_fused<false, true, 4>, p, n_games, p.lds, st);
    }
    if (plan.kind == kPlanGeneric) return launch(k_search_fused<false>, p, n_games, p.lds, st);
    if (plan.kind == kPlanStream) {
        if (build_stream) {
            if (h->cfg.dims.precision == MZS_PRECISION_BF16X3) /* hi + lo images from the fp32 store */
                hipLaunchKernelGGL(k_build_stream_x, dim3(h->stream_plan.n), dim3(256), 0, st, h->w_search_dev,
                                   h->stream_plan, h->stream_dev);
            else if (h->cfg.dims.precision == MZS_PRECISION_BF16)
                hipLaunchKernelGGL(k_build_stream_b, dim3(h->stream_plan.n), dim3(256), 0, st, h->w_search_dev,
                                   h->stream_plan, h->stream_dev);
            else
                hipLaunchKernelGGL(k_build_stream, dim3(h->stream_plan.n), dim3(256), 0, st, h->w_search_dev,
                                   h->stream_plan, h->stream_dev);
            HIP_TRY(hipGetLastError());
        }
        p.img = h->stream_dev;
        p.n_img = h->stream_plan.n;
        p.first_img = h->stream_plan.first;
    }
    return launch_quad(plan.kernel, p, plan.ql, plan.wpb, st);
}

static int fill_launch(mzs_handle* h, int32_t n_games, int32_t train, const int64_t* seeds_dev, double alpha, double lo,
                       double hi, double* dirichlet_dev, double* noise_dev, int64_t noise_stride, int64_t n_draws,
                       const int* redo_status, hipStream_t st) {
    FillParams p;
    p.seeds = reinterpret_cast<const long long*>(seeds_dev);
    p.redo_status = redo_status;
    p.overflow_flag = h->overflow_dev;
    p.n_games = n_games;
    p.A = h->cfg.dims.n_actions;
    p.n_uniform = (int)n_draws;
    p.train = train;
    p.stride = noise_stride;
    p.alpha = alpha;
    p.lo = lo;
    p.range = hi - lo;
    p.dirichlet = dirichlet_dev;
    p.noise = noise_dev;
    hipLaunchKernelGGL(k_fill_streams, dim3((n_games + kFillGames - 1) / kFillGames), dim3(64 * kFillGames), 0, st, p);
    HIP_TRY(hipGetLastError());
    return MZS_OK;
}

static int check_fill_args(mzs_handle* h, int32_t n_games, int32_t num_simulations, int32_t train,
                           const int64_t* seeds_dev, double alpha, const double* dirichlet_dev,
                           const double* noise_dev, int64_t noise_stride) {
    if (!h) return fail(MZS_E_ARG, "null handle");
    if (n_games < 0 || num_simulations < 0) return fail(MZS_E_ARG, "bad sizes");
    const int64_t n_u = mzs_max_draws(h->cfg.dims.n_actions, num_simulations);
    if (!seeds_dev || (n_u > 0 && (!noise_dev || noise_stride < n_u)))
        return fail(MZS_E_ARG, "seeds / noise buffer missing or noise_stride < mzs_max_draws(A, S)");
    const int do_dir = train && num_simulations > 0;
    if (do_dir && !dirichlet_dev) return fail(MZS_E_ARG, "train=1 needs dirichlet_dev");
    if (do_dir && !(alpha > 0.0 && alpha <= 1.0))
        return fail(MZS_E_ARG, "device Dirichlet implements NumPy's legacy gamma sampler for 0 < alpha <= 1 "
                               "only (draw the streams on the host for larger alpha)");
    if (h->cfg.dims.n_actions > 64) return fail(MZS_E_ARG, "device Dirichlet holds one gamma per lane: n_actions <= 64");
    if (n_u >= (1ll << 31)) return fail(MZS_E_ARG, "stream too long");
    return MZS_OK;
}

extern "C" {

int mzs_search(mzs_handle* h, int32_t n_games, int32_t S, int32_t train, const float* obs_dev,
               const double* dirichlet_dev, const double* noise_dev, int64_t noise_stride,
               void* stream) {
    int rc = check_search_args(h, n_games, S, train, dirichlet_dev, noise_dev, noise_stride);
    if (rc) return rc;
    if (!h->have_weights) return fail(MZS_E_STATE, "mzs_set_weights has not been called");
    if (!obs_dev) return fail(MZS_E_ARG, "obs_dev is null");
    if (n_games == 0) return MZS_OK;
    HIP_TRY(hipSetDevice(h->cfg.device));
    return search_launch(h, n_games, S, train, obs_dev, dirichlet_dev, noise_dev, noise_stride, noise_stride, 0, true,
                         static_cast<hipStream_t>(stream));
}

/* One move's searches with their random streams made on the device, lazily: the Dirichlet rows and
 * the first prefix_draws tie-break draws of every game, the search, and then, for the games whose
 * search asked for more than the prefix (status), the whole stream and the search again.  Two
 * launches more than fill + search, both of which leave at once when no game asked.  Results are
 * those of mzs_fill_streams + mzs_search, bit for bit. */
int mzs_search_streams(mzs_handle* h, int32_t n_games, int32_t S, int32_t train, const float* obs_dev,
                       const int64_t* seeds_dev, double alpha, double lo, double hi, int32_t prefix_draws,
                       double* dirichlet_dev, double* noise_dev, int64_t noise_stride, void* stream) {
    int rc = check_fill_args(h, n_games, S, train, seeds_dev, alpha, dirichlet_dev, noise_dev, noise_stride);
    if (rc) return rc;
    if ((rc = check_search_args(h, n_games, S, train, dirichlet_dev, noise_dev, noise_stride))) return rc;
    if (!h->have_weights) return fail(MZS_E_STATE, "mzs_set_weights has not been called");
    if (!obs_dev) return fail(MZS_E_ARG, "obs_dev is null");
    if (n_games == 0) return MZS_OK;
    HIP_TRY(hipSetDevice(h->cfg.device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int A = h->cfg.dims.n_actions;
    const int64_t n_u = mzs_max_draws(A, S);
    /* default prefix: nine levels per simulation, about twice what a search of these sizes consumes
     * (with the Dirichlet's words: three refreshes of the 624-word generator state at A = 2, S = 50) */
    int64_t prefix = prefix_draws > 0 ? prefix_draws : (int64_t)9 * A * S;
    prefix = prefix < n_u ? prefix : n_u;
    const int do_dir = train && S > 0;
    if ((rc = fill_launch(h, n_games, do_dir, seeds_dev, alpha, lo, hi, dirichlet_dev, noise_dev, noise_stride, prefix,
                          nullptr, st)))
        return rc;
    if ((rc = search_launch(h, n_games, S, train, obs_dev, dirichlet_dev, noise_dev, noise_stride, prefix, 0, true, st)))
        return rc;
    if (prefix == n_u) return MZS_OK;
    if ((rc = fill_launch(h, n_games, do_dir, seeds_dev, alpha, lo, hi, dirichlet_dev, noise_dev, noise_stride, n_u,
                          h->tree.status, st)))
        return rc;
    return search_launch(h, n_games, S, train, obs_dev, dirichlet_dev, noise_dev, noise_stride, n_u, 1, false, st);
}

int mzs_search_kernel_name(mzs_handle* h, int32_t num_simulations, char* name_out, size_t name_size) {
    if (!h || !name_out || name_size == 0) return fail(MZS_E_ARG, "null argument");
    if (num_simulations < 0 || num_simulations > h->cfg.max_simulations)
        return fail(MZS_E_ARG, "num_simulations exceeds max_simulations of the handle");
    const SearchPlan plan = plan_search(h, num_simulations, h->cfg.max_games);
    std::snprintf(name_out, name_size, "%s", plan.name);
    return MZS_OK;
}

int mzs_search_injected(mzs_handle* h, int32_t n_games, int32_t S, int32_t train,
                        const float* root_policy_dev, const float* sim_reward_dev,
                        const float* sim_value_dev, const float* sim_prior_dev,
                        const double* dirichlet_dev, const double* noise_dev,
                        int64_t noise_stride, void* stream) {
    int rc = check_search_args(h, n_games, S, train, dirichlet_dev, noise_dev, noise_stride);
    if (rc) return rc;
    if (!root_policy_dev || (S > 0 && (!sim_reward_dev || !sim_value_dev || !sim_prior_dev)))
        return fail(MZS_E_ARG, "injected model tables are null");
    if (n_games == 0) return MZS_OK;
    HIP_TRY(hipSetDevice(h->cfg.device));
    SearchParams p;
    fill_search_params(h, &p, n_games, S, train, dirichlet_dev, noise_dev, noise_stride);
    p.t_root_policy = root_policy_dev;
    p.t_reward = sim_reward_dev;
    p.t_value = sim_value_dev;
    p.t_prior = sim_prior_dev;
    if (h->cfg.engine == MZS_ENGINE_FUSED && mzs_max_draws(h->cfg.dims.n_actions, S) < (1ll << 31)) {
        if (quad_kernel_t qk = pick_quad_injected(h->cfg.dims.n_actions)) {
            const QuadLayout ql = make_quad_layout(p.m.A, p.n_nodes, S, false);
            const int wpb = quad_waves_per_block(ql);
            if (wpb > 0) return launch_quad(qk, p, ql, wpb, static_cast<hipStream_t>(stream));
        }
    }
    return launch(k_search_fused<true>, p, n_games, p.lds, static_cast<hipStream_t>(stream));
}

int mzs_set_result_sink(mzs_handle* h, int32_t* visits, float* root_value) {
    if (!h) return fail(MZS_E_ARG, "null handle");
    if ((visits == nullptr) != (root_value == nullptr))
        return fail(MZS_E_ARG, "pass both sink pointers or neither");
    h->res.visits_sink = visits;
    h->res.root_value_sink = root_value;
    return MZS_OK;
}

int mzs_results(mzs_handle* h, int32_t n_games, int32_t* visits_dev, float* root_value_dev,
                double* root_prior_dev, int64_t* consumed_dev, int64_t* path_len_dev,
                void* stream) {
    if (!h) return fail(MZS_E_ARG, "null handle");
    if (n_games < 0 || n_games > h->cfg.max_games) return fail(MZS_E_ARG, "bad n_games");
    HIP_TRY(hipSetDevice(h->cfg.device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t G = (size_t)n_games, A = (size_t)h->cfg.dims.n_actions;
    if (visits_dev)
        HIP_TRY(hipMemcpyAsync(visits_dev, h->res.visits, G * A * 4, hipMemcpyDeviceToDevice, st));
    if (root_value_dev)
        HIP_TRY(hipMemcpyAsync(root_value_dev, h->res.root_value, G * 4, hipMemcpyDeviceToDevice, st));
    if (root_prior_dev)
        HIP_TRY(hipMemcpyAsync(root_prior_dev, h->res.root_prior, G * A * 8, hipMemcpyDeviceToDevice, st));
    if (consumed_dev)
        HIP_TRY(hipMemcpyAsync(consumed_dev, h->res.consumed, G * 8, hipMemcpyDeviceToDevice, st));
    if (path_len_dev)
        HIP_TRY(hipMemcpyAsync(path_len_dev, h->res.path_len, G * 8, hipMemcpyDeviceToDevice, st));
    return MZS_OK;
}

void* mzs_tree_array(mzs_handle* h, int32_t which) {
    if (!h) return nullptr;
    switch (which) {
        case MZS_TREE_VISIT: return h->tree.visit;
        case MZS_TREE_VALUE_SUM: return h->tree.vsum;
        case MZS_TREE_PRIOR: return h->tree.prior;
        case MZS_TREE_REWARD: return h->tree.reward;
        case MZS_TREE_CHILD_BASE: return h->tree.child_base;
        case MZS_TREE_SIM_LEAF: return h->tree.sim_leaf;
        case MZS_TREE_MINMAX: return h->tree.minmax;
        case MZS_TREE_HIDDEN: return h->tree.hidden;
        case MZS_TREE_STATUS: return h->tree.status;
        case MZS_TREE_ROOT_VISITS: return h->res.visits;
        case MZS_TREE_ROOT_VALUE: return h->res.root_value;
        case MZS_TREE_ROOT_PRIOR: return h->res.root_prior;
        case MZS_TREE_CONSUMED: return h->res.consumed;
        case MZS_TREE_PATH_LEN: return h->res.path_len;
        default: return nullptr;
    }
}

int32_t mzs_tree_stride(const mzs_handle* h) { return h ? h->tree.stride : 0; }
int32_t mzs_hidden_stride(const mzs_handle* h) { return h ? h->tree.hid_stride : 0; }

static int run_inference(mzs_handle* h, int mode, int32_t batch, const float* in,
                         const int32_t* action, float* reward, float* hidden_out, float* policy,
                         float* value, void* stream) {
    if (!h) return fail(MZS_E_ARG, "null handle");
    if (!h->have_weights) return fail(MZS_E_STATE, "mzs_set_weights has not been called");
    if (batch < 0 || !in) return fail(MZS_E_ARG, "bad batch or null input");
    if (mode != kInferPrediction && !hidden_out) return fail(MZS_E_ARG, "hidden output is null");
    if (mode == kInferRecurrent && (!action || !reward)) return fail(MZS_E_ARG, "null action/reward");
    if (batch == 0) return MZS_OK;
    HIP_TRY(hipSetDevice(h->cfg.device));
    InferParams p;
    std::memset(&p, 0, sizeof p);
    p.m = h->model;
    p.lds = make_lds_layout(h->model, 1, 0);
    p.batch = batch;
    p.in = in;
    p.action = action;
    p.reward = reward;
    p.hidden_out = hidden_out;
    p.policy = policy;
    p.value = value;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (h->cfg.dims.model_kind == MZS_MODEL_VISION) {
        if (mode == kInferInitial) return launch(k_inference<kInferInitial, true>, p, batch, p.lds, st);
        if (mode == kInferRecurrent) return launch(k_inference<kInferRecurrent, true>, p, batch, p.lds, st);
        return launch(k_inference<kInferPrediction, true>, p, batch, p.lds, st);
    }
    if (mode == kInferInitial) return launch(k_inference<kInferInitial>, p, batch, p.lds, st);
    if (mode == kInferRecurrent) return launch(k_inference<kInferRecurrent>, p, batch, p.lds, st);
    return launch(k_inference<kInferPrediction>, p, batch, p.lds, st);
}

int mzs_initial_inference(mzs_handle* h, int32_t batch, const float* obs_dev, float* hidden_dev,
                          float* policy_dev, float* value_dev, void* stream) {
    return run_inference(h, kInferInitial, batch, obs_dev, nullptr, nullptr, hidden_dev, policy_dev,
                         value_dev, stream);
}

int mzs_recurrent_inference(mzs_handle* h, int32_t batch, const float* hidden_dev,
                            const int32_t* action_dev, float* reward_dev, float* next_hidden_dev,
                            float* policy_dev, float* value_dev, void* stream) {
    return run_inference(h, kInferRecurrent, batch, hidden_dev, action_dev, reward_dev,
                         next_hidden_dev, policy_dev, value_dev, stream);
}

int mzs_prediction_inference(mzs_handle* h, int32_t batch, const float* hidden_dev,
                             float* policy_dev, float* value_dev, void* stream) {
    if (!policy_dev && !value_dev) return fail(MZS_E_ARG, "no output requested");
    return run_inference(h, kInferPrediction, batch, hidden_dev, nullptr, nullptr, nullptr,
                         policy_dev, value_dev, stream);
}

/* ------------------------------------------------------------------ training step */

static TapeLayout make_tape_layout(const Model& m, int K) {
    TapeLayout t;
    std::memset(&t, 0, sizeof t);
    const int L = m.L;
    t.HP = round_up(m.H, 64);
    t.SP = round_up(m.Sd, 64);
    t.AP = round_up(m.A, 64);
    t.XP = round_up(m.Sd + m.A, 64);
    t.OP = round_up(m.obs_dim, 64);
    t.OPH = m.p_heads.n_pad;
    t.ODH = m.d_heads.n_pad;
    t.L = L;
    t.K = K;
    int off = 0;
    t.r_x = off, off += t.OP;
    t.r_T = off, off += (L + 1) * t.HP;
    t.r_din = off, off += t.HP;
    t.r_dmid = off, off += L * t.HP;
    t.r_dout = off, off += t.SP;
    t.p_base = off;
    int r = 0;
    t.p_h = r, r += t.SP;
    t.p_T = r, r += (L + 1) * t.HP;
    t.p_din = r, r += t.HP;
    t.p_dmid = r, r += L * t.HP;
    t.p_dpol = r, r += t.AP;
    t.p_dval = r, r += t.SP;
    t.p_out = r, r += t.OPH;
    t.p_stride = r;
    off += K * t.p_stride + t.SP; /* + one spare hidden-state row past the last record */
    t.d_base = off;
    r = 0;
    t.d_x = r, r += t.XP;
    t.d_T = r, r += (L + 1) * t.HP;
    t.d_din = r, r += t.HP;
    t.d_dmid = r, r += L * t.HP;
    t.d_drew = r, r += t.SP;
    t.d_dst = r, r += t.SP;
    t.d_out = r, r += t.ODH;
    t.d_stride = r;
    off += (K > 1 ? K - 1 : 0) * t.d_stride;
    t.copy_n = off;
    t.s_zs = off, off += m.r_out.n_pad;
    t.s_gT = off, off += t.HP;
    t.s_gT2 = off, off += t.HP;
    t.s_gh = off, off += t.XP;
    t.s_ghn = off, off += t.XP;
    t.s_scl = off, off += 4 * K;
    t.total = off;
    return t;
}

static void fill_tensor_descs(const mzs_dims& d, const TapeLayout& t, int K, TensorDesc* td) {
    const mzs_blob_offsets o = mzs_blob_layout(&d);
    const int H = d.hidden_dim, S = d.state_dim, A = d.n_actions, L = d.n_mid_layers, HP = t.HP;
    const auto set = [&](int i, size_t w, size_t b, int n_out, int n_in, int x, int dl, int on, int os,
                         int in, int l2) {
        td[i].w_off = (int)w, td[i].b_off = (int)b, td[i].n_out = n_out, td[i].n_in = n_in;
        td[i].x_off = x, td[i].d_off = dl, td[i].outer_n = on, td[i].outer_stride = os;
        td[i].inner_n = in, td[i].inner_sx = HP, td[i].inner_sd = HP, td[i].l2 = l2;
    };
    const int Km = K > 1 ? K - 1 : 0, pb = t.p_base, db = t.d_base;
    /* l2 (neural_network_mlp_model.py:280-285): zip() over the three parameter lists stops at the
     * representation's length, so the next-state head and the value head are left out */
    set(0, o.r_in_w, o.r_in_b, H, d.obs_dim, t.r_x, t.r_din, 1, 0, 1, 1);
    set(1, o.r_mid_w, o.r_mid_b, L > 0 ? H : 0, H, t.r_T, t.r_dmid, 1, 0, L, 1);
    set(2, o.r_out_w, o.r_out_b, S, H, t.r_T + L * HP, t.r_dout, 1, 0, 1, 1);
    set(3, o.d_in_w, o.d_in_b, H, S + A, db + t.d_x, db + t.d_din, Km, t.d_stride, 1, 1);
    set(4, o.d_mid_w, o.d_mid_b, L > 0 ? H : 0, H, db + t.d_T, db + t.d_dmid, Km, t.d_stride, L, 1);
    set(5, o.d_rew_w, o.d_rew_b, S, H, db + t.d_T + L * HP, db + t.d_drew, Km, t.d_stride, 1, 1);
    set(6, o.d_st_w, o.d_st_b, S, H, db + t.d_T + L * HP, db + t.d_dst, Km, t.d_stride, 1, 0);
    set(7, o.p_in_w, o.p_in_b, H, S, pb + t.p_h, pb + t.p_din, K, t.p_stride, 1, 1);
    set(8, o.p_mid_w, o.p_mid_b, L > 0 ? H : 0, H, pb + t.p_T, pb + t.p_dmid, K, t.p_stride, L, 1);
    set(9, o.p_pol_w, o.p_pol_b, A, H, pb + t.p_T + L * HP, pb + t.p_dpol, K, t.p_stride, 1, 1);
    set(10, o.p_val_w, o.p_val_b, S, H, pb + t.p_T + L * HP, pb + t.p_dval, K, t.p_stride, 1, 0);
}

/* 'lstm_model': tape layout (rows of max_batch capacity) and the gradient descriptors of every
 * parameter tensor (mzs_train_lstm.h).  Host vectors are uploaded once by mzs_train_create. */
struct LstmTrainPlan {
    LstmTrainParams p;
    std::vector<LstmGradDesc> desc;
    std::vector<int> param_desc;
    size_t tape_floats;
};

static LstmTrainPlan make_lstm_train_plan(const mzs_dims& d, int K, int Bcap) {
    LstmTrainPlan pl;
    LstmTrainParams& p = pl.p;
    std::memset(&p, 0, sizeof p);
    const mzs_lstm_offsets lo = mzs_lstm_layout(&d);
    const int H = d.hidden_dim, Sd = d.state_dim, A = d.n_actions, L = d.n_mid_layers, ob = d.obs_dim;
    p.Bcap = Bcap, p.K = K, p.L = L, p.H = H, p.Sd = Sd, p.A = A, p.obs_dim = ob;
    p.n_params = (int)lo.total;
    p.r_w = (int)lo.r_w, p.r_b = (int)lo.r_b;
    long long off = 0;
    const auto take = [&](long long n) {
        const long long at = off;
        off += (n + 63) / 64 * 64;
        return at;
    };
    p.hstate = take((long long)K * Bcap * Sd);
    p.gh = take((long long)K * Bcap * Sd);
    const mzs_lstm_head_offsets* src[4] = {&lo.rew, &lo.st, &lo.pol, &lo.val};
    for (int hd = 0; hd < 4; ++hd) {
        LstmTrainHead& t = p.head[hd];
        const mzs_lstm_head_offsets& o = *src[hd];
        const int n = o.out_dim;
        t.lin_w = (int)o.lin_w, t.lin_b = (int)o.lin_b, t.in_dim = o.in_dim, t.n = n;
        t.wt_lin = take((long long)o.in_dim * H);
        int r = 0;
        const auto field = [&](int rows_of) {
            const int at = r;
            r += (Bcap * rows_of + 63) / 64 * 64;
            return at;
        };
        t.u = field(H), t.du = field(H), t.dx = field(Sd), t.dout = field(n);
        for (int l = 0; l < L; ++l) {
            t.w_ih[l] = (int)o.w_ih[l], t.w_hh[l] = (int)o.w_hh[l];
            t.b_ih[l] = (int)o.b_ih[l], t.b_hh[l] = (int)o.b_hh[l];
            t.wt_ih[l] = take((long long)(l == 0 ? H : n) * 4 * n);
            t.gates[l] = field(4 * n), t.c[l] = field(n), t.h[l] = field(n), t.dz[l] = field(4 * n),
            t.dmid[l] = field(n), t.ca[l] = field(n);
        }
        t.stride = r;
        const int n_calls = hd < 2 ? (K > 1 ? K - 1 : 0) : K;
        t.base = take((long long)n_calls * r);
    }
    pl.tape_floats = (size_t)off;

    pl.param_desc.assign(lo.total, -1);
    const long long hs = (long long)Bcap * Sd;
    const auto add = [&](LstmGradDesc g) {
        const int di = (int)pl.desc.size();
        pl.desc.push_back(g);
        for (int r = 0; r < g.n_rows; ++r)
            for (int c = 0; c < (g.n_cols ? g.n_cols : 1); ++c) pl.param_desc[(size_t)g.w_off + (size_t)r * g.ld + c] = di;
    };
    const auto weight = [&](int w_off, int ld, int n_rows, int n_cols, long long d_base, long long d_stride, int d_ld,
                            long long x_base, long long x_stride, int x_ld, int n_calls, int t_shift, int x_src,
                            int l2) {
        LstmGradDesc g;
        g.w_off = w_off, g.ld = ld, g.n_rows = n_rows, g.n_cols = n_cols;
        g.d_base = d_base, g.d_stride = d_stride, g.x_base = x_base, g.x_stride = x_stride;
        g.d_ld = d_ld, g.x_ld = x_ld, g.n_calls = n_calls, g.t_shift = t_shift, g.x_src = x_src, g.l2 = l2;
        add(g);
    };
    const auto bias = [&](int b_off, int n_rows, long long d_base, long long d_stride, int d_ld, int n_calls, int l2) {
        weight(b_off, 1, n_rows, 0, d_base, d_stride, d_ld, 0, 0, 0, n_calls, 0, 0, l2);
    };
    /* representation: delta = total gradient of h_0, input = the observation rows; in the l2 term */
    weight(p.r_w, ob, Sd, ob, p.gh, 0, Sd, 0, 0, ob, 1, 0, 1, 1);
    bias(p.r_b, Sd, p.gh, 0, Sd, 1, 1);
    for (int hd = 0; hd < 4; ++hd) {
        const LstmTrainHead& t = p.head[hd];
        const int n = t.n, nc = hd < 2 ? (K > 1 ? K - 1 : 0) : K;
        /* l2 (neural_network_mlp_model.py:280-285): the zip() over the three parameter lists stops after
         * the representation's two tensors: reward.0 and policy.0 are the only head tensors in it */
        const int l2 = hd == kHeadRew || hd == kHeadPol;
        weight(t.lin_w, t.in_dim, H, Sd, t.base + t.du, t.stride, H, p.hstate, hs, Sd, nc, 0, 0, l2);
        if (hd < 2) weight(t.lin_w + Sd, t.in_dim, H, A, t.base + t.du, t.stride, H, 0, 0, A, nc, 0, 2, l2);
        bias(t.lin_b, H, t.base + t.du, t.stride, H, nc, l2);
        for (int l = 0; l < L; ++l) {
            const int in_l = l == 0 ? H : n;
            const long long dz = t.base + t.dz[l];
            const long long x_in = t.base + (l == 0 ? t.u : t.h[l - 1]);
            weight(t.w_ih[l], in_l, 4 * n, in_l, dz, t.stride, 4 * n, x_in, t.stride, in_l, nc, 0, 0, 0);
            /* W_hh: input = the layer's own h of the previous row; the next-state head's last layer keeps
             * its h rows in the hidden-state array (h_{k+1}) */
            if (hd == kHeadSt && l == L - 1)
                weight(t.w_hh[l], n, 4 * n, n, dz, t.stride, 4 * n, p.hstate + hs, hs, Sd, nc, 1, 0, 0);
            else
                weight(t.w_hh[l], n, 4 * n, n, dz, t.stride, 4 * n, t.base + t.h[l], t.stride, n, nc, 1, 0, 0);
            bias(t.b_ih[l], 4 * n, dz, t.stride, 4 * n, nc, 0);
            bias(t.b_hh[l], 4 * n, dz, t.stride, 4 * n, nc, 0);
        }
    }
    p.n_desc = (int)pl.desc.size();
    return pl;
}

static int lstm_train_create(mzs_handle* h, int32_t max_batch, int32_t unroll_steps) {
    const mzs_dims& d = h->cfg.dims;
    LstmTrainPlan pl = make_lstm_train_plan(d, unroll_steps, max_batch);
    for (size_t i = 0; i < pl.param_desc.size(); ++i)
        if (pl.param_desc[i] < 0) return fail(MZS_E_STATE, "lstm_model training: a parameter has no gradient descriptor");
    const size_t n = mzs_blob_total(&d);
    LstmGradDesc* desc_dev = nullptr;
    int* param_desc_dev = nullptr;
    int rc;
    if ((rc = dev_alloc(h, &h->tape_dev, pl.tape_floats)) ||
        (rc = dev_alloc(h, &h->loss_buf_dev, (size_t)max_batch * (3 * unroll_steps - 1))) ||
        (rc = dev_alloc(h, &h->wgrad_partial_dev, (size_t)((max_batch + kLstmWgradRows - 1) / kLstmWgradRows) * n)) ||
        (rc = dev_alloc(h, &h->adam_m_dev, n)) || (rc = dev_alloc(h, &h->adam_v_dev, n)) ||
        (rc = dev_alloc(h, &desc_dev, pl.desc.size())) || (rc = dev_alloc(h, &param_desc_dev, pl.param_desc.size())))
        return rc;
    HIP_TRY(hipMemcpy(desc_dev, pl.desc.data(), pl.desc.size() * sizeof(LstmGradDesc), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(param_desc_dev, pl.param_desc.data(), pl.param_desc.size() * sizeof(int), hipMemcpyHostToDevice));
    pl.p.desc = desc_dev;
    pl.p.param_desc = param_desc_dev;
    pl.p.tape = h->tape_dev;
    pl.p.blob = h->blob_dev;
    pl.p.loss_buf = h->loss_buf_dev;
    h->lstm_train = pl.p;
    h->train_batch = max_batch;
    h->train_K = unroll_steps;
    h->train_step = 0;
    return MZS_OK;
}

/* the six launches of the 'lstm_model' gradient computation (mzs_train_lstm.h) */
static int lstm_train_grad_launch(mzs_handle* h, int32_t batch, const float* obs_dev, const float* actions_dev,
                                  const float* target_value_dev, const float* target_reward_dev,
                                  const float* target_policy_dev, const float* is_ratio_dev, float inv_global_batch,
                                  float* grad_dev, float* loss_dev, float* priority_dev, hipStream_t st) {
    LstmTrainParams p = h->lstm_train;
    p.B = batch;
    p.loss_value = h->loss_value, p.loss_reward = h->loss_reward;
    p.inv_batch = inv_global_batch;
    p.obs = obs_dev, p.actions = actions_dev, p.tv = target_value_dev, p.tr = target_reward_dev;
    p.tp = target_policy_dev, p.is_ratio = is_ratio_dev;
    p.priority = priority_dev, p.grad = grad_dev, p.loss_out = loss_dev;
    const int K = p.K;
    const size_t shmem = kLstmTrainLdsBytes; /* under 64 KiB: no attribute to raise */
    hipLaunchKernelGGL(k_lstm_prep, dim3(8, 4), dim3(256), 0, st, p);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_lstm_chain_fwd, dim3(1), dim3(kLstmTrainThreads), shmem, st, p);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_lstm_heads, dim3(3 * K - 1), dim3(kLstmTrainThreads), shmem, st, p);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_lstm_chain_bwd, dim3(1), dim3(kLstmTrainThreads), shmem, st, p);
    HIP_TRY(hipGetLastError());
    const int n_grad_blocks = (p.n_params + 255) / 256;
    const int n_chunks = (batch + kLstmWgradRows - 1) / kLstmWgradRows;
    hipLaunchKernelGGL(k_lstm_wgrad_partial, dim3(n_grad_blocks, n_chunks), dim3(256), 0, st, p, h->wgrad_partial_dev);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_lstm_wgrad, dim3(n_grad_blocks + 1), dim3(256), 0, st, p,
                       (const double*)h->wgrad_partial_dev, n_chunks, n_grad_blocks);
    HIP_TRY(hipGetLastError());
    return MZS_OK;
}

int mzs_train_create(mzs_handle* h, int32_t max_batch, int32_t unroll_steps) {
    if (!h) return fail(MZS_E_ARG, "null handle");
    if (max_batch < 1 || unroll_steps < 1) return fail(MZS_E_ARG, "bad batch or unroll length");
    const mzs_dims& d = h->cfg.dims;
    if (d.state_dim > 64 || d.n_actions > 64)
        return fail(MZS_E_ARG, "training kernels hold one logit per lane: state_dim and n_actions <= 64");
    if (h->tape_dev || h->vis_train) return fail(MZS_E_STATE, "mzs_train_create was already called on this handle");
    HIP_TRY(hipSetDevice(h->cfg.device));
    if (d.model_kind == MZS_MODEL_VISION) {
        std::string err;
        const size_t nb = mzs_blob_total(&d);
        int rc = vis_train_create(&h->vis_train, d, max_batch, unroll_steps, h->cfg.device, &err);
        if (rc) return fail(rc, err);
        if ((rc = dev_alloc(h, &h->adam_m_dev, nb)) || (rc = dev_alloc(h, &h->adam_v_dev, nb))) return rc;
        h->train_batch = max_batch;
        h->train_K = unroll_steps;
        h->train_step = 0;
        return MZS_OK;
    }
    if (d.model_kind == MZS_MODEL_LSTM) return lstm_train_create(h, max_batch, unroll_steps);
    const TapeLayout t = make_tape_layout(h->model, unroll_steps);
    if ((size_t)t.total * sizeof(float) > (size_t)kMaxLdsPerBlock)
        return fail(MZS_E_ARG, "one sample's unroll does not fit 160 KiB of LDS");
    const size_t n = mzs_blob_layout(&d).total;
    int rc;
    if ((rc = dev_alloc(h, &h->tape_dev, (size_t)max_batch * t.copy_n)) ||
        (rc = dev_alloc(h, &h->loss_buf_dev, (size_t)max_batch * (3 * unroll_steps - 1))) ||
        (rc = dev_alloc(h, &h->wgrad_partial_dev, (size_t)((max_batch + kWgradChunk - 1) / kWgradChunk) * n)) ||
        (rc = dev_alloc(h, &h->adam_m_dev, n)) || (rc = dev_alloc(h, &h->adam_v_dev, n)))
        return rc;
    h->tape_layout = t;
    h->train_batch = max_batch;
    h->train_K = unroll_steps;
    h->train_step = 0;
    return MZS_OK;
}

/* the three launches of the gradient computation with the tape / loss / partial-sum buffers given */
static int train_grad_launch(mzs_handle* h, const TapeLayout& tl, int K, int32_t batch, const float* obs_dev,
                             const float* actions_dev, const float* target_value_dev, const float* target_reward_dev,
                             const float* target_policy_dev, const float* is_ratio_dev, float inv_global_batch,
                             float* grad_dev, float* loss_dev, float* priority_dev, float* tape, float* loss_buf,
                             double* partial, hipStream_t st) {
    TrainParams p;
    std::memset(&p, 0, sizeof p);
    p.m = h->model;
    p.m.w = h->w_dev; /* fp32 master store: training is fp32 whatever the search precision */
    p.m.bf16 = 0;
    p.tl = tl;
    fill_tensor_descs(h->cfg.dims, p.tl, K, p.td);
    p.bo = mzs_blob_layout(&h->cfg.dims);
    p.blob = h->blob_dev;
    p.B = batch;
    p.K = K;
    p.n_params = (int)p.bo.total;
    p.loss_value = h->loss_value, p.loss_reward = h->loss_reward;
    p.inv_batch = inv_global_batch;
    p.obs = obs_dev;
    p.actions = actions_dev;
    p.tv = target_value_dev;
    p.tr = target_reward_dev;
    p.tp = target_policy_dev;
    p.is_ratio = is_ratio_dev;
    p.tape = tape;
    p.loss_buf = loss_buf;
    p.priority = priority_dev;
    p.grad = grad_dev;
    p.loss_out = loss_dev;
    const size_t shmem = (size_t)p.tl.total * sizeof(float);
    if (shmem > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_train_sample),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    hipLaunchKernelGGL(k_train_sample, dim3(batch), dim3(kWave), shmem, st, p);
    HIP_TRY(hipGetLastError());
    const int n_grad_blocks = (p.n_params + 255) / 256;
    const int n_chunks = (batch + kWgradChunk - 1) / kWgradChunk;
    hipLaunchKernelGGL(k_train_wgrad_partial, dim3(n_grad_blocks, n_chunks), dim3(256), 0, st, p, partial);
    HIP_TRY(hipGetLastError());
    const int n_sum_blocks = (p.n_params + kWgradParams - 1) / kWgradParams;
    hipLaunchKernelGGL(k_train_wgrad, dim3(n_sum_blocks + 1), dim3(256), 0, st, p, (const double*)partial, n_chunks,
                       n_sum_blocks);
    HIP_TRY(hipGetLastError());
    return MZS_OK;
}

int mzs_train_grad(mzs_handle* h, int32_t batch, const float* obs_dev, const float* actions_dev,
                   const float* target_value_dev, const float* target_reward_dev,
                   const float* target_policy_dev, const float* is_ratio_dev, float inv_global_batch,
                   float* grad_dev, float* loss_dev, float* priority_dev, void* stream) {
    if (!h) return fail(MZS_E_ARG, "null handle");
    if (!h->tape_dev && !h->vis_train) return fail(MZS_E_STATE, "mzs_train_create has not been called");
    if (!h->have_weights) return fail(MZS_E_STATE, "mzs_set_weights has not been called");
    if (batch < 1 || batch > h->train_batch) return fail(MZS_E_ARG, "batch exceeds max_batch of mzs_train_create");
    if (!obs_dev || !actions_dev || !target_value_dev || !target_reward_dev || !target_policy_dev ||
        !grad_dev || !loss_dev || !priority_dev)
        return fail(MZS_E_ARG, "null buffer");
    HIP_TRY(hipSetDevice(h->cfg.device));
    if (h->vis_train) {
        const unsigned char* masks = h->dropout_masks;
        if (masks && h->dropout_n != vis_train_mask_count(h->vis_train, batch)) {
            char msg[160];
            std::snprintf(msg, sizeof msg, "mzs_train_set_dropout gave %lld mask bytes, a step on %d samples consumes %lld",
                          (long long)h->dropout_n, batch, (long long)vis_train_mask_count(h->vis_train, batch));
            return fail(MZS_E_ARG, msg);
        }
        h->dropout_masks = nullptr; /* injected masks serve one call */
        std::string err;
        const int rc = vis_train_grad(h->vis_train, h->blob_dev, h->w_dev, batch, obs_dev, actions_dev, target_value_dev,
                                      target_reward_dev, target_policy_dev, is_ratio_dev, inv_global_batch,
                                      h->loss_value, h->loss_reward, masks, h->dropout_seed, h->train_calls++, grad_dev,
                                      loss_dev, priority_dev, static_cast<hipStream_t>(stream), &err);
        return rc ? fail(rc, err) : MZS_OK;
    }
    if (h->cfg.dims.model_kind == MZS_MODEL_LSTM)
        return lstm_train_grad_launch(h, batch, obs_dev, actions_dev, target_value_dev, target_reward_dev,
                                      target_policy_dev, is_ratio_dev, inv_global_batch, grad_dev, loss_dev,
                                      priority_dev, static_cast<hipStream_t>(stream));
    return train_grad_launch(h, h->tape_layout, h->train_K, batch, obs_dev, actions_dev, target_value_dev,
                             target_reward_dev, target_policy_dev, is_ratio_dev, inv_global_batch, grad_dev, loss_dev,
                             priority_dev, h->tape_dev, h->loss_buf_dev, h->wgrad_partial_dev,
                             static_cast<hipStream_t>(stream));
}

/* mzs_train_grad without any allocation, neither here nor beforehand (no mzs_train_create): the
 * per-sample tapes, the per-sample loss terms and the chunk partial sums of the weight gradient
 * live in this GPU's copy of the static `scratch` array.  Nothing in it has to start at zero:
 * every element the kernels read is written earlier in the same call (k_train_sample writes the
 * whole tape row and loss row of each of the `batch` samples, k_train_wgrad_partial every partial
 * sum k_train_wgrad adds), so stale contents of a previous, larger call are never read. */
int mzs_train_grad_static(mzs_handle* h, int32_t batch, int32_t unroll_steps, const float* obs_dev,
                          const float* actions_dev, const float* target_value_dev,
                          const float* target_reward_dev, const float* target_policy_dev,
                          const float* is_ratio_dev, float inv_global_batch, float* grad_dev, float* loss_dev,
                          float* priority_dev, void* stream) {
    if (!h) return fail(MZS_E_ARG, "null handle");
    if (!h->have_weights) return fail(MZS_E_STATE, "mzs_set_weights has not been called");
    if (batch < 1 || unroll_steps < 1) return fail(MZS_E_ARG, "bad batch or unroll length");
    if (!obs_dev || !actions_dev || !target_value_dev || !target_reward_dev || !target_policy_dev ||
        !grad_dev || !loss_dev || !priority_dev)
        return fail(MZS_E_ARG, "null buffer");
    const mzs_dims& d = h->cfg.dims;
    if (d.model_kind != MZS_MODEL_MLP)
        return fail(MZS_E_ARG, "mzs_train_grad_static is built for mlp_model (the lstm_model tape of a whole batch "
                               "does not fit the static array): use mzs_train_create + mzs_train_grad");
    if (d.state_dim > 64 || d.n_actions > 64)
        return fail(MZS_E_ARG, "training kernels hold one logit per lane: state_dim and n_actions <= 64");
    const TapeLayout tl = make_tape_layout(h->model, unroll_steps);
    if ((size_t)tl.total * sizeof(float) > (size_t)kMaxLdsPerBlock)
        return fail(MZS_E_ARG, "one sample's unroll does not fit 160 KiB of LDS");
    const size_t n = mzs_blob_layout(&d).total;
    const size_t n_chunks = ((size_t)batch + kWgradChunk - 1) / kWgradChunk;
    const auto up256 = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t tape_bytes = up256((size_t)batch * tl.copy_n * sizeof(float));
    const size_t loss_bytes = up256((size_t)batch * (3 * (size_t)unroll_steps - 1) * sizeof(float));
    const size_t part_bytes = up256(n_chunks * n * sizeof(double));
    if (tape_bytes + loss_bytes + part_bytes > kStaticScratchBytes) {
        char msg[192];
        std::snprintf(msg, sizeof msg,
                      "batch %d x unroll %d needs %zu bytes of scratch, the static array holds %zu "
                      "(MZS_STATIC_SCRATCH_BYTES): use mzs_train_create + mzs_train_grad for this size",
                      batch, unroll_steps, tape_bytes + loss_bytes + part_bytes, kStaticScratchBytes);
        return fail(MZS_E_ARG, msg);
    }
    HIP_TRY(hipSetDevice(h->cfg.device));
    void* sp = nullptr; /* resolved by mzs_create on this device; a table lookup here */
    const int rc = scratch_for_device(h->cfg.device, &sp);
    if (rc) return rc;
    unsigned char* base = static_cast<unsigned char*>(sp);
    float* tape = reinterpret_cast<float*>(base);
    float* loss_buf = reinterpret_cast<float*>(base + tape_bytes);
    double* partial = reinterpret_cast<double*>(base + tape_bytes + loss_bytes);
    return train_grad_launch(h, tl, unroll_steps, batch, obs_dev, actions_dev, target_value_dev, target_reward_dev,
                             target_policy_dev, is_ratio_dev, inv_global_batch, grad_dev, loss_dev, priority_dev,
                             tape, loss_buf, partial, static_cast<hipStream_t>(stream));
}

int mzs_train_set_loss(mzs_handle* h, int32_t value_criterion, int32_t reward_criterion) {
    if (!h) return fail(MZS_E_ARG, "null handle");
    const auto ok = [](int v) { return v >= MZS_LOSS_CE && v <= MZS_LOSS_ZERO; };
    if (!ok(value_criterion) || !ok(reward_criterion)) return fail(MZS_E_ARG, "unknown loss criterion (MZS_LOSS_*)");
    h->loss_value = value_criterion, h->loss_reward = reward_criterion;
    return MZS_OK;
}

int mzs_train_set_momentum(mzs_handle* h, double momentum) {
    if (!h) return fail(MZS_E_ARG, "null handle");
    if (!(momentum >= 0.0 && momentum < 1.0)) return fail(MZS_E_ARG, "momentum must be in [0, 1)");
    h->momentum = momentum;
    return MZS_OK;
}

int mzs_train_apply(mzs_handle* h, const float* grad_dev, int32_t optimizer, double lr, void* stream) {
    if (!h || !grad_dev) return fail(MZS_E_ARG, "null argument");
    if (!h->tape_dev && !h->vis_train) return fail(MZS_E_STATE, "mzs_train_create has not been called");
    if (!h->have_weights) return fail(MZS_E_STATE, "mzs_set_weights has not been called");
    if (optimizer != MZS_OPT_ADAM && optimizer != MZS_OPT_SGD) return fail(MZS_E_ARG, "unknown optimizer");
    HIP_TRY(hipSetDevice(h->cfg.device));
    h->train_step += 1;
    ApplyParams p;
    p.blob = h->blob_dev;
    p.store = h->w_dev;
    p.store16 = h->w_search_dev != h->w_dev ? h->w_search_dev : nullptr;
    p.store_index = h->vis_train ? nullptr : h->store_index_dev;
    p.is_stat = h->vis_train ? vis_train_stat_flags(h->vis_train) : nullptr;
    p.m = h->adam_m_dev;
    p.v = h->adam_v_dev;
    p.grad = grad_dev;
    p.n = (int)mzs_blob_total(&h->cfg.dims);
    p.optimizer = optimizer;
    p.step = h->train_step;
    const double bc1 = 1.0 - std::pow(h->momentum, h->train_step), bc2 = 1.0 - std::pow(0.999, h->train_step);
    p.momentum = (float)h->momentum;
    p.one_minus_momentum = (float)(1.0 - h->momentum);
    p.lr_over_bc1 = (float)(lr / bc1);
    p.inv_sqrt_bc2 = (float)(1.0 / std::sqrt(bc2));
    p.lr = (float)lr;
    hipLaunchKernelGGL(k_train_apply, dim3((p.n + 255) / 256), dim3(256), 0,
                       static_cast<hipStream_t>(stream), p);
    if (h->vis_train) { /* the Linear layers' stores of the search kernels (mzs_vision.h, mzs_vision_mm.h) */
        const mzs_dims& d = h->cfg.dims;
        const int n_store = (int)(vis_lin_total(d.hidden_dim, d.n_mid_layers, d.state_dim, d.n_actions) +
                                  vis_mm_total(d.hidden_dim, d.n_mid_layers, d.state_dim, d.n_actions));
        hipLaunchKernelGGL(k_vis_store_refresh, dim3((n_store + 255) / 256), dim3(256), 0,
                           static_cast<hipStream_t>(stream), h->w_dev, (const int*)h->vis_store_src_dev,
                           (const float*)h->blob_dev, n_store);
    }
    HIP_TRY(hipGetLastError());
    return MZS_OK;
}

int mzs_train_dropout_count(mzs_handle* h, int32_t batch, int64_t* n_out) {
    if (!h || !n_out) return fail(MZS_E_ARG, "null argument");
    if (!h->vis_train) return fail(MZS_E_STATE, "mzs_train_dropout_count: vision_model after mzs_train_create only");
    if (batch < 1 || batch > h->train_batch) return fail(MZS_E_ARG, "batch exceeds max_batch of mzs_train_create");
    *n_out = vis_train_mask_count(h->vis_train, batch);
    return MZS_OK;
}

int mzs_train_set_dropout(mzs_handle* h, const unsigned char* masks_dev, int64_t n_masks, uint64_t seed) {
    if (!h) return fail(MZS_E_ARG, "null handle");
    if (!h->vis_train) return fail(MZS_E_STATE, "mzs_train_set_dropout: vision_model after mzs_train_create only");
    if (masks_dev && n_masks < 1) return fail(MZS_E_ARG, "mask count");
    h->dropout_masks = masks_dev;
    h->dropout_n = masks_dev ? n_masks : 0;
    if (!masks_dev) h->dropout_seed = seed, h->train_calls = 0;
    return MZS_OK;
}

int mzs_get_weights(mzs_handle* h, float* blob_out, size_t n_floats) {
    if (!h || !blob_out) return fail(MZS_E_ARG, "null argument");
    if (!h->have_weights) return fail(MZS_E_STATE, "mzs_set_weights has not been called");
    if (n_floats != mzs_blob_total(&h->cfg.dims)) return fail(MZS_E_ARG, "wrong blob size");
    HIP_TRY(hipSetDevice(h->cfg.device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(blob_out, h->blob_dev, n_floats * sizeof(float), hipMemcpyDefault));
    return MZS_OK;
}

int mzs_fill_streams(mzs_handle* h, int32_t n_games, int32_t num_simulations, int32_t train,
                     const int64_t* seeds_dev, double alpha, double lo, double hi,
                     double* dirichlet_dev, double* noise_dev, int64_t noise_stride, void* stream) {
    const int rc = check_fill_args(h, n_games, num_simulations, train, seeds_dev, alpha, dirichlet_dev, noise_dev,
                                   noise_stride);
    if (rc) return rc;
    if (n_games == 0) return MZS_OK;
    HIP_TRY(hipSetDevice(h->cfg.device));
    return fill_launch(h, n_games, train && num_simulations > 0, seeds_dev, alpha, lo, hi, dirichlet_dev, noise_dev,
                       noise_stride, mzs_max_draws(h->cfg.dims.n_actions, num_simulations), nullptr,
                       static_cast<hipStream_t>(stream));
}

}  // extern "C" (reopened below)

/* ------------------------------------------------------------------ self-play arena */

struct mzs_arena {
    mzs_arena_config cfg;
    ArenaArrays a;
    long long queued_end = -1; /* host mirror of ctl[1] (virtual row behind the last queued game) */
    int* alive_count_dev = nullptr;
    int* status_host = nullptr; /* [kArenaRing][8] pinned, coherent: the moves' status entries (ArenaArrays::status_ring) */
    std::vector<void*> allocs;
};

namespace {
template <typename T>
int arena_alloc(mzs_arena* ar, T** out, size_t n) {
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, n * sizeof(T) + 256));
    HIP_TRY(hipMemset(p, 0, n * sizeof(T) + 256));
    ar->allocs.push_back(p);
    *out = static_cast<T*>(p);
    return MZS_OK;
}
}  // namespace

extern "C" {

int mzs_arena_create(const mzs_arena_config* cfg, mzs_arena** out) {
    if (!cfg || !out) return fail(MZS_E_ARG, "null argument");
    if (cfg->env != MZS_ENV_CARTPOLE) return fail(MZS_E_ARG, "unknown environment (only MZS_ENV_CARTPOLE)");
    if (cfg->n_slots < 1 || cfg->n_rows < cfg->n_slots || cfg->max_moves < 1)
        return fail(MZS_E_ARG, "need n_slots >= 1, n_rows >= n_slots, max_moves >= 1");
    HIP_TRY(hipSetDevice(cfg->device));
    mzs_arena* ar = new (std::nothrow) mzs_arena();
    if (!ar) return fail(MZS_E_ARG, "out of host memory");
    ar->cfg = *cfg;
    ArenaArrays& a = ar->a;
    a.G = cfg->n_slots, a.n_rows = cfg->n_rows, a.T = cfg->max_moves, a.obs_dim = 4, a.A = 2;
    const size_t G = a.G, R = a.n_rows, T = a.T;
    int rc;
    if ((rc = arena_alloc(ar, &a.state, G * 4)) || (rc = arena_alloc(ar, &a.alive, G)) ||
        (rc = arena_alloc(ar, &a.row, G)) || (rc = arena_alloc(ar, &a.seed, G)) ||
        (rc = arena_alloc(ar, &a.episode, G)) || (rc = arena_alloc(ar, &a.stream_seed, G)) ||
        (rc = arena_alloc(ar, &a.ctl, 4)) || (rc = arena_alloc(ar, &a.row_seed, R)) ||
        (rc = arena_alloc(ar, &a.vrow, G)) || (rc = arena_alloc(ar, &a.row_episode, R)) ||
        (rc = arena_alloc(ar, &a.row_temp, R)) ||
        (rc = arena_alloc(ar, &a.length, R)) ||
        (rc = arena_alloc(ar, &a.obs, R * T * a.obs_dim)) || (rc = arena_alloc(ar, &a.action, R * T)) ||
        (rc = arena_alloc(ar, &a.reward, R * T)) || (rc = arena_alloc(ar, &a.root_value, R * T)) ||
        (rc = arena_alloc(ar, &a.policy, R * T * a.A)) || (rc = arena_alloc(ar, &ar->alive_count_dev, 1))) {
        mzs_arena_destroy(ar);
        return rc;
    }
    void* ring = nullptr;
    void* ring_dev = nullptr;
    if (hipHostMalloc(&ring, sizeof(int) * 8 * kArenaRing, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer(&ring_dev, ring, 0) != hipSuccess) {
        if (ring) (void)hipHostFree(ring);
        mzs_arena_destroy(ar);
        return fail(MZS_E_HIP, "pinned host memory for the arena's status entries");
    }
    ar->status_host = static_cast<int*>(ring);
    for (int i = 0; i < 8 * kArenaRing; ++i) ar->status_host[i] = -1;
    a.status_ring = static_cast<int*>(ring_dev);
    *out = ar;
    return MZS_OK;
}

void mzs_arena_destroy(mzs_arena* ar) {
    if (!ar) return;
    (void)hipSetDevice(ar->cfg.device);
    (void)hipDeviceSynchronize(); /* no kernel may still write the status entries */
    for (void* p : ar->allocs) (void)hipFree(p);
    if (ar->status_host) (void)hipHostFree(ar->status_host);
    delete ar;
}

int32_t mzs_arena_obs_dim(const mzs_arena* ar) { return ar ? ar->a.obs_dim : 0; }
int32_t mzs_arena_n_actions(const mzs_arena* ar) { return ar ? ar->a.A : 0; }

int mzs_arena_begin(mzs_arena* ar, int32_t row0, int32_t n_games, const int64_t* seeds_dev, int32_t episode,
                    float* obs_dev, void* stream) {
    if (!ar || !seeds_dev || !obs_dev) return fail(MZS_E_ARG, "null argument");
    if (n_games < 1 || row0 < 0 || row0 + n_games > ar->a.n_rows)
        return fail(MZS_E_ARG, "rows [row0, row0 + n_games) out of range");
    HIP_TRY(hipSetDevice(ar->cfg.device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(k_arena_rows, dim3((n_games + 255) / 256), dim3(256), 0, st, ar->a, row0, n_games,
                       reinterpret_cast<const long long*>(seeds_dev), episode, (double)NAN);
    hipLaunchKernelGGL(k_arena_begin, dim3((ar->a.G + 255) / 256), dim3(256), 0, st, ar->a, row0, n_games, episode,
                       obs_dev);
    HIP_TRY(hipGetLastError());
    ar->queued_end = (long long)row0 + n_games;
    return MZS_OK;
}

int mzs_arena_enqueue(mzs_arena* ar, int32_t n_games, const int64_t* seeds_dev, int32_t episode, double temperature,
                      void* stream) {
    if (!ar || !seeds_dev) return fail(MZS_E_ARG, "null argument");
    if (n_games < 1 || n_games > ar->a.n_rows) return fail(MZS_E_ARG, "n_games must be in [1, n_rows]");
    if (ar->queued_end < 0) return fail(MZS_E_STATE, "mzs_arena_begin has not been called");
    if (ar->queued_end + n_games > 0x7fffffffLL) return fail(MZS_E_ARG, "virtual row counter would overflow");
    HIP_TRY(hipSetDevice(ar->cfg.device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(k_arena_rows, dim3((n_games + 255) / 256), dim3(256), 0, st, ar->a, (int)ar->queued_end, n_games,
                       reinterpret_cast<const long long*>(seeds_dev), episode, temperature);
    hipLaunchKernelGGL(k_arena_enqueue, dim3(1), dim3(64), 0, st, ar->a, n_games);
    HIP_TRY(hipGetLastError());
    ar->queued_end += n_games;
    return MZS_OK;
}

static int arena_step_launch(mzs_arena* ar, const int32_t* visits_dev, const float* root_value_dev,
                             const double* root_prior_dev, const long long* used, double temperature,
                             int32_t num_simulations, float* obs_dev, int32_t* alive_count_dev, hipStream_t st) {
    const int from_prior = num_simulations < 1 ? 1 : 0;
    /* two launches on purpose: one block stepping every slot before it refills them (a fused
     * k_arena_move) was measured at 59 us per move against 8 + 9 us for this pair */
    hipLaunchKernelGGL(k_arena_step, dim3((ar->a.G + 255) / 256), dim3(256), 0, st, ar->a, visits_dev, root_value_dev,
                       root_prior_dev, temperature, from_prior, obs_dev);
    hipLaunchKernelGGL(k_arena_refill, dim3(1), dim3(kRefillThreads), 0, st, ar->a, obs_dev, alive_count_dev, used);
    HIP_TRY(hipGetLastError());
    return MZS_OK;
}

int mzs_arena_step(mzs_arena* ar, const int32_t* visits_dev, const float* root_value_dev,
                   const double* root_prior_dev, double temperature, int32_t num_simulations, float* obs_dev,
                   int32_t* alive_count_dev, void* stream) {
    if (!ar || !visits_dev || !root_value_dev || !root_prior_dev || !obs_dev || !alive_count_dev)
        return fail(MZS_E_ARG, "null argument");
    HIP_TRY(hipSetDevice(ar->cfg.device));
    return arena_step_launch(ar, visits_dev, root_value_dev, root_prior_dev, nullptr, temperature, num_simulations,
                             obs_dev, alive_count_dev, static_cast<hipStream_t>(stream));
}

int mzs_arena_play(mzs_arena* ar, mzs_handle* h, int32_t n_moves, int32_t num_simulations, int32_t train,
                   double alpha, double lo, double hi, int32_t prefix_draws, double temperature, float* obs_dev,
                   double* dirichlet_dev, double* noise_dev, int64_t noise_stride, int32_t* alive_count_dev,
                   void* stream) {
    if (!ar || !h || !obs_dev || !alive_count_dev) return fail(MZS_E_ARG, "null argument");
    if (n_moves < 0) return fail(MZS_E_ARG, "bad move count");
    if (ar->cfg.device != h->cfg.device) return fail(MZS_E_ARG, "arena and search handle live on different devices");
    if (ar->a.G > h->cfg.max_games) return fail(MZS_E_ARG, "the arena has more slots than the handle has games");
    for (int m = 0; m < n_moves; ++m) {
        int rc = mzs_search_streams(h, ar->a.G, num_simulations, train, obs_dev,
                                    reinterpret_cast<const int64_t*>(ar->a.stream_seed), alpha, lo, hi, prefix_draws,
                                    dirichlet_dev, noise_dev, noise_stride, stream);
        if (rc) return rc;
        if ((rc = arena_step_launch(ar, h->res.visits, h->res.root_value, h->res.root_prior, h->res.consumed,
                                    temperature, num_simulations, obs_dev, alive_count_dev,
                                    static_cast<hipStream_t>(stream))))
            return rc;
    }
    return MZS_OK;
}

int mzs_arena_status(mzs_arena* ar, int32_t move, int32_t* out6) {
    if (!ar || !out6) return fail(MZS_E_ARG, "null argument");
    if (move < 0) return fail(MZS_E_ARG, "bad move index");
    const volatile int* e = ar->status_host + (size_t)(move % kArenaRing) * 8;
    if (e[0] != move || e[6] != move)
        return fail(MZS_E_STATE, "that move has not finished yet, or its entry has been overwritten");
    for (int i = 0; i < 6; ++i) out6[i] = e[i];
    if (e[0] != move || e[6] != move) return fail(MZS_E_STATE, "the entry was overwritten while it was read");
    return MZS_OK;
}

int mzs_arena_make_batch(mzs_arena* ar, int32_t batch, const int32_t* row_pos_dev, int32_t unroll_steps,
                         int32_t td_steps, const double* discount_pow_dev, float* obs_dev, float* actions_dev,
                         float* target_value_dev, float* target_reward_dev, float* target_policy_dev,
                         void* stream) {
    if (!ar || !row_pos_dev || !discount_pow_dev || !obs_dev || !actions_dev || !target_value_dev ||
        !target_reward_dev || !target_policy_dev)
        return fail(MZS_E_ARG, "null argument");
    if (batch < 0 || unroll_steps < 1 || td_steps < 0) return fail(MZS_E_ARG, "bad sizes");
    if (batch == 0) return MZS_OK;
    HIP_TRY(hipSetDevice(ar->cfg.device));
    BatchParams p;
    p.a = ar->a;
    p.B = batch, p.K = unroll_steps, p.td = td_steps;
    p.row_pos = row_pos_dev;
    p.disc_pow = discount_pow_dev;
    p.obs = obs_dev, p.actions = actions_dev, p.tv = target_value_dev, p.tr = target_reward_dev;
    p.tp = target_policy_dev;
    const int n = batch * unroll_steps;
    hipLaunchKernelGGL(k_arena_make_batch, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), p);
    HIP_TRY(hipGetLastError());
    return MZS_OK;
}

int mzs_arena_priority(mzs_arena* ar, int32_t row0, int32_t n_rows, int32_t td_steps,
                       const double* discount_pow_dev, int32_t out_stride, double* priority_dev, void* stream) {
    if (!ar || !discount_pow_dev || !priority_dev) return fail(MZS_E_ARG, "null argument");
    if (row0 < 0 || n_rows < 0 || row0 + n_rows > ar->a.n_rows || td_steps < 0 || out_stride < 1)
        return fail(MZS_E_ARG, "rows [row0, row0 + n_rows) out of range or bad sizes");
    if (n_rows == 0) return MZS_OK;
    HIP_TRY(hipSetDevice(ar->cfg.device));
    const long long total = (long long)n_rows * out_stride;
    if (total > 0x7fffffffLL) return fail(MZS_E_ARG, "n_rows * out_stride too large");
    hipLaunchKernelGGL(k_arena_priority, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), ar->a, row0, n_rows, td_steps, discount_pow_dev,
                       out_stride, priority_dev);
    HIP_TRY(hipGetLastError());
    return MZS_OK;
}

int mzs_arena_priority_packed(mzs_arena* ar, int32_t row0, int32_t n_rows, int32_t td_steps,
                              const double* discount_pow_dev, int32_t longest, const int64_t* offsets_dev,
                              double* priority_dev, void* stream) {
    if (!ar || !discount_pow_dev || !priority_dev || !offsets_dev) return fail(MZS_E_ARG, "null argument");
    if (row0 < 0 || n_rows < 0 || row0 + n_rows > ar->a.n_rows || td_steps < 0 || longest < 1)
        return fail(MZS_E_ARG, "rows [row0, row0 + n_rows) out of range or bad sizes");
    if (n_rows == 0) return MZS_OK;
    HIP_TRY(hipSetDevice(ar->cfg.device));
    const long long total = (long long)n_rows * longest;
    if (total > 0x7fffffffLL) return fail(MZS_E_ARG, "n_rows * longest too large");
    hipLaunchKernelGGL(k_arena_priority_packed, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), ar->a, row0, n_rows, td_steps, discount_pow_dev, longest,
                       reinterpret_cast<const long long*>(offsets_dev), priority_dev);
    HIP_TRY(hipGetLastError());
    return MZS_OK;
}

int mzs_replay_sample_positions(int32_t batch, const double* const* rows, const int32_t* row_len, const double* u,
                                int64_t* pos, double* mouve) {
    if (batch < 0 || (batch > 0 && (!rows || !row_len || !pos || !mouve))) return fail(MZS_E_ARG, "null argument");
    const int bad = replay_sample_positions(batch, rows, row_len, u, pos, mouve);
    if (bad) {
        char msg[96];
        std::snprintf(msg, sizeof msg, "sample %d: probabilities contain NaN or the position is out of range", bad - 1);
        return fail(MZS_E_ARG, msg);
    }
    return MZS_OK;
}

int mzs_replay_update_value(int32_t batch, int32_t num_unroll, double* const* rows, const int32_t* row_len,
                            const int32_t* game_len, const int64_t* game_idx, const int64_t* pos,
                            const double* new_value, double* prio_game) {
    if (batch < 0 || num_unroll < 0 ||
        (batch > 0 && (!rows || !row_len || !game_len || !game_idx || !pos || !new_value || !prio_game)))
        return fail(MZS_E_ARG, "null argument");
    for (int32_t b = 0; b < batch; ++b)
        if (row_len[b] < 1 || pos[b] < 0 || game_idx[b] < 0) return fail(MZS_E_ARG, "bad row length or position");
    replay_update_value(batch, num_unroll, rows, row_len, game_len, game_idx, pos, new_value, prio_game);
    return MZS_OK;
}

void* mzs_arena_array(mzs_arena* ar, int32_t which) {
    if (!ar) return nullptr;
    switch (which) {
        case MZS_ARENA_LENGTH: return ar->a.length;
        case MZS_ARENA_OBS: return ar->a.obs;
        case MZS_ARENA_ACTION: return ar->a.action;
        case MZS_ARENA_REWARD: return ar->a.reward;
        case MZS_ARENA_ROOT_VALUE: return ar->a.root_value;
        case MZS_ARENA_POLICY: return ar->a.policy;
        case MZS_ARENA_ALIVE: return ar->a.alive;
        case MZS_ARENA_STATE: return ar->a.state;
        case MZS_ARENA_STREAM_SEED: return ar->a.stream_seed;
        case MZS_ARENA_SLOT_ROW: return ar->a.row;
        case MZS_ARENA_QUEUE: return ar->a.ctl;
        default: return nullptr;
    }
}

}  // extern "C"
