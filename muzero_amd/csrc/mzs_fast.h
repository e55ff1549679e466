/*
 * mzs_fast.h — register-resident specialisation of the fused search kernel for
 * the small MLP shapes of the benchmark configs (L = 0, H <= 64, S_dim <= 32,
 * A <= 32; compile-time SD / NA / H).
 *
 * Differences from the generic kernel (same arithmetic, bit for bit):
 *   - lane j keeps row j of all four recurrent-inference weight matrices in VGPRs
 *     (2*SD + 2*H registers), so a simulation streams no weights at all;
 *   - the two heads of each net sit in the two 32-lane halves of the wave (reward |
 *     next state, policy | value) and are reduced together with DPP butterflies;
 *   - hidden states stay in LDS for the whole search and reach HBM once at the end.
 * One wave walks several games (grid-stride), loading its weights once.
 */
#ifndef MZS_FAST_H
#define MZS_FAST_H

#include "mzs_device.h"

namespace mzs {

/* acc = bias; acc = fma(w[k], x[k], acc) for k ascending; x broadcast from LDS */
template <int K>
__device__ __forceinline__ float dot_chain(const float (&w)[K], const float* x, float acc) {
    constexpr int K4 = K / 4 * 4;
#pragma unroll
    for (int k = 0; k < K4; k += 4) {
        const float4 xv = *reinterpret_cast<const float4*>(x + k);
        acc = mzs_fma(w[k + 0], xv.x, acc);
        acc = mzs_fma(w[k + 1], xv.y, acc);
        acc = mzs_fma(w[k + 2], xv.z, acc);
        acc = mzs_fma(w[k + 3], xv.w, acc);
    }
#pragma unroll
    for (int k = K4; k < K; ++k) acc = mzs_fma(w[k], x[k], acc);
    return acc;
}

/* Lane j keeps row j of the two K = H head matrices in VGPRs; the two K = SD input
 * matrices sit in LDS, one copy per block, as [k/4][lane][4] so a lane fetches four
 * consecutive k with one ds_read_b128 (256 VGPRs per wave cannot hold all four). */
template <int SD, int NA, int H>
struct FastWeights {
    float d_h[H], p_h[H];
    float b_d_in, b_d_h, b_p_in, b_p_h;
    const float4* d_in4;  /* LDS, [(SD+3)/4][64] */
    const float4* p_in4;  /* LDS, [(SD+3)/4][64] */
};

/* acc = fma chain over k < K with weights from the block's LDS copy */
template <int K>
__device__ __forceinline__ float dot_chain_lds(const float4* w4, const float* x, float acc, int lane) {
    constexpr int K4 = K / 4 * 4;
#pragma unroll
    for (int k = 0; k < K4; k += 4) {
        const float4 wv = w4[(k / 4) * 64 + lane];
        const float4 xv = *reinterpret_cast<const float4*>(x + k);
        acc = mzs_fma(wv.x, xv.x, acc);
        acc = mzs_fma(wv.y, xv.y, acc);
        acc = mzs_fma(wv.z, xv.z, acc);
        acc = mzs_fma(wv.w, xv.w, acc);
    }
    if (K4 < K) {
        const float4 wv = w4[(K4 / 4) * 64 + lane];
        acc = mzs_fma(wv.x, x[K4], acc);
        if (K4 + 1 < K) acc = mzs_fma(wv.y, x[K4 + 1], acc);
        if (K4 + 2 < K) acc = mzs_fma(wv.z, x[K4 + 2], acc);
    }
    return acc;
}

/* per-wave LDS slice of the fast kernel */
struct FastLds {
    Node* nodes;
    unsigned short* cb;
    unsigned short* path;
    unsigned* meta;       /* [n_nodes] parent | depth << 16                      */
    unsigned short* best; /* [n_nodes] best child of an expanded node, 0 = leaf  */
    unsigned short* leafs; /* [S] node expanded by each simulation                */
    double* pbc;          /* [S+2] sqrt(n) * pb_c(n)                             */
    double* nwin;         /* [kNoiseWin] window of the tie-break stream          */
    double* rootp;        /* [NA] root-child priors after noise (f64)            */
    float* hid;           /* [S+1][32] hidden state of every expansion           */
    float* xa;            /* [64] trunk activations                              */
    float* onehot;        /* [NA][64] action columns of the dynamics input       */
    float* xo;            /* [64] observation (root only)                        */
};

constexpr int kNoiseWin = 256;

struct FastLayout {
    int off_cb, off_path, off_meta, off_best, off_leafs, off_pbc, off_nwin, off_rootp, off_hid, off_xa,
        off_onehot, off_xo, bytes;
};

/* Block-cooperative: fill the shared LDS copies (caller __syncthreads()) and this
 * lane's register rows. */
template <int SD, int NA, int H>
__device__ __forceinline__ void fast_load_weights(const Model& m, FastWeights<SD, NA, H>& W,
                                                  float* lds_d_in, float* lds_p_in, int lane) {
    const float* w = m.w;
    constexpr int KG = (SD + 3) / 4;
    for (int i = threadIdx.x; i < KG * 64 * 4; i += blockDim.x) {
        const int e = i & 3, j = (i >> 2) & 63, k = (i >> 8) * 4 + e;
        lds_d_in[i] = k < SD ? w[m.d_in.w_off + k * 64 + j] : 0.0f;
        lds_p_in[i] = k < SD ? w[m.p_in.w_off + k * 64 + j] : 0.0f;
    }
    W.d_in4 = reinterpret_cast<const float4*>(lds_d_in);
    W.p_in4 = reinterpret_cast<const float4*>(lds_p_in);
#pragma unroll
    for (int k = 0; k < H; ++k) W.d_h[k] = w[m.d_heads.w_off + k * 64 + lane];
#pragma unroll
    for (int k = 0; k < H; ++k) W.p_h[k] = w[m.p_heads.w_off + k * 64 + lane];
    W.b_d_in = w[m.d_in.b_off + lane];
    W.b_d_h = w[m.d_heads.b_off + lane];
    W.b_p_in = w[m.p_in.b_off + lane];
    W.b_p_h = w[m.p_heads.b_off + lane];
}

/* Prediction_function on the hidden state at `h` (LDS): lanes < NA get their prior,
 * the value comes back uniform.  Policy logits sit in the low 32-lane half, value logits
 * in the high half; both softmaxes run in the same instructions. */
template <int SD, int NA, int H>
__device__ __forceinline__ float fast_prediction(const FastWeights<SD, NA, H>& W,
                                                 const FastLds& L, const float* h, float* prior,
                                                 int lane) {
    const float t = mzs_elu(dot_chain_lds<SD>(W.p_in4, h, W.b_p_in, lane));
    L.xa[lane] = t;
    wave_sync();
    const float o = dot_chain<H>(W.p_h, L.xa, W.b_p_h);
    const int i = lane & 31;
    const bool lo = lane < 32;
    const bool valid = lo ? i < NA : i < SD;
    const Halves mh = halves_max(valid ? o : -__builtin_inff());
    const float e = valid ? mzs_expf(o - (lo ? mh.lo : mh.hi)) : 0.0f;
    const Halves Zh = halves_tree_sum(e);
    const float pr = e / (lo ? Zh.lo : Zh.hi);
    *prior = pr;
    const float tv = (!lo && valid) ? (float)(i - SD / 2) * pr : 0.0f;
    const float y = halves_tree_sum(tv).hi;
    wave_sync();
    return mzs_inverse_value_transform(y);
}

/* Dynamics_function on the hidden state at `h` + action: next hidden state written to
 * `hn` (LDS), reward returned uniform.  Reward logits in the low half, next-state
 * pre-activations in the high half. */
template <int SD, int NA, int H>
__device__ __forceinline__ float fast_dynamics(const FastWeights<SD, NA, H>& W, const FastLds& L,
                                               const float* h, int action, float* hn, int lane) {
    float acc = dot_chain_lds<SD>(W.d_in4, h, W.b_d_in, lane);
    acc = acc + L.onehot[action * 64 + lane];
    L.xa[lane] = mzs_elu(acc);
    wave_sync();
    const float o = dot_chain<H>(W.d_h, L.xa, W.b_d_h);
    const int i = lane & 31;
    const bool lo = lane < 32;
    const bool valid = i < SD;
    const float ninf = -__builtin_inff();
    const Halves mx = halves_max(valid ? o : ninf);
    const Halves nmn = halves_max(valid ? -o : ninf);
    /* reward head */
    const float e = (valid && lo) ? mzs_expf(o - mx.lo) : 0.0f;
    const float Z = halves_tree_sum(e).lo;
    const float pr = e / Z;
    const float tr = (valid && lo) ? (float)(i - SD / 2) * pr : 0.0f;
    const float y = halves_tree_sum(tr).lo;
    /* next state: scale_to_bound_action */
    const float mn = -nmn.hi;
    const float sc = mzs_bound_scale(mn, mx.hi);
    if (!lo && valid) hn[i] = (o - mn) / sc;
    wave_sync();
    return mzs_inverse_value_transform(y);
}

/* ------------------------------------------------------------------ select
 * ucb_score for EVERY child in the tree at once (one child per lane), then the
 * descent is a pointer chase over best[].  The tie-break draw of a child at depth d
 * is draw consumed + (d-1)*A + a whichever path is taken (mcts.py:163 draws once per
 * child scored, level by level), so off-path scores cost nothing but lanes. */
struct FastState {
    float mn, mx;
    long long consumed, win_base, path_total;
    int maxd, status;
};

__device__ __forceinline__ void fast_noise_refill(const SearchConst& c, FastState& s,
                                                  const FastLds& L, int lane) {
    s.win_base = s.consumed;
#pragma unroll
    for (int j = 0; j < kNoiseWin / kWave; ++j) {
        const long long i = s.win_base + j * kWave + lane;
        L.nwin[j * kWave + lane] = (c.noise && i < c.noise_cap) ? c.noise[i] : 0.0;
    }
    wave_sync();
}

template <int kCtrl>
__device__ __forceinline__ double dpp_mov_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), kCtrl, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), kCtrl, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

template <int kCtrl>
__device__ __forceinline__ void argmax_step(double& bs, int& ba) {
    const double os = dpp_mov_f64<kCtrl>(bs);
    const int oa = __builtin_amdgcn_update_dpp(0, ba, kCtrl, 0xF, 0xF, false);
    /* max over (score, action) tuples: ties -> larger action (mcts.py:167) */
    const bool take = (os > bs) || (os == bs && oa > ba);
    bs = take ? os : bs;
    ba = take ? oa : ba;
}

template <int NA>
__device__ __forceinline__ int fast_select(const SearchConst& c, FastState& s, const FastLds& L,
                                           int sim, int* depth, int* action, int* pbase, int lane) {
    static_assert(NA == 2 || NA == 4 || NA == 8 || NA == 16, "siblings must fill an aligned lane group");
    const int nch = NA * (sim + 1);
    if (s.consumed + (long long)NA * s.maxd > s.win_base + kNoiseWin) fast_noise_refill(c, s, L, lane);
    for (int c0 = 1; c0 <= nch; c0 += kWave) {
        const int ci = c0 + lane;
        const bool valid = ci <= nch;
        const int cc = valid ? ci : 1;
        const Node ch = L.nodes[cc];
        const unsigned meta = L.meta[cc];
        const int par = meta & 0xffff, dep = (int)(meta >> 16);
        const int np = L.nodes[par].visit;
        const double tn = L.pbc[np];
        const int a = (cc - 1) & (NA - 1);
        const long long ni = s.consumed + (long long)(dep - 1) * NA + a;
        const int wi = (int)(ni - s.win_base);
        double u;
        if (wi < kNoiseWin)
            u = L.nwin[wi];
        else
            u = (c.noise && ni < c.noise_cap) ? c.noise[ni] : 0.0;
        /* ucb_score (mcts.py:155-163) */
        const double pd = par == 0 ? L.rootp[a] : (double)ch.prior;
        const double ps = (tn * pd) / (double)(ch.visit + 1);
        double vs = 0.0;
        if (ch.visit > 0) {
            const float q = ch.vsum / (float)ch.visit;
            float x = c.g32 * q;
            x = ch.reward + x;
            if (s.mx > s.mn) x = (x - s.mn) / (s.mx - s.mn);
            vs = (double)x;
        }
        double bs = (ps + vs) + u;
        int ba = a;
        argmax_step<kDppQuadXor1>(bs, ba);
        if (NA >= 4) argmax_step<kDppQuadXor2>(bs, ba);
        if (NA >= 8) argmax_step<kDppRowHalfMirror>(bs, ba);
        if (NA >= 16) argmax_step<kDppRowMirror>(bs, ba);
        if (valid && ba == a) L.best[par] = (unsigned short)cc;
    }
    wave_sync();
    /* descent (mcts.py:172-177) */
    int node = 0, d = 0;
    if (lane == 0) L.path[0] = 0;
    for (;;) {
        const int b = __builtin_amdgcn_readfirstlane((int)L.best[node]);
        if (b == 0) break;
        node = b;
        ++d;
        if (lane == 0) L.path[d] = (unsigned short)b;
    }
    const int act = (node - 1) & (NA - 1);
    s.consumed += (long long)NA * d;
    if (s.consumed > c.noise_cap) s.status = 1;
    s.path_total += d;
    *depth = d;
    *action = act;
    *pbase = node - act;
    wave_sync();
    return node;
}

}  // namespace mzs
#endif
