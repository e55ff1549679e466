/*
 * mzs_train.h — the unroll-K training step of the MLP model on the GPU (gfx950): forward,
 * losses, backward, weight gradients, optimizer.  Included by mzsearch.hip (one library).
 *
 * Reference code restated (file:line in /root/reference):
 *   k_train_sample  muzero_model.py:609-647 compute_forward, :650-734 evaluate_loss (per sample),
 *                   :536-563 transform_with_support, neural_network_mlp_model.py:240-243
 *                   cross_entropy, :135-143 scale_to_bound_action (and its gradient)
 *   k_train_wgrad   the weight / bias gradients autograd accumulates over the batch and over
 *                   every use of a tensor (tied mid layers, shared trunks, K unroll steps) +
 *                   neural_network_mlp_model.py:280-285 l2 + the batch means of :603,:734
 *   k_train_apply   torch.optim.Adam / SGD(momentum 0.9), muzero_model.py:418-421,:764
 *
 * Layout: one wavefront per sample walks the whole unroll with its activations and deltas on
 * an LDS "tape" (lane = neuron), then copies the tape to HBM; the weight gradient is a second,
 * embarrassingly parallel pass (one thread per weight, reduction over samples and uses), so no
 * atomics and a fixed summation order.  Data-parallel ranks all-reduce the flat gradient between
 * k_train_wgrad and k_train_apply (muzero_amd/train.py).
 */
#ifndef MZS_TRAIN_H
#define MZS_TRAIN_H

#include "mzs_device.h"
#include "mzs_train_loss.h"

namespace mzs {

struct TapeLayout {
    int HP, SP, AP, XP, OP, OPH, ODH; /* padded widths (multiples of 64)                 */
    int L, K;
    int r_x, r_T, r_din, r_dmid, r_dout;                                  /* representation   */
    int p_base, p_stride, p_h, p_T, p_din, p_dmid, p_dpol, p_dval, p_out; /* K prediction recs */
    int d_base, d_stride, d_x, d_T, d_din, d_dmid, d_drew, d_dst, d_out;  /* K-1 dynamics recs */
    int copy_n;                                                           /* floats copied to HBM */
    int s_zs, s_gT, s_gT2, s_gh, s_ghn, s_scl;                            /* LDS-only scratch */
    int total;
};

/* one parameter tensor of the blob and where its (input, delta) pairs sit on a sample's tape */
struct TensorDesc {
    int w_off, b_off, n_out, n_in;
    int x_off, d_off;                     /* first use                                    */
    int outer_n, outer_stride;            /* unroll steps                                 */
    int inner_n, inner_sx, inner_sd;      /* tied mid-layer uses                          */
    int l2;                               /* 1: part of the reference's l2 term           */
};

constexpr int kTrainTensors = 11;

struct TrainParams {
    Model m;
    TapeLayout tl;
    TensorDesc td[kTrainTensors];
    mzs_blob_offsets bo;
    const float* blob; /* master weights, torch layout (include/mzs_layout.h) */
    int B, K, n_params;
    int loss_value, loss_reward; /* MZS_LOSS_*: criterion of the value / reward terms (mzs_train_set_loss) */
    float inv_batch;
    const float* obs;
    const float* actions;
    const float* tv;
    const float* tr;
    const float* tp;
    const float* is_ratio;
    float* tape;       /* [B][copy_n]   */
    float* loss_buf;   /* [B][3K-1]     */
    float* priority;   /* [K][B]        */
    float* grad;       /* [n_params]    */
    float* loss_out;   /* [3K-1] batch means of the raw terms, then loss_nn */
};

/* y = act(b + W x) from the transposed device store; x has kdim entries */
template <bool kElu>
__device__ __forceinline__ void train_dense(const float* __restrict__ w, const Layer ly, int kdim,
                                            const float* x, float* out, int lane) {
    for (int j0 = 0; j0 < ly.n_pad; j0 += kWave) {
        const int j = j0 + lane;
        const float* __restrict__ wt = w + ly.w_off + j;
        float e = w[ly.b_off + j], o = 0.0f;
        int k = 0;
        /* sixteen weight rows requested before the first fma that needs one: the chains are short and
         * the wave is alone on its SIMD, so the loop is bound by load latency, not by issue (same
         * chains, same order: even k into e, odd k into o) */
        for (; k + 15 < kdim; k += 16) {
            float wv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) wv[u] = wt[(size_t)(k + u) * ly.n_pad];
#pragma unroll
            for (int u = 0; u < 16; u += 2) {
                e = mzs_fma(wv[u], x[k + u], e);
                o = mzs_fma(wv[u + 1], x[k + u + 1], o);
            }
        }
        for (; k + 1 < kdim; k += 2) {
            e = mzs_fma(wt[(size_t)k * ly.n_pad], x[k], e);
            o = mzs_fma(wt[(size_t)(k + 1) * ly.n_pad], x[k + 1], o);
        }
        if (k < kdim) e = mzs_fma(wt[(size_t)k * ly.n_pad], x[k], e);
        float acc = e + o;
        if (kElu) acc = mzs_elu(acc);
        out[j] = acc;
    }
    wave_sync();
}

/* gx[k] (+)= sum_j W[j][k] d[j]; W in torch layout [nout][ld] */
__device__ __forceinline__ void train_back_input(const float* __restrict__ W, int nout, int nin, int ld,
                                                 const float* d, float* gx, bool accumulate, int lane) {
    for (int k0 = 0; k0 < nin; k0 += kWave) {
        const int k = k0 + lane;
        if (k < nin) {
            float e = accumulate ? gx[k] : 0.0f, o = 0.0f;
            int j = 0;
            for (; j + 15 < nout; j += 16) { /* as train_dense: sixteen rows in flight */
                float wv[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) wv[u] = W[(size_t)(j + u) * ld + k];
#pragma unroll
                for (int u = 0; u < 16; u += 2) {
                    e = mzs_fma(wv[u], d[j + u], e);
                    o = mzs_fma(wv[u + 1], d[j + u + 1], o);
                }
            }
            for (; j + 1 < nout; j += 2) {
                e = mzs_fma(W[(size_t)j * ld + k], d[j], e);
                o = mzs_fma(W[(size_t)(j + 1) * ld + k], d[j + 1], o);
            }
            if (j < nout) e = mzs_fma(W[(size_t)j * ld + k], d[j], e);
            gx[k] = e + o;
        }
    }
    wave_sync();
}

struct ScaleRec {
    int imin, imax;
    float sc;
};

/* scale_to_bound_action on z[0..n), n <= 64 */
__device__ __forceinline__ ScaleRec train_scale_fwd(const float* z, int n, float* h, int lane) {
    const bool valid = lane < n;
    const float v = valid ? z[lane] : 0.0f;
    const float pinf = __builtin_inff();
    const float mn = wave_min(valid ? v : pinf), mx = wave_max(valid ? v : -pinf);
    ScaleRec r;
    r.imin = __ffsll((long long)__builtin_amdgcn_ballot_w64(valid && v == mn)) - 1;
    r.imax = __ffsll((long long)__builtin_amdgcn_ballot_w64(valid && v == mx)) - 1;
    r.sc = mzs_bound_scale(mn, mx);
    if (valid) h[lane] = (v - mn) / r.sc;
    wave_sync();
    return r;
}

/* g = dL/dh  ->  gz = dL/dz (min and max are taken at one index each, as torch's backward does) */
__device__ __forceinline__ void train_scale_bwd(const float* g, const float* h, int n, const ScaleRec r,
                                                float* gz, int lane) {
    const bool valid = lane < n;
    const float gv = valid ? g[lane] : 0.0f, hv = valid ? h[lane] : 0.0f;
    const float gsum = train_sum(gv), ghs = train_sum(gv * hv);
    const float dsc = -ghs / r.sc;
    const float dmn = -gsum / r.sc - dsc;
    float o = gv / r.sc;
    if (lane == r.imin) o += dmn;
    if (lane == r.imax) o += dsc;
    if (valid) gz[lane] = o;
    wave_sync();
}

/* backward through one MLP trunk: gT holds dL/d(last trunk activation) on entry */
__device__ __forceinline__ void train_net_bwd(const float* __restrict__ Win, int nin, int ld,
                                              const float* __restrict__ Wmid, int H, int HP, int L,
                                              const float* T, float* din, float* dmid, float* gT,
                                              float* gT2, float* gx, int lane) {
    for (int l = L; l >= 1; --l) {
        const float* t = T + (size_t)l * HP;
        float* d = dmid + (size_t)(l - 1) * HP;
        for (int j = lane; j < H; j += kWave) d[j] = gT[j] * (t[j] > 0.0f ? 1.0f : t[j] + 1.0f);
        wave_sync();
        train_back_input(Wmid, H, H, H, d, gT2, false, lane);
        for (int j = lane; j < H; j += kWave) gT[j] = gT2[j];
        wave_sync();
    }
    for (int j = lane; j < H; j += kWave) din[j] = gT[j] * (T[j] > 0.0f ? 1.0f : T[j] + 1.0f);
    wave_sync();
    if (gx) train_back_input(Win, H, nin, ld, din, gx, false, lane);
}

__global__ void __launch_bounds__(64) k_train_sample(const TrainParams p) {
    extern __shared__ __align__(16) float T[];
    const int lane = lane_id();
    const int b = blockIdx.x;
    if (b >= p.B) return;
    const TapeLayout& t = p.tl;
    const Model& m = p.m;
    const float* __restrict__ w = m.w;
    const float* __restrict__ bl = p.blob;
    const mzs_blob_offsets& o = p.bo;
    const int S = m.Sd, A = m.A, H = m.H, L = m.L, K = p.K, HP = t.HP, B = p.B;
    const int n_terms = 3 * K - 1;

    for (int i = lane; i < t.total; i += kWave) T[i] = 0.0f;
    wave_sync();
    const float wgt = (p.is_ratio ? p.is_ratio[b] : 1.0f) * p.inv_batch;
    ScaleRec* scl = reinterpret_cast<ScaleRec*>(T + t.s_scl);

    /* ---------------------------------------------------------------- forward */
    for (int k = lane; k < m.obs_dim; k += kWave) T[t.r_x + k] = p.obs[(size_t)b * m.obs_dim + k];
    wave_sync();
    train_dense<true>(w, m.r_in, m.obs_dim, T + t.r_x, T + t.r_T, lane);
    for (int l = 0; l < L; ++l)
        train_dense<true>(w, m.r_mid, H, T + t.r_T + l * HP, T + t.r_T + (l + 1) * HP, lane);
    train_dense<false>(w, m.r_out, H, T + t.r_T + L * HP, T + t.s_zs, lane);
    {
        const ScaleRec r = train_scale_fwd(T + t.s_zs, S, T + t.p_base + t.p_h, lane);
        if (lane == 0) scl[0] = r;
    }
    for (int k = 0; k < K; ++k) {
        float* P = T + t.p_base + k * t.p_stride;
        train_dense<true>(w, m.p_in, S, P + t.p_h, P + t.p_T, lane);
        for (int l = 0; l < L; ++l)
            train_dense<true>(w, m.p_mid, H, P + t.p_T + l * HP, P + t.p_T + (l + 1) * HP, lane);
        train_dense<false>(w, m.p_heads, H, P + t.p_T + L * HP, P + t.p_out, lane);
        if (k + 1 < K) {
            float* D = T + t.d_base + k * t.d_stride;
            for (int i = lane; i < S; i += kWave) D[t.d_x + i] = P[t.p_h + i];
            for (int a = lane; a < A; a += kWave) D[t.d_x + S + a] = p.actions[((size_t)k * B + b) * A + a];
            wave_sync();
            train_dense<true>(w, m.d_in, S + A, D + t.d_x, D + t.d_T, lane);
            for (int l = 0; l < L; ++l)
                train_dense<true>(w, m.d_mid, H, D + t.d_T + l * HP, D + t.d_T + (l + 1) * HP, lane);
            train_dense<false>(w, m.d_heads, H, D + t.d_T + L * HP, D + t.d_out, lane);
            const ScaleRec r = train_scale_fwd(D + t.d_out + m.seg_state, S, P + t.p_stride + t.p_h, lane);
            if (lane == 0) scl[k + 1] = r;
        }
    }
    wave_sync();

    /* ------------------------------------------------- losses + backward through time */
    float* gT = T + t.s_gT;
    float* gT2 = T + t.s_gT2;
    float* gh = T + t.s_gh;
    float* ghn = T + t.s_ghn; /* dL/dh_k arriving from dynamics step k (zero for the last step) */
    for (int k = K - 1; k >= 0; --k) {
        float* P = T + t.p_base + k * t.p_stride;
        const float gs = (k > 0 ? 1.0f / (float)K : 1.0f) * wgt;
        const int t0 = k == 0 ? 0 : 3 * k - 1;
        /* value: single-bin support target */
        int idx;
        float wt, sm;
        const float tvk = p.tv[(size_t)k * B + b];
        train_support_target(tvk, S, &idx, &wt);
        const float lv = train_ce(P + t.p_out + m.seg_value, S, lane == idx ? wt : 0.0f, gs, P + t.p_dval,
                                  &sm, lane, p.loss_value);
        {
            /* new_priority (muzero_model.py:710-713): |h^-1(value) - target| */
            const float y = train_sum(lane < S ? (float)(lane - S / 2) * sm : 0.0f);
            if (lane == 0) p.priority[(size_t)k * B + b] = fabsf(mzs_inverse_value_transform(y) - tvk);
        }
        const float ptar = lane < A ? p.tp[((size_t)k * B + b) * A + lane] : 0.0f;
        const float lp = train_ce(P + t.p_out, A, ptar, gs, P + t.p_dpol, &sm, lane, train_policy_mode(p.loss_value));
        if (lane == 0) {
            p.loss_buf[(size_t)b * n_terms + t0] = lv;
            p.loss_buf[(size_t)b * n_terms + t0 + 1] = lp;
        }
        /* prediction heads -> trunk -> hidden state */
        train_back_input(bl + o.p_val_w, S, H, H, P + t.p_dval, gT, false, lane);
        train_back_input(bl + o.p_pol_w, A, H, H, P + t.p_dpol, gT, true, lane);
        train_net_bwd(bl + o.p_in_w, S, S, bl + o.p_mid_w, H, HP, L, P + t.p_T, P + t.p_din, P + t.p_dmid,
                      gT, gT2, gh, lane);
        if (k > 0) {
            float* D = T + t.d_base + (k - 1) * t.d_stride;
            /* total gradient of h_k, halved (register_hook, muzero_model.py:640) */
            for (int i = lane; i < S; i += kWave) gh[i] = (gh[i] + ghn[i]) * 0.5f;
            wave_sync();
            train_scale_bwd(gh, P + t.p_h, S, scl[k], D + t.d_dst, lane);
            train_support_target(p.tr[(size_t)k * B + b], S, &idx, &wt);
            const float lr = train_ce(D + t.d_out, S, lane == idx ? wt : 0.0f, gs, D + t.d_drew, &sm, lane,
                                      p.loss_reward);
            if (lane == 0) p.loss_buf[(size_t)b * n_terms + t0 + 2] = lr;
            train_back_input(bl + o.d_st_w, S, H, H, D + t.d_dst, gT, false, lane);
            train_back_input(bl + o.d_rew_w, S, H, H, D + t.d_drew, gT, true, lane);
            /* the action columns of the input get a gradient nobody uses: ask for the first S only */
            train_net_bwd(bl + o.d_in_w, S, S + A, bl + o.d_mid_w, H, HP, L, D + t.d_T, D + t.d_din,
                          D + t.d_dmid, gT, gT2, ghn, lane);
        } else {
            for (int i = lane; i < S; i += kWave) gh[i] = gh[i] + ghn[i];
            wave_sync();
            train_scale_bwd(gh, P + t.p_h, S, scl[0], T + t.r_dout, lane);
            train_back_input(bl + o.r_out_w, S, H, H, T + t.r_dout, gT, false, lane);
            train_net_bwd(bl + o.r_in_w, m.obs_dim, m.obs_dim, bl + o.r_mid_w, H, HP, L, T + t.r_T,
                          T + t.r_din, T + t.r_dmid, gT, gT2, nullptr, lane);
        }
    }
    wave_sync();
    float* dst = p.tape + (size_t)b * t.copy_n;
    for (int i = lane; i < t.copy_n; i += kWave) dst[i] = T[i];
}

constexpr int kWgradChunk = 16; /* samples per partial sum */

/* Weight gradients, pass 1: thread (blob element i, sample chunk c) = sum over the chunk's
 * samples and every use of the tensor of delta[n] * x[k]  ->  partial[c][i].  Fixed order, no
 * atomics.  (blockIdx.y = chunk) */
__global__ void __launch_bounds__(256) k_train_wgrad_partial(const TrainParams p, double* partial) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.n_params) return;
    int ti = 0, n = 0, kk = -1;
    for (int q = 0; q < kTrainTensors; ++q) {
        const TensorDesc& d = p.td[q];
        const int nw = d.n_out * d.n_in;
        if (i >= d.w_off && i < d.w_off + nw) {
            ti = q, n = (i - d.w_off) / d.n_in, kk = (i - d.w_off) % d.n_in;
        } else if (i >= d.b_off && i < d.b_off + d.n_out) {
            ti = q, n = i - d.b_off, kk = -1;
        }
    }
    const TensorDesc d = p.td[ti];
    const int b0 = blockIdx.y * kWgradChunk, b1 = min(b0 + kWgradChunk, p.B);
    double acc = 0.0;
    /* per use of the tensor, the chunk's sixteen samples are requested together (the loop waited out one trip to
     * L2 / HBM per product before); a missing sample of the last chunk reads sample b1 - 1 and adds nothing */
    for (int u = 0; u < d.outer_n; ++u)
        for (int v = 0; v < d.inner_n; ++v) {
            const int doff = d.d_off + u * d.outer_stride + v * d.inner_sd + n;
            const int xoff = d.x_off + u * d.outer_stride + v * d.inner_sx + (kk < 0 ? 0 : kk);
            float dl[kWgradChunk], x[kWgradChunk];
#pragma unroll
            for (int s = 0; s < kWgradChunk; ++s) {
                const float* tape = p.tape + (size_t)min(b0 + s, b1 - 1) * p.tl.copy_n;
                dl[s] = tape[doff];
                x[s] = tape[xoff];
            }
#pragma unroll
            for (int s = 0; s < kWgradChunk; ++s)
                if (b0 + s < b1) acc += (double)dl[s] * (kk < 0 ? 1.0 : (double)x[s]);
        }
    partial[(size_t)blockIdx.y * p.n_params + i] = acc;
}

/* pass 2: sum the chunk partials (+ l2); the last block reduces the loss terms.  A gradient block = 64
 * parameters x 4 groups of chunks (group g: chunks g, g + 4, ..., eight requested at a time), the groups added
 * in order through LDS: a fixed summation order with 32 x fewer dependent trips to memory than one thread per
 * parameter walking all chunks. */
constexpr int kWgradParams = 64; /* parameters per block of k_train_wgrad */
__global__ void __launch_bounds__(256) k_train_wgrad(const TrainParams p, const double* partial,
                                                     int n_chunks, int n_grad_blocks) {
    const int K = p.K, B = p.B, n_terms = 3 * K - 1;
    __shared__ double red[256];
    if ((int)blockIdx.x < n_grad_blocks) {
        const int i = blockIdx.x * kWgradParams + (threadIdx.x & (kWgradParams - 1)), g = threadIdx.x / kWgradParams;
        const int ii = min(i, p.n_params - 1);
        double part = 0.0;
        for (int c0 = g; c0 < n_chunks; c0 += 4 * 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = c0 + 4 * u;
                v[u] = c < n_chunks ? partial[(size_t)c * p.n_params + ii] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) part += v[u];
        }
        /* the batch's total sample weight (the l2 term's factor): the samples spread over the block's threads */
        __shared__ double wsum_s[4];
        double wpart = 0.0;
        for (int b = threadIdx.x; b < B; b += blockDim.x) wpart += (double)((p.is_ratio ? p.is_ratio[b] : 1.0f) * p.inv_batch);
        for (int off = kWave / 2; off > 0; off >>= 1) wpart += __shfl_down(wpart, off, kWave);
        if ((threadIdx.x & (kWave - 1)) == 0) wsum_s[threadIdx.x / kWave] = wpart;
        red[threadIdx.x] = part;
        __syncthreads();
        if (g != 0 || i >= p.n_params) return;
        double acc = ((red[threadIdx.x] + red[threadIdx.x + kWgradParams]) + red[threadIdx.x + 2 * kWgradParams]) +
                     red[threadIdx.x + 3 * kWgradParams];
        int l2 = 0;
        for (int q = 0; q < kTrainTensors; ++q) {
            const TensorDesc& d = p.td[q];
            if ((i >= d.w_off && i < d.w_off + d.n_out * d.n_in) || (i >= d.b_off && i < d.b_off + d.n_out))
                l2 = d.l2;
        }
        if (l2) {
            const double wsum = ((wsum_s[0] + wsum_s[1]) + wsum_s[2]) + wsum_s[3];
            acc += 2.0 * 1e-4 * wsum * (double)p.blob[i];
        }
        p.grad[i] = (float)acc;
        return;
    }
    /* loss block: batch means of the raw terms (muzero_model.py:603) and loss_nn (:725-734) */
    double sq = 0.0;
    for (int q = 0; q < kTrainTensors; ++q) {
        const TensorDesc& d = p.td[q];
        if (!d.l2) continue;
#pragma unroll 8
        for (int i = threadIdx.x; i < d.n_out * d.n_in; i += blockDim.x) sq += (double)p.blob[d.w_off + i] * p.blob[d.w_off + i];
        for (int i = threadIdx.x; i < d.n_out; i += blockDim.x) sq += (double)p.blob[d.b_off + i] * p.blob[d.b_off + i];
    }
    double tot = 0.0, wsum = 0.0;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const double wb = (double)((p.is_ratio ? p.is_ratio[b] : 1.0f) * p.inv_batch);
        double sl = 0.0;
        for (int q = 0; q < n_terms; ++q) sl += (double)p.loss_buf[(size_t)b * n_terms + q];
        tot += wb * sl;
        wsum += wb;
    }
    const auto block_sum = [&](double v) {
        red[threadIdx.x] = v;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
            __syncthreads();
        }
        const double r = red[0];
        __syncthreads();
        return r;
    };
    sq = block_sum(sq);
    tot = block_sum(tot);
    wsum = block_sum(wsum);
    if (threadIdx.x == 0) p.loss_out[n_terms] = (float)(tot + wsum * 1e-4 * sq);
    /* a term's batch mean: one wavefront per term, a lane per sample, a fixed shuffle tree */
    const int lane = threadIdx.x & (kWave - 1);
    for (int q = threadIdx.x / kWave; q < n_terms; q += blockDim.x / kWave) {
        double s = 0.0;
        for (int b = lane; b < B; b += kWave) s += (double)p.loss_buf[(size_t)b * n_terms + q];
        for (int off = kWave / 2; off > 0; off >>= 1) s += __shfl_down(s, off, kWave);
        if (lane == 0) p.loss_out[q] = (float)(s * (double)p.inv_batch);
    }
}

struct ApplyParams {
    float* blob;
    float* store;
    float* store16;         /* MZS_PRECISION_BF16: the search's copy (weights rounded to bfloat16), else null */
    const int* store_index; /* blob element -> offset in the device weight store; bit 30: a bias; null: no store
                               (vision_model: k_vis_store_refresh rebuilds its Linear stores) */
    const unsigned char* is_stat; /* vision_model: 1 = a BatchNorm running statistic, whose "gradient" slot holds
                                     the (rank-summed) change of this step: added, not optimised; else null */
    float* m;
    float* v;
    const float* grad;
    int n, optimizer, step;
    float lr_over_bc1, inv_sqrt_bc2, lr;
    float momentum, one_minus_momentum; /* Adam beta1 / SGD momentum (0.9 unless OneCycleLR cycles it) */
};

/* torch.optim.Adam (betas momentum/0.999, eps 1e-8) or SGD(momentum); refreshes the transposed
 * store the search kernels read */
__global__ void __launch_bounds__(256) k_train_apply(const ApplyParams p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.n) return;
    const float g = p.grad[i];
    float w = p.blob[i];
    if (p.is_stat && p.is_stat[i]) {
        p.blob[i] = w + g;
        return;
    }
    if (p.optimizer == 0) {
        float m = p.m[i], v = p.v[i];
        m = m + (g - m) * p.one_minus_momentum;
        v = v * 0.999f + 0.001f * g * g;
        const float denom = __builtin_sqrtf(v) * p.inv_sqrt_bc2 + 1e-8f;
        w = w - p.lr_over_bc1 * (m / denom);
        p.m[i] = m;
        p.v[i] = v;
    } else {
        const float buf = p.step == 1 ? g : p.momentum * p.m[i] + g;
        p.m[i] = buf;
        w = w - p.lr * buf;
    }
    p.blob[i] = w;
    if (!p.store_index) return;
    const int si = p.store_index[i], so = si & 0x3fffffff;
    if (si == -1) return; /* not in the store: a search never reads it (LSTM W_hh, forget-gate rows) */
    p.store[so] = w;
    if (p.store16) p.store16[so] = ((si >> 30) & 1) ? w : mzs_bf16_round(w);
}

/* vision_model: the Linear layers' stores (quads for mzs_vision.h, [K/4][N_pad][4] for mzs_vision_mm.h) are
 * permutations of blob elements with zero padding: src[i] = blob index or -1 */
__global__ void __launch_bounds__(256) k_vis_store_refresh(float* __restrict__ store, const int* __restrict__ src,
                                                           const float* __restrict__ blob, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) store[i] = src[i] >= 0 ? blob[src[i]] : 0.0f;
}

}  // namespace mzs
#endif
