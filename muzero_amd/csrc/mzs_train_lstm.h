/*
 * mzs_train_lstm.h — the unroll-K training step of the 'lstm_model' on the GPU (gfx950).
 * Included by mzsearch.hip (one library).
 *
 * Reference code restated (file:line in /root/reference):
 *   neural_network_lstm_model.py:19-21   representation = one Linear
 *   neural_network_lstm_model.py:33-47   dynamics heads  (reward, next state): Linear -> nn.LSTM
 *   neural_network_lstm_model.py:63-77   prediction heads (policy, value):     Linear -> nn.LSTM
 *   muzero_model.py:609-647 compute_forward, :650-734 evaluate_loss, neural_network_mlp_model.py
 *   :240-243 cross_entropy, :280-285 l2 (its zip() stops after the representation's two tensors)
 *
 * The reference hands every head a 2-D (B, in) tensor, which nn.LSTM reads as ONE unbatched
 * sequence of length B from a zero state: the batch axis is the time axis, rows are not
 * independent and autograd runs back along the batch.  So the parallel axis of the MLP kernels
 * (one wave per sample) does not exist here; what is parallel is
 *   - the 3K-1 head calls that do not feed the next hidden state (policy_k, value_k, reward_k):
 *     one workgroup each, forward + losses + backward in one launch (k_lstm_heads);
 *   - inside a call everything but the recurrence itself (the Linear, W_ih x_t for all t, and
 *     their transposes in the backward pass): all 16 waves of the workgroup;
 * and what is serial is the next-state chain st_0 -> st_1 -> ... (one workgroup walks it forward,
 * k_lstm_chain_fwd, and backward, k_lstm_chain_bwd) and, inside every call, the B-step
 * recurrence: one wave, lane = LSTM unit, W_hh packed in LDS so that a step is n/4 x (one
 * 128-bit weight read + one broadcast read of h + 4 FMAs) per lane.
 * Weight gradients are a separate, embarrassingly parallel pass over the stored deltas
 * (k_lstm_wgrad_partial / k_lstm_wgrad), fixed summation order, no atomics.
 */
#ifndef MZS_TRAIN_LSTM_H
#define MZS_TRAIN_LSTM_H

#include "mzs_train.h"

namespace mzs {

constexpr int kLstmTrainThreads = 1024;
enum { kHeadRew = 0, kHeadSt = 1, kHeadPol = 2, kHeadVal = 3 };

struct LstmTrainHead {
    int lin_w, lin_b;                                                       /* blob offsets   */
    int w_ih[kLstmMaxLayers], w_hh[kLstmMaxLayers], b_ih[kLstmMaxLayers], b_hh[kLstmMaxLayers];
    int in_dim, n;          /* Linear input width (S_dim or S_dim + A), LSTM width            */
    long long base, stride; /* tape offset of call 0, floats between calls                    */
    long long wt_lin, wt_ih[kLstmMaxLayers]; /* tape offsets of W_lin^T [in][H], W_ih^T [in_l][4n] */
    /* field offsets inside one call's block (rows of max_batch capacity) */
    int u, du, dx, dout;
    int gates[kLstmMaxLayers], c[kLstmMaxLayers], h[kLstmMaxLayers], dz[kLstmMaxLayers], dmid[kLstmMaxLayers];
};

/* one column range of one parameter tensor and where its (delta, input) rows sit */
struct LstmGradDesc {
    int w_off, ld, n_rows, n_cols; /* element (r, c) = blob[w_off + r * ld + c]; n_cols = 0: a bias   */
    long long d_base, d_stride;    /* delta rows: tape[d_base + call * d_stride + t * d_ld + r]       */
    long long x_base, x_stride;    /* input rows: src[x_base + call * x_stride + (t - t_shift) * x_ld + c] */
    int d_ld, x_ld, n_calls, t_shift;
    int x_src;                     /* 0 tape, 1 observations, 2 actions [K][B][A] (call stride B * A) */
    int l2;
};

struct LstmTrainParams {
    const float* blob;
    int B, Bcap, K, L, H, Sd, A, obs_dim, n_params;
    int r_w, r_b;
    float inv_batch;
    const float *obs, *actions, *tv, *tr, *tp, *is_ratio;
    float* tape;
    long long hstate, gh; /* [K][Bcap][Sd]: hidden states h_k; total gradient of h_k */
    LstmTrainHead head[4];
    const LstmGradDesc* desc;
    const int* param_desc; /* blob element -> index into desc */
    int n_desc;
    float *loss_buf, *priority, *grad, *loss_out;
};

__device__ __forceinline__ float* lt_field(const LstmTrainParams& p, int hd, int k, int off) {
    return p.tape + p.head[hd].base + (long long)k * p.head[hd].stride + off;
}
__device__ __forceinline__ float* lt_hstate(const LstmTrainParams& p, int k) {
    return p.tape + p.hstate + (long long)k * p.Bcap * p.Sd;
}
__device__ __forceinline__ float* lt_gh(const LstmTrainParams& p, int k) {
    return p.tape + p.gh + (long long)k * p.Bcap * p.Sd;
}
/* h rows of layer l; the next-state head's last layer writes straight into the hidden-state array */
__device__ __forceinline__ float* lt_layer_h(const LstmTrainParams& p, int hd, int k, int l) {
    return hd == kHeadSt && l == p.L - 1 ? lt_hstate(p, k + 1) : lt_field(p, hd, k, p.head[hd].h[l]);
}

/* Gate activations of the training recurrence on the hardware exp2 / reciprocal (1 ulp each): the
 * training path is held to a tolerance (1e-5 of the largest gradient), not to the bit-exact
 * contract of the search kernels, and the recurrence is latency-bound on exactly these. */
__device__ __forceinline__ float lt_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * x));
}
__device__ __forceinline__ float lt_tanh(float x) {
    /* (1 - e) / (1 + e), e = exp(-2|x|): no cancellation against 1 for large |x|, sign restored */
    const float e = __builtin_amdgcn_exp2f(-2.88539008177792681472f * fabsf(x));
    return copysignf((1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e), x);
}

/* LDS of the recurrence: packed W_hh (n_k4 * n float4), then h (64) and dz (256) rows */
__host__ __device__ inline int lstm_train_lds_floats(int n_max) {
    const int nk = (n_max + 3) / 4 * 4;
    return 4 * nk * n_max + 64 + 256;
}

/* Forward of call k of head hd: rows x_t = (h_k[t], action_k[t]) -> u, gates, c, h on the tape. */
__device__ void lstm_head_fwd(const LstmTrainParams& p, int hd, int k, float* lds) {
    const LstmTrainHead& hh = p.head[hd];
    const int B = p.B, H = p.H, n = hh.n, Sd = p.Sd, A = p.A, L = p.L;
    const int tid = threadIdx.x, nt = blockDim.x, n4 = 4 * n, nk = (n + 3) / 4 * 4;
    const float* xh = lt_hstate(p, k);
    const float* __restrict__ xa = hd < 2 ? p.actions + (size_t)k * B * A : nullptr;
    float* u = lt_field(p, hd, k, hh.u);
    {
        const float* __restrict__ WT = p.tape + hh.wt_lin;
        for (int idx = tid; idx < B * H; idx += nt) {
            const int t = idx / H, j = idx - t * H;
            float acc = p.blob[hh.lin_b + j];
            for (int kk = 0; kk < Sd; ++kk) acc = fmaf(WT[kk * H + j], xh[(size_t)t * Sd + kk], acc);
            if (xa)
                for (int a = 0; a < A; ++a) acc = fmaf(WT[(Sd + a) * H + j], xa[(size_t)t * A + a], acc);
            u[idx] = acc;
        }
    }
    float4* W4 = reinterpret_cast<float4*>(lds);
    float* hbuf = lds + 4 * nk * n;
    for (int l = 0; l < L; ++l) {
        const int in_l = l == 0 ? H : n;
        /* W_hh -> LDS as [k][unit] float4 (i, f, g, o rows of that unit), zero rows up to nk */
        const float* __restrict__ Whh = p.blob + hh.w_hh[l];
        for (int idx = tid; idx < nk * n; idx += nt) {
            const int kk = idx / n, j = idx - kk * n;
            float4 w = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (kk < n) {
                w.x = Whh[(size_t)j * n + kk];
                w.y = Whh[(size_t)(n + j) * n + kk];
                w.z = Whh[(size_t)(2 * n + j) * n + kk];
                w.w = Whh[(size_t)(3 * n + j) * n + kk];
            }
            W4[idx] = w;
        }
        __syncthreads(); /* u (or the layer below) is complete */
        const float* xs = l == 0 ? u : lt_layer_h(p, hd, k, l - 1);
        const float* __restrict__ WT = p.tape + hh.wt_ih[l];
        float* g = lt_field(p, hd, k, hh.gates[l]);
        const float* __restrict__ bi = p.blob + hh.b_ih[l];
        const float* __restrict__ bh = p.blob + hh.b_hh[l];
        /* input half of the gate pre-activations for every row at once */
        for (int idx = tid; idx < B * n4; idx += nt) {
            const int t = idx / n4, jj = idx - t * n4;
            float acc = bi[jj] + bh[jj];
            const float* x = xs + (size_t)t * in_l;
            for (int kk = 0; kk < in_l; ++kk) acc = fmaf(WT[kk * n4 + jj], x[kk], acc);
            g[idx] = acc;
        }
        __syncthreads();
        /* the recurrence along the batch axis: wave 0, lane = unit */
        if (tid < kWave) {
            const int lane = tid, j = lane < n ? lane : 0;
            const bool act = lane < n;
            float* cb = lt_field(p, hd, k, hh.c[l]);
            float* hb = lt_layer_h(p, hd, k, l);
            hbuf[lane] = 0.0f;
            wave_sync();
            float cprev = 0.0f;
            float z0 = g[j], z1 = g[n + j], z2 = g[2 * n + j], z3 = g[3 * n + j];
            for (int t = 0; t < B; ++t) {
                float y0 = 0.0f, y1 = 0.0f, y2 = 0.0f, y3 = 0.0f;
                if (t + 1 < B) { /* next row's input half: in flight during this step */
                    const float* gn = g + (size_t)(t + 1) * n4;
                    y0 = gn[j], y1 = gn[n + j], y2 = gn[2 * n + j], y3 = gn[3 * n + j];
                }
                for (int kk = 0; kk < nk; kk += 4) {
                    const float4 hv = *reinterpret_cast<const float4*>(hbuf + kk);
                    const float4 w0 = W4[kk * n + j], w1 = W4[(kk + 1) * n + j], w2 = W4[(kk + 2) * n + j],
                                 w3 = W4[(kk + 3) * n + j];
                    z0 = fmaf(w0.x, hv.x, z0), z1 = fmaf(w0.y, hv.x, z1), z2 = fmaf(w0.z, hv.x, z2), z3 = fmaf(w0.w, hv.x, z3);
                    z0 = fmaf(w1.x, hv.y, z0), z1 = fmaf(w1.y, hv.y, z1), z2 = fmaf(w1.z, hv.y, z2), z3 = fmaf(w1.w, hv.y, z3);
                    z0 = fmaf(w2.x, hv.z, z0), z1 = fmaf(w2.y, hv.z, z1), z2 = fmaf(w2.z, hv.z, z2), z3 = fmaf(w2.w, hv.z, z3);
                    z0 = fmaf(w3.x, hv.w, z0), z1 = fmaf(w3.y, hv.w, z1), z2 = fmaf(w3.z, hv.w, z2), z3 = fmaf(w3.w, hv.w, z3);
                }
                const float ig = lt_sigmoid(z0), fg = lt_sigmoid(z1), gg = lt_tanh(z2), og = lt_sigmoid(z3);
                const float cv = fg * cprev + ig * gg;
                const float hv = og * lt_tanh(cv);
                wave_sync(); /* every lane has read the old h */
                if (act) {
                    float* gt = g + (size_t)t * n4;
                    gt[j] = ig, gt[n + j] = fg, gt[2 * n + j] = gg, gt[3 * n + j] = og;
                    cb[(size_t)t * n + j] = cv;
                    hb[(size_t)t * n + j] = hv;
                    hbuf[lane] = hv;
                }
                wave_sync();
                cprev = cv;
                z0 = y0, z1 = y1, z2 = y2, z3 = y3;
            }
        }
        __syncthreads();
    }
}

/* Backward of call k of head hd.  dOut [B][n] = dL/d(output rows).  Leaves on the tape: dz of every
 * layer, du, and dx [B][Sd] = dL/d(h_k rows) through this call. */
__device__ void lstm_head_bwd(const LstmTrainParams& p, int hd, int k, const float* dOut, float* lds) {
    const LstmTrainHead& hh = p.head[hd];
    const int B = p.B, H = p.H, n = hh.n, Sd = p.Sd, L = p.L, in = hh.in_dim;
    const int tid = threadIdx.x, nt = blockDim.x, n4 = 4 * n;
    float4* Wb = reinterpret_cast<float4*>(lds);
    float* dzb = lds + 4 * ((n + 3) / 4 * 4) * n + 64;
    float* du = lt_field(p, hd, k, hh.du);
    for (int l = L - 1; l >= 0; --l) {
        const int in_l = l == 0 ? H : n;
        /* W_hh -> LDS as [row group][unit] float4: rows 4g..4g+3 of column `unit` */
        const float* __restrict__ Whh = p.blob + hh.w_hh[l];
        for (int idx = tid; idx < n * n; idx += nt) {
            const int gq = idx / n, kk = idx - gq * n;
            Wb[idx] = make_float4(Whh[(size_t)(4 * gq) * n + kk], Whh[(size_t)(4 * gq + 1) * n + kk],
                                  Whh[(size_t)(4 * gq + 2) * n + kk], Whh[(size_t)(4 * gq + 3) * n + kk]);
        }
        __syncthreads(); /* dOut of this layer is complete */
        const float* g = lt_field(p, hd, k, hh.gates[l]);
        const float* cb = lt_field(p, hd, k, hh.c[l]);
        float* dz = lt_field(p, hd, k, hh.dz[l]);
        const float* dO = l == L - 1 ? dOut : lt_field(p, hd, k, hh.dmid[l]);
        if (tid < kWave) {
            const int lane = tid, j = lane < n ? lane : 0;
            const bool act = lane < n;
            for (int i = lane; i < 256; i += kWave) dzb[i] = 0.0f;
            wave_sync();
            float dcn = 0.0f, dhr = 0.0f;
            int t = B - 1;
            const float* gt = g + (size_t)t * n4;
            float ig = gt[j], fg = gt[n + j], gg = gt[2 * n + j], og = gt[3 * n + j];
            float cv = cb[(size_t)t * n + j], cp = t > 0 ? cb[(size_t)(t - 1) * n + j] : 0.0f;
            float dov = dO[(size_t)t * n + j];
            for (; t >= 0; --t) {
                float nig = 0.0f, nfg = 0.0f, ngg = 0.0f, nog = 0.0f, ncp = 0.0f, ndo = 0.0f;
                if (t > 0) { /* the previous row's record: in flight during this step */
                    const float* gp = g + (size_t)(t - 1) * n4;
                    nig = gp[j], nfg = gp[n + j], ngg = gp[2 * n + j], nog = gp[3 * n + j];
                    ncp = t > 1 ? cb[(size_t)(t - 2) * n + j] : 0.0f;
                    ndo = dO[(size_t)(t - 1) * n + j];
                }
                const float dh = dov + dhr;
                const float tc = lt_tanh(cv);
                const float dc = dh * og * (1.0f - tc * tc) + dcn;
                const float d0 = dc * gg * ig * (1.0f - ig);
                const float d1 = dc * cp * fg * (1.0f - fg);
                const float d2 = dc * ig * (1.0f - gg * gg);
                const float d3 = dh * tc * og * (1.0f - og);
                dcn = dc * fg;
                if (act) {
                    float* dzt = dz + (size_t)t * n4;
                    dzt[j] = d0, dzt[n + j] = d1, dzt[2 * n + j] = d2, dzt[3 * n + j] = d3;
                    dzb[j] = d0, dzb[n + j] = d1, dzb[2 * n + j] = d2, dzb[3 * n + j] = d3;
                }
                wave_sync();
                /* dL/dh_{t-1} through W_hh: lane = unit of h_{t-1} */
                float a0 = 0.0f, a1 = 0.0f;
                for (int gq = 0; gq + 1 < n; gq += 2) {
                    const float4 w0 = Wb[gq * n + j], w1 = Wb[(gq + 1) * n + j];
                    const float4 e0 = *reinterpret_cast<const float4*>(dzb + 4 * gq);
                    const float4 e1 = *reinterpret_cast<const float4*>(dzb + 4 * gq + 4);
                    a0 = fmaf(w0.x, e0.x, a0), a0 = fmaf(w0.y, e0.y, a0), a0 = fmaf(w0.z, e0.z, a0), a0 = fmaf(w0.w, e0.w, a0);
                    a1 = fmaf(w1.x, e1.x, a1), a1 = fmaf(w1.y, e1.y, a1), a1 = fmaf(w1.z, e1.z, a1), a1 = fmaf(w1.w, e1.w, a1);
                }
                if (n & 1) {
                    const float4 w0 = Wb[(n - 1) * n + j];
                    const float4 e0 = *reinterpret_cast<const float4*>(dzb + 4 * (n - 1));
                    a0 = fmaf(w0.x, e0.x, a0), a0 = fmaf(w0.y, e0.y, a0), a0 = fmaf(w0.z, e0.z, a0), a0 = fmaf(w0.w, e0.w, a0);
                }
                dhr = a0 + a1;
                wave_sync();
                cv = cp; /* c_{t-1} */
                ig = nig, fg = nfg, gg = ngg, og = nog, cp = ncp, dov = ndo;
            }
        }
        __syncthreads();
        /* dL/d(layer input rows) = W_ih^T dz, every row at once */
        const float* __restrict__ Wih = p.blob + hh.w_ih[l];
        float* dst = l == 0 ? du : lt_field(p, hd, k, hh.dmid[l - 1]);
        for (int idx = tid; idx < B * in_l; idx += nt) {
            const int t = idx / in_l, kk = idx - t * in_l;
            const float* d = dz + (size_t)t * n4;
            float a0 = 0.0f, a1 = 0.0f;
            int jj = 0;
            for (; jj + 1 < n4; jj += 2) {
                a0 = fmaf(Wih[(size_t)jj * in_l + kk], d[jj], a0);
                a1 = fmaf(Wih[(size_t)(jj + 1) * in_l + kk], d[jj + 1], a1);
            }
            dst[idx] = a0 + a1;
        }
        __syncthreads();
    }
    /* dL/dh_k rows through the Linear (the action columns get a gradient nobody uses) */
    const float* __restrict__ Wl = p.blob + hh.lin_w;
    float* dx = lt_field(p, hd, k, hh.dx);
    for (int idx = tid; idx < B * Sd; idx += nt) {
        const int t = idx / Sd, kk = idx - t * Sd;
        const float* d = du + (size_t)t * H;
        float a0 = 0.0f, a1 = 0.0f;
        int j = 0;
        for (; j + 1 < H; j += 2) {
            a0 = fmaf(Wl[(size_t)j * in + kk], d[j], a0);
            a1 = fmaf(Wl[(size_t)(j + 1) * in + kk], d[j + 1], a1);
        }
        if (j < H) a0 = fmaf(Wl[(size_t)j * in + kk], d[j], a0);
        dx[idx] = a0 + a1;
    }
    __syncthreads();
}

/* transposed copies of the forward weights (coalesced over the output neuron): blockIdx.y = head */
__global__ void __launch_bounds__(256) k_lstm_prep(const LstmTrainParams p) {
    const LstmTrainHead& hh = p.head[blockIdx.y];
    const int H = p.H, n = hh.n, in = hh.in_dim, n4 = 4 * n;
    const int stride = gridDim.x * blockDim.x, i0 = blockIdx.x * blockDim.x + threadIdx.x;
    for (int idx = i0; idx < H * in; idx += stride) {
        const int j = idx / in, kk = idx - j * in;
        p.tape[hh.wt_lin + (size_t)kk * H + j] = p.blob[hh.lin_w + idx];
    }
    for (int l = 0; l < p.L; ++l) {
        const int in_l = l == 0 ? H : n;
        for (int idx = i0; idx < n4 * in_l; idx += stride) {
            const int jj = idx / in_l, kk = idx - jj * in_l;
            p.tape[hh.wt_ih[l] + (size_t)kk * n4 + jj] = p.blob[hh.w_ih[l] + idx];
        }
    }
}

/* h_0 = representation(obs), then the next-state calls st_0 .. st_{K-2} one after the other */
__global__ void __launch_bounds__(kLstmTrainThreads) k_lstm_chain_fwd(const LstmTrainParams p) {
    extern __shared__ __align__(16) float lstm_lds[];
    const int B = p.B, Sd = p.Sd, ob = p.obs_dim;
    float* h0 = lt_hstate(p, 0);
    for (int idx = threadIdx.x; idx < B * Sd; idx += blockDim.x) {
        const int t = idx / Sd, j = idx - t * Sd;
        float acc = p.blob[p.r_b + j];
        for (int kk = 0; kk < ob; ++kk) acc = fmaf(p.blob[p.r_w + (size_t)j * ob + kk], p.obs[(size_t)t * ob + kk], acc);
        h0[idx] = acc;
    }
    __syncthreads();
    for (int k = 0; k + 1 < p.K; ++k) lstm_head_fwd(p, kHeadSt, k, lstm_lds);
}

/* blockIdx.x: [0, K) value_k, [K, 2K) policy_k, [2K, 3K-1) reward of dynamics call k (reward_{k+1}).
 * Forward, the cross-entropy terms of these outputs with their gradients, backward. */
__global__ void __launch_bounds__(kLstmTrainThreads) k_lstm_heads(const LstmTrainParams p) {
    extern __shared__ __align__(16) float lstm_lds[];
    const int K = p.K, B = p.B, S = p.Sd, A = p.A, n_terms = 3 * K - 1;
    const int bx = blockIdx.x;
    const int hd = bx < K ? kHeadVal : (bx < 2 * K ? kHeadPol : kHeadRew);
    const int k = bx < K ? bx : (bx < 2 * K ? bx - K : bx - 2 * K);
    lstm_head_fwd(p, hd, k, lstm_lds);
    const int kt = hd == kHeadRew ? k + 1 : k; /* unroll position of the loss term */
    const int t0 = kt == 0 ? 0 : 3 * kt - 1;
    const float* out = lt_layer_h(p, hd, k, p.L - 1);
    float* dout = lt_field(p, hd, k, p.head[hd].dout);
    const int lane = lane_id(), wave = threadIdx.x / kWave, n_waves = blockDim.x / kWave;
    for (int t = wave; t < B; t += n_waves) {
        const float wgt = (p.is_ratio ? p.is_ratio[t] : 1.0f) * p.inv_batch;
        const float gs = (kt > 0 ? 1.0f / (float)K : 1.0f) * wgt;
        float sm;
        if (hd == kHeadPol) {
            const float ptar = lane < A ? p.tp[((size_t)kt * B + t) * A + lane] : 0.0f;
            const float lp = train_ce(out + (size_t)t * A, A, ptar, gs, dout + (size_t)t * A, &sm, lane);
            if (lane == 0) p.loss_buf[(size_t)t * n_terms + t0 + 1] = lp;
        } else {
            const float target = hd == kHeadVal ? p.tv[(size_t)kt * B + t] : p.tr[(size_t)kt * B + t];
            int idx;
            float wt;
            train_support_target(target, S, &idx, &wt);
            const float lv = train_ce(out + (size_t)t * S, S, lane == idx ? wt : 0.0f, gs, dout + (size_t)t * S, &sm, lane);
            if (lane == 0) p.loss_buf[(size_t)t * n_terms + t0 + (hd == kHeadVal ? 0 : 2)] = lv;
            if (hd == kHeadVal) { /* new_priority (muzero_model.py:710-713) */
                const float y = train_sum(lane < S ? (float)(lane - S / 2) * sm : 0.0f);
                if (lane == 0) p.priority[(size_t)kt * B + t] = fabsf(mzs_inverse_value_transform(y) - target);
            }
        }
    }
    __syncthreads();
    lstm_head_bwd(p, hd, k, dout, lstm_lds);
}

/* total gradient of h_k, halved for k > 0 (register_hook, muzero_model.py:640), then the
 * next-state call that produced h_k, for k = K-1 .. 1; finally the gradient of h_0 */
__global__ void __launch_bounds__(kLstmTrainThreads) k_lstm_chain_bwd(const LstmTrainParams p) {
    extern __shared__ __align__(16) float lstm_lds[];
    const int K = p.K, B = p.B, Sd = p.Sd;
    for (int k = K - 1; k >= 0; --k) {
        float* gh = lt_gh(p, k);
        const float* dv = lt_field(p, kHeadVal, k, p.head[kHeadVal].dx);
        const float* dp = lt_field(p, kHeadPol, k, p.head[kHeadPol].dx);
        const float* dr = k + 1 < K ? lt_field(p, kHeadRew, k, p.head[kHeadRew].dx) : nullptr;
        const float* ds = k + 1 < K ? lt_field(p, kHeadSt, k, p.head[kHeadSt].dx) : nullptr;
        const float scale = k > 0 ? 0.5f : 1.0f;
        for (int idx = threadIdx.x; idx < B * Sd; idx += blockDim.x) {
            float v = dv[idx] + dp[idx];
            if (dr) v += dr[idx] + ds[idx];
            gh[idx] = v * scale;
        }
        __syncthreads();
        if (k > 0) lstm_head_bwd(p, kHeadSt, k - 1, gh, lstm_lds);
    }
}

constexpr int kLstmWgradRows = 16; /* batch rows per partial sum */

/* pass 1: thread (blob element, row chunk) = sum over the chunk's rows and over the tensor's calls
 * of delta * input -> partial[chunk][element] */
__global__ void __launch_bounds__(256) k_lstm_wgrad_partial(const LstmTrainParams p, double* partial) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.n_params) return;
    const LstmGradDesc d = p.desc[p.param_desc[i]];
    const int rel = i - d.w_off;
    const int r = d.n_cols ? rel / d.ld : rel, c = d.n_cols ? rel - r * d.ld : 0;
    const int B = p.B;
    const int b0 = blockIdx.y * kLstmWgradRows, b1 = min(b0 + kLstmWgradRows, B);
    const float* xsrc = d.x_src == 0 ? p.tape : (d.x_src == 1 ? p.obs : p.actions);
    const long long xs = d.x_src == 2 ? (long long)B * p.A : d.x_stride;
    double acc = 0.0;
    for (int call = 0; call < d.n_calls; ++call) {
        const float* dl = p.tape + d.d_base + call * d.d_stride + r;
        const float* x = xsrc + d.x_base + call * xs + c;
        for (int t = max(b0, d.t_shift); t < b1; ++t) {
            const float xv = d.n_cols ? x[(size_t)(t - d.t_shift) * d.x_ld] : 1.0f;
            acc += (double)dl[(size_t)t * d.d_ld] * (double)xv;
        }
    }
    partial[(size_t)blockIdx.y * p.n_params + i] = acc;
}

/* pass 2: chunk partials in order (+ l2); the last block reduces the loss terms */
__global__ void __launch_bounds__(256) k_lstm_wgrad(const LstmTrainParams p, const double* partial, int n_chunks,
                                                    int n_grad_blocks) {
    const int K = p.K, B = p.B, n_terms = 3 * K - 1;
    if ((int)blockIdx.x < n_grad_blocks) {
        const int i = blockIdx.x * blockDim.x + threadIdx.x;
        if (i >= p.n_params) return;
        double acc = 0.0;
        for (int c = 0; c < n_chunks; ++c) acc += partial[(size_t)c * p.n_params + i];
        if (p.desc[p.param_desc[i]].l2) {
            double wsum = 0.0;
            for (int b = 0; b < B; ++b) wsum += (double)((p.is_ratio ? p.is_ratio[b] : 1.0f) * p.inv_batch);
            acc += 2.0 * 1e-4 * wsum * (double)p.blob[i];
        }
        p.grad[i] = (float)acc;
        return;
    }
    __shared__ double red[256];
    double sq = 0.0;
    for (int i = threadIdx.x; i < p.n_params; i += blockDim.x)
        if (p.desc[p.param_desc[i]].l2) sq += (double)p.blob[i] * p.blob[i];
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
    for (int q = threadIdx.x; q < n_terms; q += blockDim.x) {
        double s = 0.0;
        for (int b = 0; b < B; ++b) s += (double)p.loss_buf[(size_t)b * n_terms + q];
        p.loss_out[q] = (float)(s * (double)p.inv_batch);
    }
    if (threadIdx.x == 0) p.loss_out[n_terms] = (float)(tot + wsum * 1e-4 * sq);
}

}  // namespace mzs
#endif
