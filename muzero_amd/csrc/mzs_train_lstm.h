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
 *     their transposes in the backward pass): mini-GEMMs of the whole workgroup, rows staged
 *     through LDS, 8 rows x 1 column per lane (lstm_block_gemm);
 * and what is serial is the next-state chain st_0 -> st_1 -> ... (one workgroup walks it forward,
 * k_lstm_chain_fwd, and backward, k_lstm_chain_bwd) and, inside every call, the B-step
 * recurrence.  Waves 0..3 of the workgroup are the four gates there: wave q keeps its gate's rows
 * of W_hh in registers (lane = unit), so a step is 32 FMAs per lane, the gate activation, and ONE
 * LDS-only barrier to swap gates (forward) or partial W_hh^T dz sums (backward) through LDS; rows
 * are prefetched a register chunk ahead so that no global-memory wait sits inside a step.
 * Weight gradients are a separate, embarrassingly parallel pass over the stored deltas
 * (k_lstm_wgrad_partial / k_lstm_wgrad), fixed summation order, no atomics.
 */
#ifndef MZS_TRAIN_LSTM_H
#define MZS_TRAIN_LSTM_H

#include "mzs_train.h"

namespace mzs {

constexpr int kLstmTrainThreads = 1024; /* 16 waves for the mini-GEMMs; waves 0..3 = the four gates of the recurrence */
enum { kHeadRew = 0, kHeadSt = 1, kHeadPol = 2, kHeadVal = 3 };

struct LstmTrainHead {
    int lin_w, lin_b;                                                       /* blob offsets   */
    int w_ih[kLstmMaxLayers], w_hh[kLstmMaxLayers], b_ih[kLstmMaxLayers], b_hh[kLstmMaxLayers];
    int in_dim, n;          /* Linear input width (S_dim or S_dim + A), LSTM width            */
    long long base, stride; /* tape offset of call 0, floats between calls                    */
    long long wt_lin, wt_ih[kLstmMaxLayers]; /* tape offsets of W_lin^T [in][H], W_ih^T [in_l][4n] */
    /* field offsets inside one call's block (rows of max_batch capacity) */
    int u, du, dx, dout;
    int ca[kLstmMaxLayers]; /* [B][n] backward coefficient o (1 - tanh^2 c) of the layer's rows */
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
    int loss_value, loss_reward; /* MZS_LOSS_* (mzs_train_set_loss) */
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

/* The barrier inside the recurrence loops: what is exchanged is in LDS, so wait for LDS only.
 * (__syncthreads() is also a workgroup-scope fence on global memory: it would drain the step's
 * global stores and the prefetch of the next row, a full memory round trip per step.) */
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

/* Tape pointers inside the (non-inlined) recurrence functions, tagged as global memory: through a
 * generic pointer every access would be a FLAT operation, which also counts as an LDS operation,
 * and the LDS-only wait before each barrier would then wait for global memory after all. */
typedef __attribute__((address_space(1))) float gfloat;
typedef __attribute__((address_space(1))) const float gcfloat;
__device__ __forceinline__ gfloat* as_global(float* p) { return (gfloat*)p; }
__device__ __forceinline__ gcfloat* as_global(const float* p) { return (gcfloat*)p; }

/* LDS of one workgroup: the row tile of the mini-GEMMs, then the exchange buffers of the recurrence */
constexpr int kLstmGemmLdsFloats = 12288;
constexpr int kLstmRecurLdsFloats = 1024;
constexpr int kLstmTrainLdsBytes = (kLstmGemmLdsFloats + kLstmRecurLdsFloats) * 4;
constexpr int kLstmGemmRows = 8; /* rows per lane in the mini-GEMM */

/* rows of a mini-GEMM's left operand */
struct LstmRows {
    const float* a;
    int lda;
    __device__ __forceinline__ const float* ptr(int t, int k) const { return a + (size_t)t * lda + k; }
};

/* C[t][j] = bias0[j] + bias1[j] + sum_k X[t][k] * Wt[k * ldw + j]   (t < B, j < N, k < Kd)
 * X rows are staged through LDS a tile at a time; a lane owns column j of 8 rows, so one weight
 * (coalesced over j) feeds 8 FMAs and the X operands are 128-bit broadcast reads.  Called by the whole
 * workgroup; every thread must have finished writing X before the call returns from its first barrier. */
__device__ void lstm_block_gemm(const LstmRows X, int B, int Kd, const float* __restrict__ Wt, int ldw, int N,
                                const float* __restrict__ bias0, const float* __restrict__ bias1, float* C, int ldc,
                                float* xl) {
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & (kWave - 1), wave = tid / kWave, n_waves = nt / kWave;
    const int Kp = (Kd + 3) & ~3;
    int RB = (kLstmGemmLdsFloats / Kp) & ~(kLstmGemmRows - 1);
    RB = RB > 128 ? 128 : RB;
    const int ncg = (N + kWave - 1) / kWave;
    for (int r0 = 0; r0 < B; r0 += RB) {
        const int rows = B - r0 < RB ? B - r0 : RB;
        __syncthreads(); /* X is complete; the previous tile has been consumed */
        for (int base = tid; base < RB * Kp; base += 4 * nt) { /* four loads in flight per lane */
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = base + i * nt, r = idx / Kp, k = idx - r * Kp;
                const bool ok = r < rows && k < Kd; /* rows past the tile's end also fail r < rows */
                const float x = *X.ptr(ok ? r0 + r : r0, ok ? k : 0);
                v[i] = ok ? x : 0.0f;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (base + i * nt < RB * Kp) xl[base + i * nt] = v[i];
        }
        __syncthreads();
        const int n_task = ncg * ((rows + kLstmGemmRows - 1) / kLstmGemmRows);
        for (int task = wave; task < n_task; task += n_waves) {
            const int cg = task % ncg, rt = task / ncg;
            const int j = cg * kWave + lane, jj = j < N ? j : N - 1;
            const float b = (bias0 ? bias0[jj] : 0.0f) + (bias1 ? bias1[jj] : 0.0f);
            float acc[kLstmGemmRows];
#pragma unroll
            for (int i = 0; i < kLstmGemmRows; ++i) acc[i] = b;
            const float* xr = xl + (size_t)rt * kLstmGemmRows * Kp;
            const float* __restrict__ wc = Wt + jj;
#pragma unroll 2
            for (int k = 0; k < Kp; k += 4) {
                /* rows past Kd meet zero X columns: any finite weight will do, stay inside the tensor */
                const int k1 = k + 1 < Kd ? k + 1 : Kd - 1, k2 = k + 2 < Kd ? k + 2 : Kd - 1,
                          k3 = k + 3 < Kd ? k + 3 : Kd - 1;
                const float w0 = wc[(size_t)k * ldw], w1 = wc[(size_t)k1 * ldw], w2 = wc[(size_t)k2 * ldw],
                            w3 = wc[(size_t)k3 * ldw];
#pragma unroll
                for (int i = 0; i < kLstmGemmRows; ++i) {
                    const float4 x = *reinterpret_cast<const float4*>(xr + (size_t)i * Kp + k);
                    acc[i] = fmaf(w0, x.x, acc[i]);
                    acc[i] = fmaf(w1, x.y, acc[i]);
                    acc[i] = fmaf(w2, x.z, acc[i]);
                    acc[i] = fmaf(w3, x.w, acc[i]);
                }
            }
            if (j < N)
#pragma unroll
                for (int i = 0; i < kLstmGemmRows; ++i) {
                    const int r = rt * kLstmGemmRows + i;
                    if (r < rows) C[(size_t)(r0 + r) * ldc + j] = acc[i];
                }
        }
    }
    __syncthreads();
}

/* The forward recurrence of one layer along the batch axis.  Waves 0..3 = gates i, f, g, o (torch
 * row order), lane = unit; the wave's W_hh rows sit in NK registers per lane.  g [B][4n] holds the
 * input half of the pre-activations on entry and the activated gates on return.  One LDS-only
 * barrier per step: the gates are exchanged through a double-buffered LDS row, h is kept in a
 * private LDS row per wave.  The input halves arrive in register chunks of 8 rows, loaded one chunk
 * ahead, and every store is unconditional (lanes past n repeat lane n-1 and all four waves hold the
 * same c and h: identical values to identical addresses), so the step is straight-line code and no
 * global-memory wait sits inside it.  Waves 4.. only keep the barrier count. */
template <int NK>
__device__ __noinline__ void lstm_recur_fwd(const float* __restrict__ Whh, int n, int B, float* g, float* cb, float* hb,
                               float* ls) {
    const int q = threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1), j = lane < n ? lane : n - 1;
    if (q >= 4) {
        for (int t = 0; t < B; ++t) lds_barrier();
        __syncthreads();
        return;
    }
    const bool act = lane < n;
    const int n4 = 4 * n;
    float w[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) w[k] = k < n ? Whh[(size_t)(q * n + j) * n + k] : 0.0f;
    float* hbuf = ls + q * kWave;  /* [4][64] */
    float* gact = ls + 4 * kWave;  /* [2][4][64] */
    hbuf[lane] = 0.0f;
    wave_sync();
    /* tanh(x) = 2 sigmoid(2x) - 1: one branch-free activation for all four gates */
    const float a_in = q == 2 ? 2.0f : 1.0f, a_mul = q == 2 ? 2.0f : 1.0f, a_add = q == 2 ? -1.0f : 0.0f;
    float cprev = 0.0f;
    gfloat* gq = as_global(g + q * n + j); /* this lane's gate entry of row t: gq[t * n4] */
    gfloat* cbj = as_global(cb + j);
    gfloat* hbj = as_global(hb + j);
    const auto step = [&](int t, float z) {
        float z0 = z, z1 = 0.0f;
#pragma unroll
        for (int k = 0; k < NK; k += 4) {
            const float4 hv = *reinterpret_cast<const float4*>(hbuf + k);
            z0 = fmaf(w[k], hv.x, z0);
            z1 = fmaf(w[k + 1], hv.y, z1);
            z0 = fmaf(w[k + 2], hv.z, z0);
            z1 = fmaf(w[k + 3], hv.w, z1);
        }
        const float a = fmaf(lt_sigmoid(a_in * (z0 + z1)), a_mul, a_add);
        float* ga = gact + (t & 1) * 4 * kWave;
        ga[q * kWave + lane] = a;
        gq[(size_t)t * n4] = a;
        lds_barrier();
        const float ig = ga[lane], fg = ga[kWave + lane], gg = ga[2 * kWave + lane], og = ga[3 * kWave + lane];
        const float cv = fg * cprev + ig * gg;
        const float hv = og * lt_tanh(cv);
        hbuf[lane] = act ? hv : 0.0f;
        cbj[(size_t)t * n] = cv;
        hbj[(size_t)t * n] = hv;
        wave_sync();
        cprev = cv;
    };
    constexpr int CH = 8;
    float zc[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) zc[i] = gq[(size_t)(i < B ? i : B - 1) * n4];
    int t0 = 0;
    for (; t0 + CH <= B; t0 += CH) {
        float zn[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int tn = t0 + CH + i;
            zn[i] = gq[(size_t)(tn < B ? tn : B - 1) * n4];
        }
#pragma unroll
        for (int i = 0; i < CH; ++i) step(t0 + i, zc[i]);
#pragma unroll
        for (int i = 0; i < CH; ++i) zc[i] = zn[i];
    }
#pragma unroll
    for (int i = 0; i < CH; ++i)
        if (t0 + i < B) step(t0 + i, zc[i]);
    __syncthreads();
}

/* Everything of the backward cell step that does not depend on the incoming gradient, for every row
 * at once: A = o (1 - tanh^2 c) to ca, and the four factors P_q that turn dL/dc (dL/dh for the output
 * gate) into the pre-activation gradient of gate q, written where dz will go. */
__device__ void lstm_bwd_coefficients(int n, int B, const float* g, const float* cb, float* ca, float* dz) {
    const int n4 = 4 * n;
    for (int idx = threadIdx.x; idx < B * n; idx += blockDim.x) {
        const int t = idx / n, j = idx - t * n;
        const float* gt = g + (size_t)t * n4;
        const float ig = gt[j], fg = gt[n + j], gg = gt[2 * n + j], og = gt[3 * n + j];
        const float cp = t > 0 ? cb[idx - n] : 0.0f;
        const float tc = lt_tanh(cb[idx]);
        ca[idx] = og * (1.0f - tc * tc);
        float* d = dz + (size_t)t * n4;
        d[j] = gg * ig * (1.0f - ig);
        d[n + j] = cp * fg * (1.0f - fg);
        d[2 * n + j] = ig * (1.0f - gg * gg);
        d[3 * n + j] = tc * og * (1.0f - og);
    }
    __syncthreads();
}

/* The backward recurrence of one layer (after lstm_bwd_coefficients).  Every wave 0..3 carries
 * dL/dc along (dc = dh A + dc_next f), turns it into the pre-activation gradient of ITS gate
 * (d = P_q dc, P_o dh for the output gate) and sums its gate's share of W_hh^T dz (lane = unit of
 * h_{t-1}, the wave's W_hh rows in registers, column `lane`); the four shares meet through a
 * double-buffered LDS row: one LDS-only barrier per step.  Rows arrive in register chunks of 4, one
 * chunk ahead; stores are unconditional as in the forward loop. */
template <int NK>
__device__ __noinline__ void lstm_recur_bwd(const float* __restrict__ Whh, int n, int B, const float* g, const float* ca,
                               const float* dO, float* dz, float* ls) {
    const int q = threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1), j = lane < n ? lane : n - 1;
    if (q >= 4) {
        for (int t = 0; t < B; ++t) lds_barrier();
        __syncthreads();
        return;
    }
    const bool act = lane < n;
    const int n4 = 4 * n;
    float w[NK];
#pragma unroll
    for (int r = 0; r < NK; ++r) w[r] = r < n ? Whh[(size_t)(q * n + r) * n + j] : 0.0f;
    float* dzb = ls + q * kWave;   /* [4][64]: this wave's dz row */
    float* part = ls + 4 * kWave;  /* [2][4][64] */
    dzb[lane] = 0.0f;
    float dcn = 0.0f, dhr = 0.0f;
    gcfloat* fq = as_global(g + n + j);   /* forget gate of row t: fq[t * n4] */
    gfloat* dq = as_global(dz + q * n + j); /* P_q on entry, dz on return: dq[t * n4] */
    gcfloat* dOj = as_global(dO + j);
    gcfloat* caj = as_global(ca + j);
    struct Row {
        float dov, a, f, p;
    };
    const auto load = [&](int t) {
        const int tt = t > 0 ? t : 0;
        Row r;
        r.dov = dOj[(size_t)tt * n], r.a = caj[(size_t)tt * n], r.f = fq[(size_t)tt * n4], r.p = dq[(size_t)tt * n4];
        return r;
    };
    const auto step = [&](int t, const Row r) {
        const float dh = r.dov + dhr;
        const float dc = dh * r.a + dcn;
        dcn = dc * r.f;
        const float d = (q == 3 ? dh : dc) * r.p;
        dq[(size_t)t * n4] = d;
        dzb[lane] = act ? d : 0.0f;
        wave_sync();
        float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
        for (int k = 0; k < NK; k += 4) {
            const float4 e = *reinterpret_cast<const float4*>(dzb + k);
            a0 = fmaf(w[k], e.x, a0);
            a1 = fmaf(w[k + 1], e.y, a1);
            a0 = fmaf(w[k + 2], e.z, a0);
            a1 = fmaf(w[k + 3], e.w, a1);
        }
        float* pp = part + (t & 1) * 4 * kWave;
        pp[q * kWave + lane] = a0 + a1;
        lds_barrier();
        dhr = (pp[lane] + pp[kWave + lane]) + (pp[2 * kWave + lane] + pp[3 * kWave + lane]);
    };
    constexpr int CH = 4;
    Row rc[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) rc[i] = load(B - 1 - i);
    int t1 = B; /* rows [0, t1) are still to do; rc[i] = row t1 - 1 - i */
    for (; t1 >= CH; t1 -= CH) {
        Row rn[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) rn[i] = load(t1 - CH - 1 - i);
#pragma unroll
        for (int i = 0; i < CH; ++i) step(t1 - 1 - i, rc[i]);
#pragma unroll
        for (int i = 0; i < CH; ++i) rc[i] = rn[i];
    }
#pragma unroll
    for (int i = 0; i < CH; ++i)
        if (t1 - 1 - i >= 0) step(t1 - 1 - i, rc[i]);
    __syncthreads();
}

/* Forward of call k of head hd: rows x_t = (h_k[t], action_k[t]) -> u, gates, c, h on the tape. */
__device__ void lstm_head_fwd(const LstmTrainParams& p, int hd, int k, float* lds) {
    const LstmTrainHead& hh = p.head[hd];
    const int B = p.B, H = p.H, n = hh.n, Sd = p.Sd, A = p.A, L = p.L;
    float* xl = lds;
    float* ls = lds + kLstmGemmLdsFloats;
    float* u = lt_field(p, hd, k, hh.u);
    {
        const LstmRows X{lt_hstate(p, k), Sd};
        const float* __restrict__ WT = p.tape + hh.wt_lin;
        lstm_block_gemm(X, B, Sd, WT, H, H, p.blob + hh.lin_b, nullptr, u, H, xl);
        if (hd < 2) { /* the action columns of the dynamics input (one-hot, or zero for absorbing rows) */
            const float* __restrict__ xa = p.actions + (size_t)k * B * A;
            for (int idx = threadIdx.x; idx < B * H; idx += blockDim.x) {
                const int t = idx / H, j = idx - t * H;
                float acc = u[idx];
                for (int a = 0; a < A; ++a) acc = fmaf(WT[(size_t)(Sd + a) * H + j], xa[(size_t)t * A + a], acc);
                u[idx] = acc;
            }
        }
    }
    for (int l = 0; l < L; ++l) {
        const int in_l = l == 0 ? H : n;
        float* g = lt_field(p, hd, k, hh.gates[l]);
        /* input half of the gate pre-activations for every row at once */
        const LstmRows X{l == 0 ? u : lt_layer_h(p, hd, k, l - 1), in_l};
        lstm_block_gemm(X, B, in_l, p.tape + hh.wt_ih[l], 4 * n, 4 * n, p.blob + hh.b_ih[l], p.blob + hh.b_hh[l], g,
                        4 * n, xl);
        float* cb = lt_field(p, hd, k, hh.c[l]);
        float* hb = lt_layer_h(p, hd, k, l);
        if (n <= 32)
            lstm_recur_fwd<32>(p.blob + hh.w_hh[l], n, B, g, cb, hb, ls);
        else
            lstm_recur_fwd<64>(p.blob + hh.w_hh[l], n, B, g, cb, hb, ls);
    }
}

/* Backward of call k of head hd.  dOut [B][n] = dL/d(output rows).  Leaves on the tape: dz of every
 * layer, du, and dx [B][Sd] = dL/d(h_k rows) through this call. */
__device__ void lstm_head_bwd(const LstmTrainParams& p, int hd, int k, const float* dOut, float* lds) {
    const LstmTrainHead& hh = p.head[hd];
    const int B = p.B, H = p.H, n = hh.n, Sd = p.Sd, L = p.L;
    float* xl = lds;
    float* ls = lds + kLstmGemmLdsFloats;
    float* du = lt_field(p, hd, k, hh.du);
    __syncthreads(); /* dOut is complete */
    for (int l = L - 1; l >= 0; --l) {
        const int in_l = l == 0 ? H : n;
        const float* g = lt_field(p, hd, k, hh.gates[l]);
        const float* cb = lt_field(p, hd, k, hh.c[l]);
        float* dz = lt_field(p, hd, k, hh.dz[l]);
        const float* dO = l == L - 1 ? dOut : lt_field(p, hd, k, hh.dmid[l]);
        float* ca = lt_field(p, hd, k, hh.ca[l]);
        lstm_bwd_coefficients(n, B, g, cb, ca, dz);
        if (n <= 32)
            lstm_recur_bwd<32>(p.blob + hh.w_hh[l], n, B, g, ca, dO, dz, ls);
        else
            lstm_recur_bwd<64>(p.blob + hh.w_hh[l], n, B, g, ca, dO, dz, ls);
        /* dL/d(layer input rows) = dz W_ih, every row at once (W_ih [4n][in_l] is its own "transposed" form) */
        const LstmRows X{dz, 4 * n};
        lstm_block_gemm(X, B, 4 * n, p.blob + hh.w_ih[l], in_l, in_l, nullptr, nullptr,
                        l == 0 ? du : lt_field(p, hd, k, hh.dmid[l - 1]), in_l, xl);
    }
    /* dL/dh_k rows through the Linear (the action columns get a gradient nobody uses) */
    const LstmRows X{du, H};
    lstm_block_gemm(X, B, H, p.blob + hh.lin_w, hh.in_dim, Sd, nullptr, nullptr, lt_field(p, hd, k, hh.dx), Sd, xl);
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
            const float lp = train_ce(out + (size_t)t * A, A, ptar, gs, dout + (size_t)t * A, &sm, lane,
                                      train_policy_mode(p.loss_value));
            if (lane == 0) p.loss_buf[(size_t)t * n_terms + t0 + 1] = lp;
        } else {
            const float target = hd == kHeadVal ? p.tv[(size_t)kt * B + t] : p.tr[(size_t)kt * B + t];
            int idx;
            float wt;
            train_support_target(target, S, &idx, &wt);
            const float lv = train_ce(out + (size_t)t * S, S, lane == idx ? wt : 0.0f, gs, dout + (size_t)t * S, &sm, lane,
                                      hd == kHeadVal ? p.loss_value : p.loss_reward);
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

constexpr int kLstmWgradRows = 8; /* batch rows per partial sum */

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
#pragma unroll 4
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
    for (int di = 0; di < p.n_desc; ++di) { /* the few tensors of the l2 term, one after the other */
        const LstmGradDesc& d = p.desc[di];
        if (!d.l2) continue;
        const int nc = d.n_cols ? d.n_cols : 1;
        for (int e = threadIdx.x; e < d.n_rows * nc; e += blockDim.x) {
            const float wv = p.blob[d.w_off + (e / nc) * d.ld + e % nc];
            sq += (double)wv * wv;
        }
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
    for (int q = threadIdx.x; q < n_terms; q += blockDim.x) {
        double s = 0.0;
        for (int b = 0; b < B; ++b) s += (double)p.loss_buf[(size_t)b * n_terms + q];
        p.loss_out[q] = (float)(s * (double)p.inv_batch);
    }
    if (threadIdx.x == 0) p.loss_out[n_terms] = (float)(tot + wsum * 1e-4 * sq);
}

}  // namespace mzs
#endif
