/*
 * mzs_vision_train.hip — the training step of the reference's 'vision_model' on the GPU (gfx950).
 *
 * Reference code restated (file:line in /root/reference):
 *   muzero_model.py:609-647   compute_forward (representation, prediction, K x (dynamics, prediction), the
 *                             x 0.5 hook on every next hidden state)
 *   muzero_model.py:650-734   evaluate_loss (the criteria of mzs_train.h: train_ce / train_support_target),
 *                             l2 over the zip()-truncated parameter lists, importance weights, mean
 *   neural_network_vision_model.py:41-300  Residual_block / Down_sample / the three networks, in TRAINING
 *                             mode: BatchNorm2d on batch statistics (running statistics updated per
 *                             application, momentum 0.1, unbiased variance), Dropout2d(0.5) per (sample, channel)
 *
 * Shape of the work: channel counts 1..4 on maps 49 x 49 .. 7 x 7 and three small MLP heads — nothing
 * GEMM-shaped (K = 9..36 per output); every layer is HBM/L2-bound elementwise or stencil work over
 * [B][C][S][S] fp32 tensors.  So: one plain kernel per layer and direction, a thread per output element
 * (coalesced along x), per-channel / per-weight reductions as one block each with fp64 accumulators in a
 * fixed order (no atomics: results are reproducible run to run), and a host-side tape that replays the
 * backward launches in reverse.  All launches of a step go to one stream without host synchronisation.
 *
 * Numerics: fp32 storage, fp32 multiply-adds in the stencils, fp64 accumulators in the reductions.
 * Summation orders are NOT those of torch's kernels; the step is checked against the float64 oracle
 * (oracle/mz_vision_train_oracle.py, itself pinned to the reference's Muzero.train) to 1e-5.
 */
#include "mzs_vision_train.h"

#include <cstring>
#include <functional>
#include <vector>

#include "../../include/mzsearch.h"
#include "mzs_train_loss.h"

namespace mzs {
namespace {

constexpr int kT = 256;
constexpr int kMaxUnroll = 32;
constexpr int kHid = MZS_VIS_HIDDEN;
constexpr int kS4 = MZS_VIS_S4 * MZS_VIS_S4; /* 49 */

inline unsigned nblk(size_t n) { return (unsigned)((n + kT - 1) / kT); }

/* reductions over n elements: chunks of >= 1024 elements (four per thread, all requested at once), at most 512 */
struct Chunks {
    int n;
    size_t per;
};
inline Chunks chunks_of(size_t n, size_t min_per = (size_t)kT * 4) {
    size_t per = min_per;
    if ((n + per - 1) / per > 512) per = (n + 511) / 512;
    return Chunks{(int)((n + per - 1) / per), per};
}

/* sum of one fp64 value per lane of a wavefront, fixed order; the total in lane 0 */
__device__ __forceinline__ double wave_sum64(double v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
    return v;
}

/* sum over the kT threads of a block, fixed order; the total in thread 0 only */
__device__ __forceinline__ double block_sum(double v, double* red) {
    v = wave_sum64(v);
    if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x / kWave] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0)
        for (int w = 0; w < kT / kWave; ++w) r += red[w];
    __syncthreads();
    return r;
}

/* the chunks of a two-step reduction, added by one wavefront (lane k takes chunks k, k + 64, ...); stride in
 * doubles between consecutive chunks; the total in lane 0 */
__device__ __forceinline__ double wave_sum_chunks(const double* __restrict__ part, int chunks, size_t stride, int lane) {
    double v = 0.0;
    for (int k = lane; k < chunks; k += kWave) v += part[(size_t)k * stride];
    return wave_sum64(v);
}

/* the second step of a weight-gradient reduction, done by the waves of the NEXT kernel on the stream (the layer's
 * data-gradient kernel) before their own work: item i = the chunks of part[.][i] added into dst[i] */
__device__ __forceinline__ void finish_items(const double* __restrict__ part, int items, int chunks,
                                             float* __restrict__ dst) {
    if (!part) return;
    const int lane = threadIdx.x % kWave;
    for (int i = blockIdx.x * (kT / kWave) + threadIdx.x / kWave; i < items; i += gridDim.x * (kT / kWave)) {
        const double acc = wave_sum_chunks(part + i, chunks, (size_t)items, lane);
        if (lane == 0) dst[i] += (float)acc;
    }
}

constexpr int kU = 4; /* elements a thread of a reduction requests before it adds any */

/* ------------------------------------------------------------------ 3x3 convolution, padding 1, no bias */
__global__ void __launch_bounds__(kT) k_vt_conv3_fwd(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ res, float* __restrict__ y, int B,
                                                     int Ci, int Co, int S, int So, int stride) {
    const size_t i = (size_t)blockIdx.x * kT + threadIdx.x;
    if (i >= (size_t)B * Co * So * So) return;
    const int ox = (int)(i % So), oy = (int)((i / So) % So), co = (int)((i / ((size_t)So * So)) % Co);
    const int b = (int)(i / ((size_t)So * So * Co));
    const float* xb = x + (size_t)b * Ci * S * S;
    const float* wc = w + (size_t)co * Ci * 9;
    float acc = 0.0f;
    for (int ci = 0; ci < Ci; ++ci)
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * stride + ky - 1;
            if (iy < 0 || iy >= S) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * stride + kx - 1;
                if (ix < 0 || ix >= S) continue;
                acc = __builtin_fmaf(wc[ci * 9 + ky * 3 + kx], xb[((size_t)ci * S + iy) * S + ix], acc);
            }
        }
    y[i] = res ? acc + res[i] : acc;
}

__global__ void __launch_bounds__(kT) k_vt_conv3_bwd_x(const float* __restrict__ dy, const float* __restrict__ w,
                                                       float* __restrict__ dx, int B, int Ci, int Co, int S, int So,
                                                       int stride, const double* __restrict__ part, int chunks,
                                                       float* __restrict__ gw) {
    finish_items(part, Co * Ci * 9, chunks, gw); /* of the k_vt_conv3_bwd_w launched before this kernel */
    const size_t i = (size_t)blockIdx.x * kT + threadIdx.x;
    if (i >= (size_t)B * Ci * S * S) return;
    const int ix = (int)(i % S), iy = (int)((i / S) % S), ci = (int)((i / ((size_t)S * S)) % Ci);
    const int b = (int)(i / ((size_t)S * S * Ci));
    const float* dyb = dy + (size_t)b * Co * So * So;
    float acc = 0.0f;
    for (int co = 0; co < Co; ++co)
        for (int ky = 0; ky < 3; ++ky) {
            const int ty = iy + 1 - ky;
            if (ty < 0 || ty % stride) continue;
            const int oy = ty / stride;
            if (oy >= So) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int tx = ix + 1 - kx;
                if (tx < 0 || tx % stride) continue;
                const int ox = tx / stride;
                if (ox >= So) continue;
                acc = __builtin_fmaf(w[((size_t)(co * Ci + ci) * 3 + ky) * 3 + kx],
                                     dyb[((size_t)co * So + oy) * So + ox], acc);
            }
        }
    dx[i] += acc;
}

/* Reductions over (sample, pixel) run in two steps: block (item, chunk) sums its chunk of the n elements into
 * part[chunk][item] (fp64; at most 64 chunks), then ONE WAVEFRONT per item adds the chunks, a lane each, in a
 * fixed shuffle tree — enough blocks to fill the chip (a per-channel or per-weight sum has 1..108 items), a
 * fixed summation order, no atomics.  The BatchNorm sums are finished inside the kernels that consume them. */
__global__ void __launch_bounds__(kT) k_vt_accum_fin(const double* __restrict__ part, int items, int chunks,
                                                     float* __restrict__ dst) {
    const int i = blockIdx.x * (kT / kWave) + threadIdx.x / kWave, lane = threadIdx.x % kWave;
    if (i >= items) return;
    const double acc = wave_sum_chunks(part + i, chunks, (size_t)items, lane);
    if (lane == 0) dst[i] += (float)acc;
}

/* block (weight element, chunk): sum over the chunk's samples and output pixels (small layers: nine times the
 * blocks of the kernel below) */
__global__ void __launch_bounds__(kT) k_vt_conv3_bwd_w1(const float* __restrict__ x, const float* __restrict__ dy,
                                                        double* __restrict__ part, float* __restrict__ gw, int B, int Ci,
                                                        int Co, int S, int So, int stride, size_t per) {
    __shared__ double red[kT];
    const int wi = blockIdx.x;
    const int kx = wi % 3, ky = (wi / 3) % 3, ci = (wi / 9) % Ci, co = wi / (9 * Ci);
    const size_t n_all = (size_t)B * So * So, j0 = (size_t)blockIdx.y * per;
    const size_t n = j0 + per < n_all ? j0 + per : n_all;
    double acc = 0.0;
    for (size_t base = j0 + threadIdx.x; base < n; base += (size_t)kU * kT) {
        float dv[kU], xv[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const size_t j = base + (size_t)u * kT;
            const int ox = (int)(j % So), oy = (int)((j / So) % So), b = (int)(j / ((size_t)So * So));
            const int iy = oy * stride + ky - 1, ix = ox * stride + kx - 1;
            const bool ok = j < n && iy >= 0 && iy < S && ix >= 0 && ix < S;
            const size_t di = ok ? (((size_t)b * Co + co) * So + oy) * So + ox : 0;
            const size_t xi = ok ? (((size_t)b * Ci + ci) * S + iy) * S + ix : 0;
            const float d0 = dy[di], x0 = x[xi];
            dv[u] = ok ? d0 : 0.0f;
            xv[u] = x0;
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) acc += (double)dv[u] * (double)xv[u];
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) {
        if (gridDim.y == 1)
            gw[wi] += (float)acc;
        else
            part[(size_t)blockIdx.y * gridDim.x + wi] = acc;
    }
}

/* block ((co, ci) pair, chunk): the nine taps' sums over the chunk's samples and output pixels — dy is read
 * once for the nine, the nine inputs of an output pixel are neighbours */
__global__ void __launch_bounds__(kT) k_vt_conv3_bwd_w(const float* __restrict__ x, const float* __restrict__ dy,
                                                       double* __restrict__ part, float* __restrict__ gw, int B, int Ci,
                                                       int Co, int S, int So, int stride, size_t per) {
    __shared__ double red[kT];
    const int pair = blockIdx.x, ci = pair % Ci, co = pair / Ci;
    const size_t n_all = (size_t)B * So * So, j0 = (size_t)blockIdx.y * per;
    const size_t n = j0 + per < n_all ? j0 + per : n_all;
    double acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = 0.0;
    for (size_t base = j0 + threadIdx.x; base < n; base += (size_t)kU * kT) {
        float dv[kU], xv[kU][9];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const size_t j = base + (size_t)u * kT;
            const bool ok = j < n;
            const int ox = (int)(j % So), oy = (int)((j / So) % So), b = ok ? (int)(j / ((size_t)So * So)) : 0;
            dv[u] = ok ? dy[(((size_t)b * Co + co) * So + oy) * So + ox] : 0.0f;
            const float* xc = x + ((size_t)b * Ci + ci) * S * S;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int iy = oy * stride + ky - 1, ix = ox * stride + kx - 1;
                    xv[u][ky * 3 + kx] = ok && iy >= 0 && iy < S && ix >= 0 && ix < S ? xc[(size_t)iy * S + ix] : 0.0f;
                }
        }
#pragma unroll
        for (int u = 0; u < kU; ++u)
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[t] += (double)dv[u] * (double)xv[u][t];
    }
    const int items = Co * Ci * 9;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const double r = block_sum(acc[t], red);
        if (threadIdx.x == 0) {
            const int wi = pair * 9 + t;
            if (gridDim.y == 1)
                gw[wi] += (float)r;
            else
                part[(size_t)blockIdx.y * items + wi] = r;
        }
    }
}

/* ------------------------------------------------------------------ BatchNorm2d (training) + relu + Dropout2d */
/* block (channel, chunk) -> part[chunk][2 C] = (sum x, sum x^2) */
__global__ void __launch_bounds__(kT) k_vt_bn_stats(const float* __restrict__ x, double* __restrict__ part, int B,
                                                    int C, int SS, size_t per) {
    __shared__ double red[kT];
    const int c = blockIdx.x;
    const size_t n_all = (size_t)B * SS, j0 = (size_t)blockIdx.y * per;
    const size_t n = j0 + per < n_all ? j0 + per : n_all;
    double s = 0.0, q = 0.0;
    for (size_t base = j0 + threadIdx.x; base < n; base += (size_t)kU * kT) {
        float xv[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const size_t j = base + (size_t)u * kT;
            xv[u] = j < n ? x[((size_t)(j / SS) * C + c) * SS + j % SS] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            s += (double)xv[u];
            q += (double)xv[u] * (double)xv[u];
        }
    }
    s = block_sum(s, red);
    q = block_sum(q, red);
    if (threadIdx.x == 0) {
        part[((size_t)blockIdx.y * C + c) * 2] = s;
        part[((size_t)blockIdx.y * C + c) * 2 + 1] = q;
    }
}
/* the chunks of k_vt_bn_stats -> mean and 1 / sqrt(var + eps) of channel c (wave-uniform result in every lane's
 * return is NOT given: lane 0 holds it) */
__device__ __forceinline__ void bn_finish_stats(const double* __restrict__ part, int chunks, int C, int c, size_t n,
                                                int lane, double* m_out, double* var_out) {
    const double s = wave_sum_chunks(part + 2 * c, chunks, 2 * (size_t)C, lane);
    const double q = wave_sum_chunks(part + 2 * c + 1, chunks, 2 * (size_t)C, lane);
    const double m = s / (double)n;
    const double var = q / (double)n - m * m;
    *m_out = m;
    *var_out = var > 0.0 ? var : 0.0;
}

/* y = relu(gamma xhat + beta) * (mask ? 2 mask : 1).  Every block finishes the batch statistics from the chunk
 * sums itself (its first wave, a lane per chunk: the same numbers in every block); block 0 also keeps them for
 * the backward pass and moves the running statistics (torch: running = 0.9 running + 0.1 stat, unbiased var). */
__global__ void __launch_bounds__(kT) k_vt_bn_act_fwd(const float* __restrict__ x, const double* __restrict__ part,
                                                      int chunks, float* mean, float* rstd, float* run_mean,
                                                      float* run_var, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta,
                                                      const unsigned char* __restrict__ mask, float* __restrict__ y,
                                                      int B, int C, int SS) {
    __shared__ float s_mean[4], s_rstd[4];
    if (threadIdx.x < kWave) {
        const size_t n = (size_t)B * SS;
        for (int c = 0; c < C; ++c) {
            double m, var;
            bn_finish_stats(part, chunks, C, c, n, threadIdx.x, &m, &var);
            if (threadIdx.x == 0) {
                s_mean[c] = (float)m;
                s_rstd[c] = (float)(1.0 / sqrt(var + 1e-5));
                if (blockIdx.x == 0) {
                    mean[c] = s_mean[c];
                    rstd[c] = s_rstd[c];
                    run_mean[c] = 0.9f * run_mean[c] + 0.1f * (float)m;
                    run_var[c] = 0.9f * run_var[c] + 0.1f * (float)(var * ((double)n / ((double)n - 1.0)));
                }
            }
        }
    }
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * kT + threadIdx.x;
    if (i >= (size_t)B * C * SS) return;
    const int c = (int)((i / SS) % C);
    const size_t bc = i / SS;
    const float f = mask ? (mask[bc] ? 2.0f : 0.0f) : 1.0f;
    const float v = gamma[c] * ((x[i] - s_mean[c]) * s_rstd[c]) + beta[c];
    y[i] = v > 0.0f ? v * f : 0.0f;
}

/* per channel: s1 = sum d, s2 = sum d xhat with d = dL/d(bn output); block (channel, chunk) -> part[chunk][2 C] */
__global__ void __launch_bounds__(kT) k_vt_bn_bwd_red(const float* __restrict__ x, const float* __restrict__ mean,
                                                      const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta,
                                                      const unsigned char* __restrict__ mask,
                                                      const float* __restrict__ dy, double* __restrict__ part, int B,
                                                      int C, int SS, size_t per) {
    __shared__ double red[kT];
    const int c = blockIdx.x;
    const size_t n_all = (size_t)B * SS, j0 = (size_t)blockIdx.y * per;
    const size_t n = j0 + per < n_all ? j0 + per : n_all;
    const float mc = mean[c], rc = rstd[c], gc = gamma[c], bc = beta[c];
    double s1 = 0.0, s2 = 0.0;
    for (size_t base = j0 + threadIdx.x; base < n; base += (size_t)kU * kT) {
        float xv[kU], dv[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const size_t j = base + (size_t)u * kT;
            const bool ok = j < n;
            const size_t b = ok ? j / SS : 0, i = (b * C + c) * SS + (ok ? j % SS : 0);
            const float f = !ok ? 0.0f : mask ? (mask[b * C + c] ? 2.0f : 0.0f) : 1.0f;
            xv[u] = x[i];
            dv[u] = dy[i] * f;
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const float xh = (xv[u] - mc) * rc;
            const float d = gc * xh + bc > 0.0f ? dv[u] : 0.0f;
            s1 += (double)d;
            s2 += (double)d * (double)xh;
        }
    }
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    if (threadIdx.x == 0) {
        part[((size_t)blockIdx.y * C + c) * 2] = s1;
        part[((size_t)blockIdx.y * C + c) * 2 + 1] = s2;
    }
}
__global__ void __launch_bounds__(kT) k_vt_bn_bwd_x(const float* __restrict__ x, const float* __restrict__ mean,
                                                    const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta,
                                                    const unsigned char* __restrict__ mask,
                                                    const float* __restrict__ dy, const double* __restrict__ part,
                                                    int chunks, float* g_gamma, float* g_beta,
                                                    float* __restrict__ dx, int B, int C, int SS) {
    /* every block finishes the two per-channel sums from the chunks of k_vt_bn_bwd_red (first wave, a lane per
     * chunk); block 0 adds them to the gradients of gamma and beta */
    __shared__ double sums[8];
    if (threadIdx.x < kWave) {
        for (int c = 0; c < C; ++c) {
            const double s1 = wave_sum_chunks(part + 2 * c, chunks, 2 * (size_t)C, threadIdx.x);
            const double s2 = wave_sum_chunks(part + 2 * c + 1, chunks, 2 * (size_t)C, threadIdx.x);
            if (threadIdx.x == 0) {
                sums[2 * c] = s1;
                sums[2 * c + 1] = s2;
                if (blockIdx.x == 0) {
                    g_gamma[c] += (float)s2;
                    g_beta[c] += (float)s1;
                }
            }
        }
    }
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * kT + threadIdx.x;
    if (i >= (size_t)B * C * SS) return;
    const int c = (int)((i / SS) % C);
    const size_t bc = i / SS;
    const float f = mask ? (mask[bc] ? 2.0f : 0.0f) : 1.0f;
    const float xh = (x[i] - mean[c]) * rstd[c];
    const float d = gamma[c] * xh + beta[c] > 0.0f ? dy[i] * f : 0.0f;
    const double n = (double)B * SS;
    const double g = (double)gamma[c] * (double)rstd[c] / n * (n * (double)d - sums[2 * c] - (double)xh * sums[2 * c + 1]);
    dx[i] += (float)g;
}

/* ------------------------------------------------------------------ small elementwise / pooling layers */
__global__ void __launch_bounds__(kT) k_vt_axpy(float* __restrict__ dst, const float* __restrict__ src, float a,
                                                size_t n) {
    const size_t i = (size_t)blockIdx.x * kT + threadIdx.x;
    if (i < n) dst[i] += a * src[i];
}

__global__ void __launch_bounds__(kT) k_vt_relu_fwd(const float* __restrict__ x, float* __restrict__ y, size_t n) {
    const size_t i = (size_t)blockIdx.x * kT + threadIdx.x;
    if (i < n) y[i] = x[i] > 0.0f ? x[i] : 0.0f;
}
__global__ void __launch_bounds__(kT) k_vt_relu_bwd(const float* __restrict__ x, const float* __restrict__ dy,
                                                    float* __restrict__ dx, size_t n) {
    const size_t i = (size_t)blockIdx.x * kT + threadIdx.x;
    if (i < n && x[i] > 0.0f) dx[i] += dy[i];
}

/* AvgPool2d(3, stride 2, padding 1), count_include_pad: / 9 */
__global__ void __launch_bounds__(kT) k_vt_pool_fwd(const float* __restrict__ x, float* __restrict__ y, int BC, int S,
                                                    int So) {
    const size_t i = (size_t)blockIdx.x * kT + threadIdx.x;
    if (i >= (size_t)BC * So * So) return;
    const int ox = (int)(i % So), oy = (int)((i / So) % So);
    const float* xc = x + (i / ((size_t)So * So)) * S * S;
    float acc = 0.0f;
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = 2 * oy + ky - 1;
        if (iy < 0 || iy >= S) continue;
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = 2 * ox + kx - 1;
            if (ix >= 0 && ix < S) acc += xc[(size_t)iy * S + ix];
        }
    }
    y[i] = acc / 9.0f;
}
__global__ void __launch_bounds__(kT) k_vt_pool_bwd(const float* __restrict__ dy, float* __restrict__ dx, int BC,
                                                    int S, int So) {
    const size_t i = (size_t)blockIdx.x * kT + threadIdx.x;
    if (i >= (size_t)BC * S * S) return;
    const int ix = (int)(i % S), iy = (int)((i / S) % S);
    const float* dc = dy + (i / ((size_t)S * S)) * So * So;
    float acc = 0.0f;
    for (int ky = 0; ky < 3; ++ky) {
        const int ty = iy + 1 - ky;
        if (ty < 0 || (ty & 1) || ty / 2 >= So) continue;
        for (int kx = 0; kx < 3; ++kx) {
            const int tx = ix + 1 - kx;
            if (tx < 0 || (tx & 1) || tx / 2 >= So) continue;
            acc += dc[(size_t)(ty / 2) * So + tx / 2];
        }
    }
    dx[i] += acc / 9.0f;
}

/* torch.cat([hidden, plane], dim=1) on 7 x 7 maps; plane = (argmax(one-hot row) + 1) / A (muzero_model.py:505-516) */
__global__ void __launch_bounds__(kT) k_vt_cat_fwd(const float* __restrict__ h, const float* __restrict__ actions,
                                                   float* __restrict__ y, int B, int A) {
    const size_t i = (size_t)blockIdx.x * kT + threadIdx.x;
    if (i >= (size_t)B * 4 * kS4) return;
    const int p = (int)(i % kS4), c = (int)((i / kS4) % 4), b = (int)(i / (4 * kS4));
    if (c < 3) {
        y[i] = h[((size_t)b * 3 + c) * kS4 + p];
        return;
    }
    int best = 0;
    float bv = actions[(size_t)b * A];
    for (int a = 1; a < A; ++a) {
        const float v = actions[(size_t)b * A + a];
        if (v > bv) bv = v, best = a;
    }
    y[i] = (float)(best + 1) / (float)A;
}
__global__ void __launch_bounds__(kT) k_vt_cat_bwd(const float* __restrict__ dy, float* __restrict__ dh, int B) {
    const size_t i = (size_t)blockIdx.x * kT + threadIdx.x;
    if (i >= (size_t)B * 3 * kS4) return;
    const int p = (int)(i % kS4), c = (int)((i / kS4) % 3), b = (int)(i / (3 * kS4));
    dh[i] += dy[((size_t)b * 4 + c) * kS4 + p];
}

/* 1 x 1 convolution with bias on 7 x 7 maps; the output is the flattened [B][Co * 49] row of the head MLP */
__global__ void __launch_bounds__(kT) k_vt_conv1_fwd(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, float* __restrict__ y, int B,
                                                     int Ci, int Co) {
    const size_t i = (size_t)blockIdx.x * kT + threadIdx.x;
    if (i >= (size_t)B * Co * kS4) return;
    const int p = (int)(i % kS4), co = (int)((i / kS4) % Co), b = (int)(i / ((size_t)Co * kS4));
    float acc = bias[co];
    for (int ci = 0; ci < Ci; ++ci) acc = __builtin_fmaf(w[co * Ci + ci], x[((size_t)b * Ci + ci) * kS4 + p], acc);
    y[i] = acc;
}
__global__ void __launch_bounds__(kT) k_vt_conv1_bwd_x(const float* __restrict__ dy, const float* __restrict__ w,
                                                       float* __restrict__ dx, int B, int Ci, int Co,
                                                       const double* __restrict__ part, int chunks,
                                                       float* __restrict__ gw) {
    finish_items(part, Co * Ci + Co, chunks, gw); /* of the k_vt_conv1_bwd_w launched before this kernel */
    const size_t i = (size_t)blockIdx.x * kT + threadIdx.x;
    if (i >= (size_t)B * Ci * kS4) return;
    const int p = (int)(i % kS4), ci = (int)((i / kS4) % Ci), b = (int)(i / ((size_t)Ci * kS4));
    float acc = 0.0f;
    for (int co = 0; co < Co; ++co) acc = __builtin_fmaf(w[co * Ci + ci], dy[((size_t)b * Co + co) * kS4 + p], acc);
    dx[i] += acc;
}
/* blocks [0, Co Ci): weights; [Co Ci, Co Ci + Co): biases */
__global__ void __launch_bounds__(kT) k_vt_conv1_bwd_w(const float* __restrict__ x, const float* __restrict__ dy,
                                                       double* __restrict__ part, float* __restrict__ gw, int B, int Ci,
                                                       int Co, size_t per) {
    __shared__ double red[kT];
    const int wi = blockIdx.x;
    const bool is_bias = wi >= Co * Ci;
    const int co = is_bias ? wi - Co * Ci : wi / Ci, ci = is_bias ? 0 : wi % Ci;
    const size_t n_all = (size_t)B * kS4, j0 = (size_t)blockIdx.y * per;
    const size_t n = j0 + per < n_all ? j0 + per : n_all;
    double acc = 0.0;
    for (size_t base = j0 + threadIdx.x; base < n; base += (size_t)kU * kT) {
        float dv[kU], xv[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const size_t j = base + (size_t)u * kT;
            const bool ok = j < n;
            const size_t b = ok ? j / kS4 : 0, p = ok ? j % kS4 : 0;
            dv[u] = ok ? dy[(b * Co + co) * kS4 + p] : 0.0f;
            xv[u] = is_bias ? 1.0f : x[(b * Ci + ci) * kS4 + p];
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) acc += (double)dv[u] * (double)xv[u];
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) {
        if (gridDim.y == 1)
            gw[wi] += (float)acc;
        else
            part[(size_t)blockIdx.y * gridDim.x + wi] = acc;
    }
}

/* ------------------------------------------------------------------ Linear (+ relu) */
/* W in the quad layout of the search kernels' store (mzs_vision.h: element (k, n), k < 4 (K / 4), at
 * ((k / 4) N + n) 4 + k % 4, the last K % 4 rows behind as [k][n]): thread (b, n) fetches four k per 16-byte
 * load and neighbouring n touch neighbouring addresses; x[b][.] is the same address for the lanes of a row */
__global__ void __launch_bounds__(kT) k_vt_linear_fwd(const float* __restrict__ x, const float* __restrict__ Wq,
                                                      const float* __restrict__ bias, float* __restrict__ y, int B,
                                                      int N, int K, int relu) {
    const size_t i = (size_t)blockIdx.x * kT + threadIdx.x;
    if (i >= (size_t)B * N) return;
    const int n = (int)(i % N), b = (int)(i / N);
    const float* xr = x + (size_t)b * K;
    const float4* wq = reinterpret_cast<const float4*>(Wq) + n;
    const int Q = K >> 2;
    float e = bias[n], o = 0.0f;
    constexpr int QB = 8; /* quads requested before the chains that consume them */
    int q = 0;
    for (; q + QB <= Q; q += QB) {
        float4 w[QB];
#pragma unroll
        for (int u = 0; u < QB; ++u) w[u] = wq[(size_t)(q + u) * N];
#pragma unroll
        for (int u = 0; u < QB; ++u) {
            e = __builtin_fmaf(w[u].x, xr[4 * (q + u)], e);
            o = __builtin_fmaf(w[u].y, xr[4 * (q + u) + 1], o);
            e = __builtin_fmaf(w[u].z, xr[4 * (q + u) + 2], e);
            o = __builtin_fmaf(w[u].w, xr[4 * (q + u) + 3], o);
        }
    }
    for (; q < Q; ++q) {
        const float4 w = wq[(size_t)q * N];
        e = __builtin_fmaf(w.x, xr[4 * q], e);
        o = __builtin_fmaf(w.y, xr[4 * q + 1], o);
        e = __builtin_fmaf(w.z, xr[4 * q + 2], e);
        o = __builtin_fmaf(w.w, xr[4 * q + 3], o);
    }
    const float* wt = Wq + (size_t)4 * Q * N + n;
    for (int k = 4 * Q; k < K; ++k) e = __builtin_fmaf(wt[(size_t)(k - 4 * Q) * N], xr[k], e);
    const float v = e + o;
    y[i] = relu && v < 0.0f ? 0.0f : v;
}
/* the Linear backward kernels take dz = dy where the relu passed (y > 0), formed on the fly when `relu` */
__global__ void __launch_bounds__(kT) k_vt_linear_bwd_x(const float* __restrict__ dy, const float* __restrict__ y,
                                                        const float* __restrict__ W, float* __restrict__ dx, int B,
                                                        int N, int K, int relu) {
    const size_t i = (size_t)blockIdx.x * kT + threadIdx.x;
    if (i >= (size_t)B * K) return;
    const int k = (int)(i % K), b = (int)(i / K);
    const float* dr = dy + (size_t)b * N;
    const float* yr = y + (size_t)b * N;
    float e = 0.0f, o = 0.0f;
    int n = 0;
    for (; n < N; n += 16) { /* sixteen weight rows in flight */
        float w[16], d[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { /* clamped addresses, selects after the loads: no branch between requests */
            const bool ok = n + u < N;
            const int nn = ok ? n + u : n;
            w[u] = W[(size_t)nn * K + k];
            const float yv = yr[nn], dv = dr[nn];
            d[u] = !ok || (relu && !(yv > 0.0f)) ? 0.0f : dv;
        }
#pragma unroll
        for (int u = 0; u < 16; u += 2) {
            e = __builtin_fmaf(w[u], d[u], e);
            o = __builtin_fmaf(w[u + 1], d[u + 1], o);
        }
    }
    dx[i] += e + o;
}
/* threads [0, N K): weights; [N K, N K + N): biases; blockIdx.y = chunk of `per` samples -> part[chunk][item],
 * or straight into the gradient when there is one chunk */
__global__ void __launch_bounds__(kT) k_vt_linear_bwd_w(const float* __restrict__ x, const float* __restrict__ dy,
                                                        const float* __restrict__ y, double* __restrict__ part,
                                                        float* __restrict__ gw, int B, int N, int K, int per,
                                                        int relu) {
    const size_t i = (size_t)blockIdx.x * kT + threadIdx.x, items = (size_t)N * K + N;
    if (i >= items) return;
    const int b0 = blockIdx.y * per, b1 = min(b0 + per, B);
    const bool is_w = i < (size_t)N * K;
    const int k = is_w ? (int)(i % K) : 0, n = is_w ? (int)(i / K) : (int)(i - (size_t)N * K);
    double acc = 0.0;
    for (int b = b0; b < b1; b += 8) { /* eight samples in flight */
        float d[8], xv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool ok = b + u < b1;
            const size_t bb = ok ? (size_t)(b + u) : (size_t)b0;
            d[u] = !ok || (relu && !(y[bb * N + n] > 0.0f)) ? 0.0f : dy[bb * N + n];
            xv[u] = is_w ? x[bb * K + k] : 1.0f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += (double)d[u] * (double)xv[u];
    }
    if (gridDim.y == 1)
        gw[i] += (float)acc;
    else
        part[(size_t)blockIdx.y * items + i] = acc;
}

/* ------------------------------------------------------------------ losses (muzero_model.py:650-713) */
struct LossParams {
    const float* val[kMaxUnroll];
    const float* pol[kMaxUnroll];
    const float* rew[kMaxUnroll]; /* rew[0] unused */
    float* gval[kMaxUnroll];
    float* gpol[kMaxUnroll];
    float* grew[kMaxUnroll];
    const float* tv;
    const float* tr;
    const float* tp;
    const float* is_ratio;
    float* loss_buf; /* [B][3K-1] */
    float* priority; /* [K][B]    */
    int B, K, S, A, loss_value, loss_reward;
    float inv_batch;
};

/* one wavefront per (unroll step, sample): the criteria of k_train_sample on the vision heads' logits */
__global__ void __launch_bounds__(kWave) k_vt_loss(const LossParams p) {
    const int lane = lane_id();
    const int k = blockIdx.x / p.B, b = blockIdx.x % p.B;
    const int S = p.S, A = p.A, K = p.K, B = p.B, n_terms = 3 * K - 1;
    const float wgt = (p.is_ratio ? p.is_ratio[b] : 1.0f) * p.inv_batch;
    const float gs = (k > 0 ? 1.0f / (float)K : 1.0f) * wgt;
    const int t0 = k == 0 ? 0 : 3 * k - 1;
    int idx;
    float wt, sm;
    const float tvk = p.tv[(size_t)k * B + b];
    train_support_target(tvk, S, &idx, &wt);
    const float lv = train_ce(p.val[k] + (size_t)b * S, S, lane == idx ? wt : 0.0f, gs, p.gval[k] + (size_t)b * S, &sm,
                              lane, p.loss_value);
    const float y = train_sum(lane < S ? (float)(lane - S / 2) * sm : 0.0f);
    if (lane == 0) p.priority[(size_t)k * B + b] = fabsf(mzs_inverse_value_transform(y) - tvk);
    const float ptar = lane < A ? p.tp[((size_t)k * B + b) * A + lane] : 0.0f;
    const float lp = train_ce(p.pol[k] + (size_t)b * A, A, ptar, gs, p.gpol[k] + (size_t)b * A, &sm, lane,
                              train_policy_mode(p.loss_value));
    float lr = 0.0f;
    if (k > 0) {
        train_support_target(p.tr[(size_t)k * B + b], S, &idx, &wt);
        lr = train_ce(p.rew[k] + (size_t)b * S, S, lane == idx ? wt : 0.0f, gs, p.grew[k] + (size_t)b * S, &sm, lane,
                      p.loss_reward);
    }
    if (lane == 0) {
        p.loss_buf[(size_t)b * n_terms + t0] = lv;
        p.loss_buf[(size_t)b * n_terms + t0 + 1] = lp;
        if (k > 0) p.loss_buf[(size_t)b * n_terms + t0 + 2] = lr;
    }
}

/* sum of squares of the tensors inside the reference's l2 term: block partials (k_vt_finalize adds them) */
__global__ void __launch_bounds__(kT) k_vt_l2sq(const float* __restrict__ blob, const unsigned char* __restrict__ l2_flag,
                                                int n, double* __restrict__ part) {
    __shared__ double red[kT];
    const int i = blockIdx.x * kT + threadIdx.x;
    const float w = i < n && l2_flag[i] ? blob[i] : 0.0f;
    const double r = block_sum((double)w * (double)w, red);
    if (threadIdx.x == 0) part[blockIdx.x] = r;
}

struct FinalParams {
    const float* blob;
    const float* stat; /* running statistics after this step's forward passes (blob-sized copy) */
    const unsigned char* stat_flag;
    const unsigned char* l2_flag;
    const float* is_ratio;
    const float* loss_buf;
    const double* sq_part; /* k_vt_l2sq: one partial per gradient block */
    float* grad;
    float* loss_out;
    int n_params, B, K;
    float inv_batch;
};

/* blocks [0, n_grad_blocks): the l2 term of the gradient and the running-statistics slots; last block: the
 * batch means of the raw loss terms (muzero_model.py:603) and loss_nn (:725-734) */
__global__ void __launch_bounds__(kT) k_vt_finalize(const FinalParams p, int n_grad_blocks) {
    __shared__ double red[kT];
    const int B = p.B, n_terms = 3 * p.K - 1;
    if ((int)blockIdx.x < n_grad_blocks) {
        const int i = blockIdx.x * kT + threadIdx.x;
        if (i >= p.n_params) return;
        if (p.stat_flag[i]) {
            /* not a gradient: this shard's share of the change of a running statistic; data-parallel ranks
             * SUM the slots like the rest of the bucket and mzs_train_apply adds the sum to the blob */
            p.grad[i] = (p.stat[i] - p.blob[i]) * ((float)B * p.inv_batch);
        } else if (p.l2_flag[i]) {
            double wsum = 0.0;
            for (int b = 0; b < B; ++b) wsum += (double)((p.is_ratio ? p.is_ratio[b] : 1.0f) * p.inv_batch);
            p.grad[i] = (float)((double)p.grad[i] + 2.0 * 1e-4 * wsum * (double)p.blob[i]);
        }
        return;
    }
    double sq = 0.0;
    for (int i = threadIdx.x; i < n_grad_blocks; i += kT) sq += p.sq_part[i];
    double tot = 0.0, wsum = 0.0;
    for (int b = threadIdx.x; b < B; b += kT) {
        const double wb = (double)((p.is_ratio ? p.is_ratio[b] : 1.0f) * p.inv_batch);
        double sl = 0.0;
        for (int q = 0; q < n_terms; ++q) sl += (double)p.loss_buf[(size_t)b * n_terms + q];
        tot += wb * sl;
        wsum += wb;
    }
    sq = block_sum(sq, red);
    tot = block_sum(tot, red);
    wsum = block_sum(wsum, red);
    if (threadIdx.x == 0) p.loss_out[n_terms] = (float)(tot + wsum * 1e-4 * sq);
    for (int q = 0; q < n_terms; ++q) {
        double s = 0.0;
        for (int b = threadIdx.x; b < B; b += kT) s += (double)p.loss_buf[(size_t)b * n_terms + q];
        s = block_sum(s, red);
        if (threadIdx.x == 0) p.loss_out[q] = (float)(s * (double)p.inv_batch);
    }
}

/* Dropout2d(p = 0.5) masks from a counter-based generator: bit 40 of splitmix64(seed, step, index).
 * (torch draws them from its Philox stream, which nothing outside torch reproduces; parity tests inject
 * the reference's recorded masks instead: mzs_train_set_dropout.) */
__global__ void __launch_bounds__(kT) k_vt_draw_masks(unsigned char* masks, size_t n, uint64_t seed, uint64_t step) {
    const size_t i = (size_t)blockIdx.x * kT + threadIdx.x;
    if (i >= n) return;
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (step * 0x100000001B3ull + i + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    masks[i] = (unsigned char)((z >> 40) & 1);
}

}  // namespace

/* ================================================================== host side: tensors, tape, networks */

struct Tn {
    float* v;
    float* g;
    int c, s; /* channels x side x side per sample (s = 1: a [B][c] matrix) */
    size_t n; /* floats */
};

struct VisTrain {
    mzs_dims d;
    mzs_vision_offsets o;
    int device = 0, maxB = 0, K = 0;
    size_t n_params = 0, arena = 0;
    float* vals = nullptr;
    float* grads = nullptr;
    float* stat = nullptr;
    float* loss_buf = nullptr;
    double* sq_part = nullptr;
    unsigned char* stat_flag = nullptr;
    unsigned char* l2_flag = nullptr;
    unsigned char* own_masks = nullptr;
    size_t own_masks_n = 0;
    /* per call */
    bool dry = true, no_grad = false;
    size_t used = 0, mask_pos = 0;
    int B = 0;
    hipStream_t st = nullptr;
    const float* blob = nullptr;
    const float* quads = nullptr; /* the search kernels' quad store of the Linear layers (refreshed by every apply) */
    float* gw = nullptr;
    const unsigned char* masks = nullptr;
    const float* actions = nullptr;
    hipStream_t main_st = nullptr; /* the caller's stream */
    hipStream_t side = nullptr;    /* the prediction passes run beside the dynamics chain (vis_train_create) */
    std::vector<hipEvent_t> events;
    size_t n_events = 0;
    std::vector<std::function<void()>> tape;
    std::vector<void*> allocs;

    Tn tensor(int c, int s) {
        Tn t;
        t.c = c, t.s = s, t.n = (size_t)B * c * s * s;
        t.v = dry ? nullptr : vals + used;
        t.g = dry ? nullptr : grads + used;
        used += (t.n + 63) & ~(size_t)63;
        return t;
    }
    float* scratch(size_t n) {
        float* p = dry ? nullptr : vals + used;
        used += (n + 63) & ~(size_t)63;
        return p;
    }
    void back(std::function<void()> fn) {
        if (!dry && !no_grad) tape.push_back(std::move(fn));
    }
    const char* failed = nullptr; /* first internal failure of the step being built (vis_train_grad reports it) */
    hipEvent_t event() {
        if (n_events == events.size()) {
            hipEvent_t e = nullptr;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
                failed = "hipEventCreateWithFlags failed";
                return nullptr;
            }
            events.push_back(e);
        }
        return events[n_events++];
    }
    /* `to` continues after what `from` holds so far — in the forward pass now, and the other way round at the
     * same point of the backward replay */
    void link(hipStream_t from, hipStream_t to) {
        if (dry || from == to) return;
        hipEvent_t e = event();
        if (!e) return;
        (void)hipEventRecord(e, from);
        (void)hipStreamWaitEvent(to, e, 0);
        if (no_grad) return;
        hipEvent_t r = event();
        if (!r) return;
        tape.push_back([=] {
            (void)hipEventRecord(r, to);
            (void)hipStreamWaitEvent(from, r, 0);
        });
    }
};

#define VT_LAUNCH(vt, kernel, blocks, ...)                                                   \
    do {                                                                                     \
        if (!(vt).dry) hipLaunchKernelGGL(kernel, dim3(blocks), dim3(kT), 0, (vt).st, __VA_ARGS__); \
    } while (0)

namespace {

Tn conv3(VisTrain& v, const Tn& x, size_t w_off, int Ci, int Co, int stride, const Tn* res, bool need_dx = true) {
    const int S = x.s, So = (S - 1) / stride + 1, B = v.B;
    Tn y = v.tensor(Co, So);
    const float* w = v.blob + w_off;
    VT_LAUNCH(v, k_vt_conv3_fwd, nblk(y.n), x.v, w, res ? res->v : nullptr, y.v, B, Ci, Co, S, So, stride);
    float* gw = v.gw + w_off;
    const Tn r = res ? *res : Tn{};
    const bool has_res = res != nullptr;
    const int items = Co * Ci * 9;
    const Chunks ch = chunks_of((size_t)B * So * So);
    double* part = v.no_grad ? nullptr : reinterpret_cast<double*>(v.scratch(2 * (size_t)items * ch.n));
    const hipStream_t st = v.st;
    v.back([=] {
        if (has_res) hipLaunchKernelGGL(k_vt_axpy, dim3(nblk(y.n)), dim3(kT), 0, st, r.g, y.g, 1.0f, y.n);
        if ((size_t)B * So * So >= (size_t)1 << 16) /* large layers: one block per (co, ci), dy read once for nine taps */
            hipLaunchKernelGGL(k_vt_conv3_bwd_w, dim3(Co * Ci, ch.n), dim3(kT), 0, st, x.v, y.g, part, gw, B, Ci, Co, S,
                               So, stride, ch.per);
        else
            hipLaunchKernelGGL(k_vt_conv3_bwd_w1, dim3(items, ch.n), dim3(kT), 0, st, x.v, y.g, part, gw, B, Ci, Co, S,
                               So, stride, ch.per);
        if (need_dx)
            hipLaunchKernelGGL(k_vt_conv3_bwd_x, dim3(nblk(x.n)), dim3(kT), 0, st, y.g, w, x.g, B, Ci, Co, S, So, stride,
                               ch.n > 1 ? (const double*)part : nullptr, ch.n, gw);
        else if (ch.n > 1)
            hipLaunchKernelGGL(k_vt_accum_fin, dim3((items + 3) / 4), dim3(kT), 0, st, (const double*)part, items, ch.n, gw);
    });
    return y;
}

/* relu(batchnorm(x)) [x Dropout2d] on batch statistics */
Tn bn_act(VisTrain& v, const Tn& x, const mzs_vis_bn& bn, bool dropout) {
    const int C = x.c, SS = x.s * x.s, B = v.B;
    Tn y = v.tensor(C, x.s);
    float* mean = v.scratch(C);
    float* rstd = v.scratch(C);
    const Chunks ch = chunks_of((size_t)B * SS);
    double* part = reinterpret_cast<double*>(v.scratch(4 * (size_t)C * ch.n));
    const unsigned char* mask = nullptr;
    if (dropout) {
        mask = v.masks ? v.masks + v.mask_pos : nullptr;
        v.mask_pos += (size_t)B * C;
    }
    const float* gamma = v.blob + bn.w;
    const float* beta = v.blob + bn.b;
    double* part2 = v.no_grad ? nullptr : reinterpret_cast<double*>(v.scratch(4 * (size_t)C * ch.n));
    if (!v.dry) {
        hipLaunchKernelGGL(k_vt_bn_stats, dim3(C, ch.n), dim3(kT), 0, v.st, x.v, part, B, C, SS, ch.per);
        hipLaunchKernelGGL(k_vt_bn_act_fwd, dim3(nblk(y.n)), dim3(kT), 0, v.st, x.v, (const double*)part, ch.n, mean,
                           rstd, v.stat + bn.mean, v.stat + bn.var, gamma, beta, mask, y.v, B, C, SS);
    }
    float* gg = v.gw + bn.w;
    float* gb = v.gw + bn.b;
    const hipStream_t st = v.st;
    v.back([=] {
        hipLaunchKernelGGL(k_vt_bn_bwd_red, dim3(C, ch.n), dim3(kT), 0, st, x.v, mean, rstd, gamma, beta, mask, y.g,
                           part2, B, C, SS, ch.per);
        hipLaunchKernelGGL(k_vt_bn_bwd_x, dim3(nblk(x.n)), dim3(kT), 0, st, x.v, mean, rstd, gamma, beta, mask, y.g,
                           (const double*)part2, ch.n, gg, gb, x.g, B, C, SS);
    });
    return y;
}

/* Residual_block (neural_network_vision_model.py:41-82): x + conv_a(R(conv_b(R(conv_a(R(x)))))) */
Tn res_block(VisTrain& v, const Tn& x, const mzs_vis_res& r) {
    const int C = r.C;
    Tn y = conv3(v, bn_act(v, x, r.bn, true), r.conv_a, C, C, 1, nullptr);
    y = conv3(v, bn_act(v, y, r.bn, true), r.conv_b, C, C, 1, nullptr);
    return conv3(v, bn_act(v, y, r.bn, true), r.conv_a, C, C, 1, &x);
}

Tn pool(VisTrain& v, const Tn& x) {
    const int S = x.s, So = (S - 1) / 2 + 1, BC = v.B * x.c;
    Tn y = v.tensor(x.c, So);
    VT_LAUNCH(v, k_vt_pool_fwd, nblk(y.n), x.v, y.v, BC, S, So);
    const hipStream_t st = v.st;
    v.back([=] { hipLaunchKernelGGL(k_vt_pool_bwd, dim3(nblk(x.n)), dim3(kT), 0, st, y.g, x.g, BC, S, So); });
    return y;
}

Tn relu(VisTrain& v, const Tn& x) {
    Tn y = v.tensor(x.c, x.s);
    VT_LAUNCH(v, k_vt_relu_fwd, nblk(y.n), x.v, y.v, y.n);
    const hipStream_t st = v.st;
    v.back([=] { hipLaunchKernelGGL(k_vt_relu_bwd, dim3(nblk(x.n)), dim3(kT), 0, st, x.v, y.g, x.g, x.n); });
    return y;
}

/* 1 x 1 convolution + bias on a 7 x 7 map, flattened */
Tn conv1(VisTrain& v, const Tn& x, size_t w_off, size_t b_off, int Ci, int Co) {
    const int B = v.B;
    Tn y = v.tensor(Co * kS4, 1);
    const float* w = v.blob + w_off;
    VT_LAUNCH(v, k_vt_conv1_fwd, nblk(y.n), x.v, w, v.blob + b_off, y.v, B, Ci, Co);
    float* gw = v.gw + w_off;
    const int items = Co * Ci + Co; /* the bias follows its weight in the blob: one contiguous run of slots */
    if (b_off != w_off + (size_t)Co * Ci) v.failed = "blob layout: a 1 x 1 convolution's bias does not follow its weight";
    const Chunks ch = chunks_of((size_t)B * kS4);
    double* part = v.no_grad ? nullptr : reinterpret_cast<double*>(v.scratch(2 * (size_t)items * ch.n));
    const hipStream_t st = v.st;
    v.back([=] {
        hipLaunchKernelGGL(k_vt_conv1_bwd_w, dim3(items, ch.n), dim3(kT), 0, st, x.v, y.g, part, gw, B, Ci, Co, ch.per);
        hipLaunchKernelGGL(k_vt_conv1_bwd_x, dim3(nblk(x.n)), dim3(kT), 0, st, y.g, w, x.g, B, Ci, Co,
                           ch.n > 1 ? (const double*)part : nullptr, ch.n, gw);
    });
    return y;
}

Tn linear(VisTrain& v, const Tn& x, size_t w_off, size_t b_off, size_t q_off, int N, int K, bool act) {
    const int B = v.B;
    Tn y = v.tensor(N, 1);
    const float* W = v.blob + w_off;
    VT_LAUNCH(v, k_vt_linear_fwd, nblk(y.n), x.v, v.quads + q_off, v.blob + b_off, y.v, B, N, K, act ? 1 : 0);
    float* gw = v.gw + w_off;
    const int items = N * K + N; /* the bias follows its weight in the blob */
    if (b_off != w_off + (size_t)N * K) v.failed = "blob layout: a Linear layer's bias does not follow its weight";
    const Chunks ch = chunks_of((size_t)B, 8);
    double* part = v.no_grad ? nullptr : reinterpret_cast<double*>(v.scratch(2 * (size_t)items * ch.n));
    const hipStream_t st = v.st;
    const int relu = act ? 1 : 0;
    v.back([=] {
        hipLaunchKernelGGL(k_vt_linear_bwd_w, dim3(nblk(items), ch.n), dim3(kT), 0, st, x.v, y.g, y.v, part, gw, B, N, K,
                           (int)ch.per, relu);
        if (ch.n > 1)
            hipLaunchKernelGGL(k_vt_accum_fin, dim3((items + 3) / 4), dim3(kT), 0, st, (const double*)part, items, ch.n, gw);
        hipLaunchKernelGGL(k_vt_linear_bwd_x, dim3(nblk(x.n)), dim3(kT), 0, st, y.g, y.v, W, x.g, B, N, K, relu);
    });
    return y;
}

/* the Linear layers' copies in the search kernels' quad store (mzs_vision.h vis_lin_offsets: head after head —
 * 0 reward, 1 value, 2 policy — in, mid, out, each from a multiple of four floats) */
struct QuadOff {
    size_t in_w, mid_w, out_w;
};
QuadOff quad_offsets(const mzs_dims& d, int head) {
    const auto r4 = [](size_t n) { return (n + 3) & ~(size_t)3; };
    const size_t H = (size_t)d.hidden_dim, Sd = (size_t)d.state_dim;
    const size_t head_sd = r4(H * kHid) + r4(H * H) + r4(Sd * H); /* heads 0 and 1 both end in Sd outputs */
    QuadOff q;
    q.in_w = (size_t)head * head_sd;
    q.mid_w = q.in_w + r4(H * kHid);
    q.out_w = q.mid_w + r4(H * H);
    return q;
}

Tn mlp(VisTrain& v, const Tn& x, const mzs_vis_mlp& m, int head) {
    const int H = v.d.hidden_dim;
    const QuadOff q = quad_offsets(v.d, head);
    Tn y = linear(v, x, m.in_w, m.in_b, q.in_w, H, kHid, true);
    for (int l = 0; l < v.d.n_mid_layers; ++l) y = linear(v, y, m.mid_w, m.mid_b, q.mid_w, H, H, true);
    return linear(v, y, m.out_w, m.out_b, q.out_w, m.n_out, H, false);
}

Tn representation(VisTrain& v, const float* obs) {
    const mzs_vision_offsets& o = v.o;
    Tn x;
    x.v = const_cast<float*>(obs), x.g = nullptr, x.c = MZS_VIS_C, x.s = MZS_VIS_SIDE;
    x.n = (size_t)v.B * MZS_VIS_C * MZS_VIS_SIDE * MZS_VIS_SIDE;
    Tn y = conv3(v, x, o.r_conv_in, MZS_VIS_C, MZS_VIS_C / 2, 2, nullptr, /*need_dx=*/false);
    y = res_block(v, res_block(v, y, o.r_res_in), o.r_res_in);
    y = conv3(v, y, o.r_conv_out, MZS_VIS_C / 2, MZS_VIS_C, 2, nullptr);
    y = res_block(v, res_block(v, y, o.r_res_out), o.r_res_out);
    y = pool(v, y);
    for (int i = 0; i < 3; ++i) y = res_block(v, y, o.r_res_out);
    y = pool(v, y);
    return res_block(v, y, o.r_res);
}

void prediction(VisTrain& v, const Tn& h, Tn* pol, Tn* val) {
    const mzs_vision_offsets& o = v.o;
    Tn x = h;
    for (int l = 0; l < v.d.n_mid_layers; ++l) x = res_block(v, x, o.p_res);
    *val = mlp(v, conv1(v, x, o.p_conv_val, o.p_conv_val_b, MZS_VIS_C, MZS_VIS_C), o.p_v, 1);
    *pol = mlp(v, conv1(v, x, o.p_conv_pol, o.p_conv_pol_b, MZS_VIS_C, MZS_VIS_C), o.p_p, 2);
}

void dynamics(VisTrain& v, const Tn& h, const float* actions_k, Tn* rew, Tn* next) {
    const mzs_vision_offsets& o = v.o;
    const int B = v.B;
    Tn x = v.tensor(MZS_VIS_C + 1, MZS_VIS_S4);
    VT_LAUNCH(v, k_vt_cat_fwd, nblk(x.n), h.v, actions_k, x.v, B, v.d.n_actions);
    const hipStream_t st = v.st;
    v.back([=] { hipLaunchKernelGGL(k_vt_cat_bwd, dim3(nblk(h.n)), dim3(kT), 0, st, x.g, h.g, B); });
    Tn y = bn_act(v, conv3(v, x, o.d_conv, MZS_VIS_C + 1, MZS_VIS_C, 1, nullptr), o.d_bn, false);
    for (int l = 0; l < v.d.n_mid_layers; ++l) y = res_block(v, y, o.d_res);
    *next = relu(v, y);
    *rew = mlp(v, conv1(v, x, o.d_conv_rew, o.d_conv_rew_b, MZS_VIS_C + 1, MZS_VIS_C), o.d_fc, 0);
}

/* a prediction pass on the side stream: it reads the hidden state h through an alias with a gradient buffer of
 * its own (its backward pass runs beside the dynamics chain's, which adds to h.g too) */
void prediction_beside(VisTrain& v, const Tn& h, Tn* pol, Tn* val) {
    Tn hp = v.tensor(h.c, h.s);
    hp.v = h.v;
    const hipStream_t main_st = v.main_st;
    v.back([=] { hipLaunchKernelGGL(k_vt_axpy, dim3(nblk(h.n)), dim3(kT), 0, main_st, h.g, hp.g, 1.0f, h.n); });
    v.link(v.main_st, v.side);
    v.st = v.side;
    prediction(v, hp, pol, val);
    v.st = v.main_st;
}

/* Builds (and unless vt.dry, launches) one training step.  Two streams: the representation and the dynamics chain
 * on the caller's, the K + 1 prediction passes on a side stream (a prediction hangs off the chain: nothing but the
 * losses waits for it) — forward and backward. */
void build_step(VisTrain& v, const float* obs, LossParams* lp) {
    v.used = 0, v.mask_pos = 0, v.no_grad = false, v.n_events = 0;
    v.failed = nullptr;
    v.tape.clear();
    v.st = v.main_st;
    const int K = v.K;
    Tn h = representation(v, obs);
    Tn pol, val, rew, nh;
    prediction_beside(v, h, &pol, &val);
    lp->val[0] = val.v, lp->pol[0] = pol.v, lp->rew[0] = nullptr;
    lp->gval[0] = val.g, lp->gpol[0] = pol.g, lp->grew[0] = nullptr;
    for (int k = 0; k < K; ++k) {
        /* the K-th dynamics / prediction calls feed no loss term (zip() in evaluate_loss, muzero_model.py:655)
         * but their forward passes run, in training mode: running statistics move, masks are consumed */
        v.no_grad = k == K - 1;
        dynamics(v, h, v.actions + (size_t)k * v.B * v.d.n_actions, &rew, &nh);
        /* register_hook(grad * 0.5) on the next hidden state (muzero_model.py:640): the TOTAL gradient arriving
         * at it — from the prediction and from the next dynamics step — is halved on its way into this one */
        Tn hs = v.tensor(nh.c, nh.s);
        hs.v = nh.v;
        const hipStream_t st = v.st;
        v.back([=] { hipLaunchKernelGGL(k_vt_axpy, dim3(nblk(nh.n)), dim3(kT), 0, st, nh.g, hs.g, 0.5f, nh.n); });
        h = hs;
        prediction_beside(v, h, &pol, &val);
        if (k + 1 < K) {
            lp->val[k + 1] = val.v, lp->pol[k + 1] = pol.v, lp->rew[k + 1] = rew.v;
            lp->gval[k + 1] = val.g, lp->gpol[k + 1] = pol.g, lp->grew[k + 1] = rew.g;
        }
    }
    v.no_grad = false;
    v.link(v.side, v.main_st); /* the losses wait for the predictions; the predictions' backward passes for the losses */
}

std::vector<unsigned char> l2_flags(const mzs_dims& d, const mzs_vision_offsets& o) {
    /* neural_network_mlp_model.py:280-285: zip() over the three networks' parameters() stops at the shortest
     * list — dynamics: 15 parameters (L >= 1) — so the FIRST 15 of each network count.  parameters() order:
     * registration order, a Residual_block = BatchNorm2d weight, bias, convolution_1 (conv_a), convolution_3. */
    const int C = MZS_VIS_C, H = d.hidden_dim, Sd = d.state_dim, A = d.n_actions;
    struct R {
        size_t off, n;
    };
    const auto res = [](std::vector<R>& l, const mzs_vis_res& r) {
        const size_t c = (size_t)r.C;
        l.push_back({r.bn.w, c}), l.push_back({r.bn.b, c}), l.push_back({r.conv_a, c * c * 9}), l.push_back({r.conv_b, c * c * 9});
    };
    const auto mlp = [&](std::vector<R>& l, const mzs_vis_mlp& m) {
        l.push_back({m.in_w, (size_t)H * kHid}), l.push_back({m.in_b, (size_t)H});
        l.push_back({m.mid_w, (size_t)H * H}), l.push_back({m.mid_b, (size_t)H});
        l.push_back({m.out_w, (size_t)m.n_out * H}), l.push_back({m.out_b, (size_t)m.n_out});
    };
    std::vector<R> r, dy, p;
    r.push_back({o.r_conv_in, (size_t)(C / 2) * C * 9});
    res(r, o.r_res_in);
    r.push_back({o.r_conv_out, (size_t)C * (C / 2) * 9});
    res(r, o.r_res_out);
    res(r, o.r_res);
    r.push_back({o.r_conv, (size_t)C * C * 9}), r.push_back({o.r_bn.w, (size_t)C}), r.push_back({o.r_bn.b, (size_t)C});
    dy.push_back({o.d_conv, (size_t)C * (C + 1) * 9}), dy.push_back({o.d_bn.w, (size_t)C}), dy.push_back({o.d_bn.b, (size_t)C});
    res(dy, o.d_res);
    dy.push_back({o.d_conv_rew, (size_t)C * (C + 1)}), dy.push_back({o.d_conv_rew_b, (size_t)C});
    mlp(dy, o.d_fc);
    res(p, o.p_res);
    p.push_back({o.p_conv_val, (size_t)C * C}), p.push_back({o.p_conv_val_b, (size_t)C});
    mlp(p, o.p_v);
    p.push_back({o.p_conv_pol, (size_t)C * C}), p.push_back({o.p_conv_pol_b, (size_t)C});
    mlp(p, o.p_p);
    (void)Sd, (void)A;
    const size_t n = std::min(r.size(), std::min(dy.size(), p.size()));
    std::vector<unsigned char> f(o.total, 0);
    for (const auto* l : {&r, &dy, &p})
        for (size_t i = 0; i < n; ++i)
            for (size_t j = 0; j < (*l)[i].n; ++j) f[(*l)[i].off + j] = 1;
    return f;
}

int vt_fail(std::string* err, const std::string& msg) {
    if (err) *err = msg;
    return MZS_E_HIP;
}

#define VT_TRY(expr)                                                                         \
    do {                                                                                     \
        const hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess) return vt_fail(err, std::string(#expr ": ") + hipGetErrorString(e_)); \
    } while (0)

}  // namespace

int vis_train_create(VisTrain** out, const mzs_dims& d, int max_batch, int K, int device, std::string* err) {
    if (K > kMaxUnroll) {
        if (err) *err = "vision_model training: unroll_steps <= 32";
        return MZS_E_ARG;
    }
    if (d.n_mid_layers < 1) {
        if (err) *err = "vision_model training: number_of_hidden_layer >= 1";
        return MZS_E_ARG;
    }
    VisTrain* v = new VisTrain;
    v->d = d, v->o = mzs_vision_layout(&d), v->device = device, v->maxB = max_batch, v->K = K;
    v->n_params = v->o.total;
    /* dry run at the largest batch: sizes the arenas and counts the mask bytes */
    v->dry = true, v->B = max_batch;
    LossParams lp;
    build_step(*v, nullptr, &lp);
    if (v->failed) {
        if (err) *err = v->failed;
        delete v;
        return MZS_E_ARG;
    }
    v->arena = v->used;
    v->own_masks_n = v->mask_pos;
    const auto alloc = [&](void** p, size_t bytes) {
        const hipError_t e = hipMalloc(p, bytes + 256);
        if (e == hipSuccess) {
            v->allocs.push_back(*p);
            return hipMemset(*p, 0, bytes + 256);
        }
        return e;
    };
    std::vector<unsigned char> sf(v->n_params, 0);
    const mzs_vis_bn* bns[] = {&v->o.r_res_in.bn, &v->o.r_res_out.bn, &v->o.r_res.bn, &v->o.r_bn,
                               &v->o.d_bn,        &v->o.d_res.bn,     &v->o.p_res.bn};
    const int cs[] = {MZS_VIS_C / 2, MZS_VIS_C, MZS_VIS_C, MZS_VIS_C, MZS_VIS_C, MZS_VIS_C, MZS_VIS_C};
    for (int i = 0; i < 7; ++i)
        for (int c = 0; c < cs[i]; ++c) sf[bns[i]->mean + c] = 1, sf[bns[i]->var + c] = 1;
    const std::vector<unsigned char> lf = l2_flags(d, v->o);
    hipError_t e;
    if ((e = alloc((void**)&v->vals, v->arena * sizeof(float))) != hipSuccess ||
        (e = alloc((void**)&v->grads, v->arena * sizeof(float))) != hipSuccess ||
        (e = alloc((void**)&v->stat, v->n_params * sizeof(float))) != hipSuccess ||
        (e = alloc((void**)&v->loss_buf, (size_t)max_batch * (3 * K) * sizeof(float))) != hipSuccess ||
        (e = alloc((void**)&v->sq_part, (size_t)nblk(v->n_params) * sizeof(double))) != hipSuccess ||
        (e = alloc((void**)&v->stat_flag, v->n_params)) != hipSuccess ||
        (e = alloc((void**)&v->l2_flag, v->n_params)) != hipSuccess ||
        (e = alloc((void**)&v->own_masks, v->own_masks_n)) != hipSuccess ||
        (e = hipMemcpy(v->stat_flag, sf.data(), sf.size(), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(v->l2_flag, lf.data(), lf.size(), hipMemcpyHostToDevice)) != hipSuccess) {
        const std::string msg = std::string("vision training buffers: ") + hipGetErrorString(e);
        vis_train_destroy(v);
        return vt_fail(err, msg);
    }
    if ((e = hipStreamCreateWithFlags(&v->side, hipStreamNonBlocking)) != hipSuccess) {
        const std::string msg = std::string("vision training side stream: ") + hipGetErrorString(e);
        vis_train_destroy(v);
        return vt_fail(err, msg);
    }
    *out = v;
    return MZS_OK;
}

void vis_train_destroy(VisTrain* v) {
    if (!v) return;
    if (v->side) {
        (void)hipStreamSynchronize(v->side);
        (void)hipStreamDestroy(v->side);
    }
    for (hipEvent_t e : v->events) (void)hipEventDestroy(e);
    for (void* p : v->allocs) (void)hipFree(p);
    delete v;
}

int64_t vis_train_mask_count(VisTrain* v, int batch) {
    v->dry = true, v->B = batch;
    LossParams lp;
    build_step(*v, nullptr, &lp);
    return (int64_t)v->mask_pos;
}

const unsigned char* vis_train_stat_flags(const VisTrain* v) { return v->stat_flag; }

int vis_train_grad(VisTrain* v, const float* blob_dev, const float* quad_store_dev, int batch, const float* obs_dev, const float* actions_dev,
                   const float* tv_dev, const float* tr_dev, const float* tp_dev, const float* is_ratio_dev,
                   float inv_global_batch, int loss_value, int loss_reward, const unsigned char* masks_dev,
                   uint64_t seed, uint64_t step, float* grad_dev, float* loss_dev, float* priority_dev,
                   hipStream_t st, std::string* err) {
    const int K = v->K;
    v->st = st, v->main_st = st, v->blob = blob_dev, v->gw = grad_dev;
    v->actions = actions_dev;
    v->quads = quad_store_dev;
    /* a dry pass first: the step's arena use and mask count, checked before anything is launched */
    const size_t n_masks = (size_t)vis_train_mask_count(v, batch);
    if (v->failed) return vt_fail(err, v->failed);
    if (v->used > v->arena || n_masks > v->own_masks_n) return vt_fail(err, "vision training arena too small (internal)");
    v->dry = false, v->B = batch;
    if (!masks_dev) {
        hipLaunchKernelGGL(k_vt_draw_masks, dim3(nblk(n_masks)), dim3(kT), 0, st, v->own_masks, n_masks, seed, step);
        masks_dev = v->own_masks;
    }
    v->masks = masks_dev;
    VT_TRY(hipMemsetAsync(v->grads, 0, v->arena * sizeof(float), st));
    VT_TRY(hipMemsetAsync(grad_dev, 0, v->n_params * sizeof(float), st));
    VT_TRY(hipMemcpyAsync(v->stat, blob_dev, v->n_params * sizeof(float), hipMemcpyDeviceToDevice, st));
    LossParams lp;
    std::memset(&lp, 0, sizeof lp);
    build_step(*v, obs_dev, &lp);
    if (v->failed) { /* launches already queued read only buffers this object owns; the step's outputs are undefined */
        (void)hipStreamSynchronize(v->side);
        v->tape.clear();
        return vt_fail(err, v->failed);
    }
    lp.tv = tv_dev, lp.tr = tr_dev, lp.tp = tp_dev, lp.is_ratio = is_ratio_dev;
    lp.loss_buf = v->loss_buf, lp.priority = priority_dev;
    lp.B = batch, lp.K = K, lp.S = v->d.state_dim, lp.A = v->d.n_actions;
    lp.loss_value = loss_value, lp.loss_reward = loss_reward, lp.inv_batch = inv_global_batch;
    hipLaunchKernelGGL(k_vt_loss, dim3(K * batch), dim3(kWave), 0, st, lp);
    for (auto it = v->tape.rbegin(); it != v->tape.rend(); ++it) (*it)();
    v->tape.clear();
    FinalParams fp;
    fp.blob = blob_dev, fp.stat = v->stat, fp.stat_flag = v->stat_flag, fp.l2_flag = v->l2_flag;
    fp.is_ratio = is_ratio_dev, fp.loss_buf = v->loss_buf, fp.grad = grad_dev, fp.loss_out = loss_dev;
    fp.n_params = (int)v->n_params, fp.B = batch, fp.K = K, fp.inv_batch = inv_global_batch;
    const int ngb = (int)nblk(v->n_params);
    fp.sq_part = v->sq_part;
    hipLaunchKernelGGL(k_vt_l2sq, dim3(ngb), dim3(kT), 0, st, blob_dev, (const unsigned char*)v->l2_flag, (int)v->n_params,
                       v->sq_part);
    hipLaunchKernelGGL(k_vt_finalize, dim3(ngb + 1), dim3(kT), 0, st, fp, ngb);
    VT_TRY(hipGetLastError());
    return MZS_OK;
}

}  // namespace mzs
