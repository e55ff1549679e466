// mfma_probe.hip — layout / broadcast / numerics / latency probe of v_mfma_f32_4x4x1_16b_f32 on gfx950.
// Diagnostic only (not part of the library).  Build: hipcc --offload-arch=gfx950 -O2 -o mfma_probe mfma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CBSZ, int ABID, int BLGP>
__global__ void k_layout(const float* a, const float* b, float* d) {
    int l = threadIdx.x;
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, CBSZ, ABID, BLGP);
    for (int r = 0; r < 4; ++r) d[r * 64 + l] = c[r];
}

// numerics: chain of K mfma (k ascending) vs fmaf chain. lane=neuron(B operand), rows=games (A operand, broadcast)
__global__ void k_chain(const float* w /*[K][64]*/, const float* x /*[K][4]*/, const float* bias, float* out, int K) {
    int l = threadIdx.x;
    f32x4 c = {bias[l], bias[l], bias[l], bias[l]};
    for (int k = 0; k < K; ++k) {
        float av = x[k * 4 + (l & 3)];
        c = __builtin_amdgcn_mfma_f32_4x4x1f32(av, w[k * 64 + l], c, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) out[r * 64 + l] = c[r];
}

__global__ void k_latency(float* out, long long* cyc, int n) {
    int l = threadIdx.x;
    f32x4 c = {0, 0, 0, 0};
    float a = 1.0f + l * 1e-3f, b = 1.0f - l * 1e-3f;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0);
    long long t1 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < 4; ++r) out[r * 64 + l] = c[r];
    if (l == 0) cyc[0] = t1 - t0;
}

__global__ void k_latency_cbsz(float* out, long long* cyc, int n) {
    int l = threadIdx.x;
    f32x4 c = {0, 0, 0, 0};
    float a = 1.0f + l * 1e-3f, b = 1.0f - l * 1e-3f;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
        c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, 0, 0);
        c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, 1, 0);
        c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, 2, 0);
        c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, 3, 0);
        c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, 4, 0);
        c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, 5, 0);
        c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, 6, 0);
        c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, 7, 0);
        c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, 8, 0);
        c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, 9, 0);
        c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, 10, 0);
        c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, 11, 0);
        c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, 12, 0);
        c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, 13, 0);
        c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, 14, 0);
        c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, 15, 0);
    }
    __builtin_amdgcn_s_waitcnt(0);
    long long t1 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < 4; ++r) out[r * 64 + l] = c[r];
    if (l == 0) cyc[0] = t1 - t0;
}

__global__ void k_latency_16(float* out, long long* cyc, int n) {
    int l = threadIdx.x;
    f32x4 c = {0, 0, 0, 0};
    float a = 1.0f + l * 1e-3f, b = 1.0f - l * 1e-3f;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0);
    long long t1 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < 4; ++r) out[r * 64 + l] = c[r];
    if (l == 0) cyc[0] = t1 - t0;
}

// dependent v_fma chain for reference
__global__ void k_latency_fma(float* out, long long* cyc, int n) {
    int l = threadIdx.x;
    float c = 0.f, a = 1.0f + l * 1e-3f, b = 1.0f - l * 1e-3f;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) c = __builtin_fmaf(a, c, b);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[l] = c;
    if (l == 0) cyc[0] = t1 - t0;
}


template <int NCH>
__global__ void k_tput(float* out, long long* cyc, int n) {
    int l = threadIdx.x;
    f32x4 c[NCH];
    for (int j = 0; j < NCH; ++j) c[j] = f32x4{0, 0, 0, 0};
    float a = 1.0f + l * 1e-3f, b = 1.0f - l * 1e-3f;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int j = 0; j < NCH; ++j) c[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c[j], 4, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0);
    long long t1 = __builtin_amdgcn_s_memtime();
    f32x4 s = c[0];
    for (int j = 1; j < NCH; ++j) s += c[j];
    for (int r = 0; r < 4; ++r) out[r * 64 + l] = s[r];
    if (l == 0) cyc[0] = t1 - t0;
}
// one dependent mfma chain with NV independent v_fma per mfma in its shadow
template <int NV>
__global__ void k_shadow(float* out, long long* cyc, int n) {
    int l = threadIdx.x;
    f32x4 c = {0, 0, 0, 0};
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = l * 0.5f + j;
    float a = 1.0f + l * 1e-3f, b = 1.0f - l * 1e-3f;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; ++j) v[j] = __builtin_fmaf(v[j], a, b);
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int j = 0; j < 8; ++j) s += v[j];
    for (int r = 0; r < 4; ++r) out[r * 64 + l] = c[r] + s;
    if (l == 0) cyc[0] = t1 - t0;
}

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(r_), __LINE__); exit(1);} } while (0)

template <int CBSZ, int ABID, int BLGP>
void layout(const char* name) {
    float ha[64], hb[64], hd[256], *a, *b, *d;
    CK(hipMalloc(&a, 256)); CK(hipMalloc(&b, 256)); CK(hipMalloc(&d, 1024));
    int amap[256], bmap[256];
    for (int pass = 0; pass < 2; ++pass) {
        for (int l = 0; l < 64; ++l) { ha[l] = pass == 0 ? (float)l : 1.0f; hb[l] = pass == 0 ? 1.0f : (float)l; }
        CK(hipMemcpy(a, ha, 256, hipMemcpyHostToDevice)); CK(hipMemcpy(b, hb, 256, hipMemcpyHostToDevice));
        hipLaunchKernelGGL((k_layout<CBSZ, ABID, BLGP>), dim3(1), dim3(64), 0, 0, a, b, d);
        CK(hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost));
        for (int i = 0; i < 256; ++i) (pass == 0 ? amap : bmap)[i] = (int)hd[i];
    }
    printf("== %s (cbsz=%d abid=%d blgp=%d): D[reg r][lane l] = A[lane]*B[lane]\n", name, CBSZ, ABID, BLGP);
    for (int r = 0; r < 4; ++r) {
        printf(" r=%d:", r);
        for (int l = 0; l < 64; ++l) printf(" %d*%d", amap[r * 64 + l], bmap[r * 64 + l]);
        printf("\n");
    }
    CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(d));
}

int main() {
    
    
    
    
    
    
    
    // numerics
    {
        const int K = 64;
        float hw[K * 64], hx[K * 4], hbias[64], hout[256];
        srand(1);
        auto rnd = []() { return (float)((rand() / (double)RAND_MAX) * 2 - 1); };
        for (auto& v : hw) v = rnd() * 0.3f;
        for (auto& v : hx) v = rnd() * 3.0f;
        for (auto& v : hbias) v = rnd();
        // some subnormal-producing entries
        hw[5] = 1e-30f; hx[0] = 1e-12f;
        float *w, *x, *bias, *out;
        CK(hipMalloc(&w, sizeof hw)); CK(hipMalloc(&x, sizeof hx)); CK(hipMalloc(&bias, 256)); CK(hipMalloc(&out, 1024));
        CK(hipMemcpy(w, hw, sizeof hw, hipMemcpyHostToDevice)); CK(hipMemcpy(x, hx, sizeof hx, hipMemcpyHostToDevice));
        CK(hipMemcpy(bias, hbias, 256, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, w, x, bias, out, K);
        CK(hipMemcpy(hout, out, 1024, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int r = 0; r < 4; ++r)
            for (int l = 0; l < 64; ++l) {
                float acc = hbias[l];
                for (int k = 0; k < K; ++k) acc = fmaf(hx[k * 4 + r], hw[k * 64 + l], acc);
                if (memcmp(&acc, &hout[r * 64 + l], 4)) { if (bad < 5) printf("  mismatch r=%d l=%d cpu=%.9g gpu=%.9g\n", r, l, acc, hout[r * 64 + l]); ++bad; }
            }
        printf("== numerics: K=%d chain vs fmaf: %d / 256 mismatches (assuming D[r=game][lane=neuron])\n", K, bad);
    }
    // latency
    {
        float* out; long long* cyc; long long h;
        CK(hipMalloc(&out, 1024)); CK(hipMalloc(&cyc, 8));
        const int n = 1000;
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k_latency, dim3(1), dim3(64), 0, 0, out, cyc, n); CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
            printf("== 4x4x1 dependent chain: %.2f cycles/mfma\n", (double)h / (16.0 * n));
            hipLaunchKernelGGL(k_latency_cbsz, dim3(1), dim3(64), 0, 0, out, cyc, n); CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
            printf("== 4x4x1 cbsz4 dependent chain: %.2f cycles/mfma\n", (double)h / (16.0 * n));

            hipLaunchKernelGGL(k_tput<2>, dim3(1), dim3(64), 0, 0, out, cyc, n); CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
            printf("== 4x4x1 2 chains: %.2f cycles/mfma\n", (double)h / (32.0 * n));
            hipLaunchKernelGGL(k_tput<4>, dim3(1), dim3(64), 0, 0, out, cyc, n); CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
            printf("== 4x4x1 4 chains: %.2f cycles/mfma\n", (double)h / (64.0 * n));
            hipLaunchKernelGGL(k_shadow<1>, dim3(1), dim3(64), 0, 0, out, cyc, n); CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
            printf("== 4x4x1 chain + 1 fma: %.2f cycles/mfma\n", (double)h / (16.0 * n));
            hipLaunchKernelGGL(k_shadow<2>, dim3(1), dim3(64), 0, 0, out, cyc, n); CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
            printf("== 4x4x1 chain + 2 fma: %.2f cycles/mfma\n", (double)h / (16.0 * n));
            hipLaunchKernelGGL(k_shadow<3>, dim3(1), dim3(64), 0, 0, out, cyc, n); CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
            printf("== 4x4x1 chain + 3 fma: %.2f cycles/mfma\n", (double)h / (16.0 * n));
            hipLaunchKernelGGL(k_latency_16, dim3(1), dim3(64), 0, 0, out, cyc, n); CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
            printf("== 16x16x4 dependent chain: %.2f cycles/mfma\n", (double)h / (16.0 * n));
            hipLaunchKernelGGL(k_latency_fma, dim3(1), dim3(64), 0, 0, out, cyc, n); CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
            printf("== v_fma_f32 dependent chain: %.2f cycles/fma\n", (double)h / (16.0 * n));
        }
    }
    return 0;
}
