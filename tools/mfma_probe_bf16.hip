// mfma_probe_bf16.hip — what v_mfma_f32_4x4x4_16B_bf16 computes on gfx950, bit for bit.
// Diagnostic only (not part of the library).  Dumps raw (A, B, C, D) cases for offline analysis
// with exact arithmetic (tools/mfma_bf16_analyze.py), checks v_cvt_pk_bf16_f32 against the integer
// round-to-nearest-even formula, and times dependent chains.
// Build: hipcc --offload-arch=gfx950 -O2 -o mfma_probe_bf16 mfma_probe_bf16.hip
// Run:   ./mfma_probe_bf16 gpurun_out/mfma_bf16_cases.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(r_), __LINE__); exit(1);} } while (0)

// one MFMA per test t: lane l supplies A (4 bf16), B (4 bf16) and C (4 floats); D comes back per lane
template <int BLGP>
__global__ void k_cases(const s16x4* a, const s16x4* b, const f32x4* c, f32x4* d, int T) {
    const int l = threadIdx.x;
    for (int t = 0; t < T; ++t) {
        const f32x4 r = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a[t * 64 + l], b[t * 64 + l], c[t * 64 + l], 0, 0, BLGP);
        d[t * 64 + l] = r;
    }
}

__global__ void k_cvt(const float* x, unsigned short* hw, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 < n) {
        const f32x2 v = {x[2 * i], x[2 * i + 1]};
        const bf16x2 r = __builtin_convertvector(v, bf16x2);
        unsigned u;
        __builtin_memcpy(&u, &r, 4);
        hw[2 * i] = (unsigned short)(u & 0xffffu);
        hw[2 * i + 1] = (unsigned short)(u >> 16);
    }
}

template <int NCH>
__global__ void k_tput(float* out, long long* cyc, int n) {
    const int l = threadIdx.x;
    f32x4 c[NCH];
    for (int j = 0; j < NCH; ++j) c[j] = f32x4{0, 0, 0, 0};
    s16x4 a = {(short)(0x3f80 + l), (short)0x3f81, (short)0x3f82, (short)0x3f83};
    s16x4 b = {(short)0x3f00, (short)(0x3f01 + l), (short)0x3f02, (short)0x3f03};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int j = 0; j < NCH; ++j) c[j] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, b, c[j], 0, 0, 4);
    }
    __builtin_amdgcn_s_waitcnt(0);
    const long long t1 = __builtin_amdgcn_s_memtime();
    f32x4 s = c[0];
    for (int j = 1; j < NCH; ++j) s += c[j];
    for (int r = 0; r < 4; ++r) out[r * 64 + l] = s[r];
    if (l == 0) cyc[0] = t1 - t0;
}

static uint32_t rng_state = 12345u;
static uint32_t rnd32() {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 17;
    rng_state ^= rng_state << 5;
    return rng_state;
}
// random bf16 with exponent in [-spread, spread] around 1.0
static uint16_t rnd_bf16(int spread) {
    const int e = 127 + (int)(rnd32() % (2 * spread + 1)) - spread;
    const uint16_t m = (uint16_t)(rnd32() & 0x7f);
    const uint16_t s = (uint16_t)(rnd32() & 1);
    return (uint16_t)((s << 15) | (e << 7) | m);
}
static float rnd_f32(int spread) {
    const int e = 127 + (int)(rnd32() % (2 * spread + 1)) - spread;
    const uint32_t u = ((rnd32() & 1u) << 31) | ((uint32_t)e << 23) | (rnd32() & 0x7fffffu);
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// designed cases from a file: case n sits on the diagonal of MFMA n / 64 (lane n % 64 supplies A row i = lane & 3
// and B column j = lane & 3 of its block, C and D are register lane & 3)
__global__ void k_diag(const s16x4* a, const s16x4* b, const float* c, float* d, int T) {
    const int l = threadIdx.x;
    for (int t = 0; t < T; ++t) {
        f32x4 cc = {0, 0, 0, 0};
        cc[l & 3] = c[t * 64 + l];
        const f32x4 r = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a[t * 64 + l], b[t * 64 + l], cc, 0, 0, 0);
        d[t * 64 + l] = r[l & 3];
    }
}

static int run_file(const char* in, const char* out) {
    FILE* f = fopen(in, "rb");
    if (!f) { printf("cannot open %s\n", in); return 1; }
    int n = 0;
    if (fread(&n, 4, 1, f) != 1 || n <= 0 || n % 64) { printf("bad case count\n"); return 1; }
    std::vector<uint16_t> ha((size_t)n * 4), hb((size_t)n * 4);
    std::vector<float> hc(n), hd(n);
    if (fread(ha.data(), 2, ha.size(), f) != ha.size() || fread(hb.data(), 2, hb.size(), f) != hb.size() ||
        fread(hc.data(), 4, n, f) != (size_t)n) { printf("short file\n"); return 1; }
    fclose(f);
    s16x4 *a, *b; float *c, *d;
    CK(hipMalloc(&a, ha.size() * 2)); CK(hipMalloc(&b, hb.size() * 2)); CK(hipMalloc(&c, n * 4)); CK(hipMalloc(&d, n * 4));
    CK(hipMemcpy(a, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(b, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(c, hc.data(), n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_diag, dim3(1), dim3(64), 0, 0, a, b, c, d, n / 64);
    CK(hipMemcpy(hd.data(), d, n * 4, hipMemcpyDeviceToHost));
    f = fopen(out, "wb");
    if (!f) { printf("cannot open %s\n", out); return 1; }
    fwrite(hd.data(), 4, n, f);
    fclose(f);
    printf("== %d designed cases -> %s\n", n, out);
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 3 && !strcmp(argv[1], "run")) return run_file(argv[2], argv[3]);
    const char* path = argc > 1 ? argv[1] : "mfma_bf16_cases.bin";
    // ---- cases: three regimes of exponent spread; BLGP 0 and 4 (broadcast of lanes 0..15's B)
    const int T = 768;
    std::vector<uint16_t> ha((size_t)T * 64 * 4), hb((size_t)T * 64 * 4);
    std::vector<float> hc((size_t)T * 64 * 4), hd0((size_t)T * 64 * 4), hd4((size_t)T * 64 * 4);
    for (int t = 0; t < T; ++t) {
        const int regime = t % 3;
        const int sp_ab = regime == 0 ? 1 : (regime == 1 ? 4 : 12);
        const int sp_c = regime == 0 ? 2 : (regime == 1 ? 8 : 24);
        for (int i = 0; i < 64 * 4; ++i) {
            ha[(size_t)t * 256 + i] = rnd_bf16(sp_ab);
            hb[(size_t)t * 256 + i] = rnd_bf16(sp_ab);
            hc[(size_t)t * 256 + i] = (t < 12) ? 0.0f : rnd_f32(sp_c);
        }
        if (t < 6)  // tiny integers: exact under every hypothesis -> pins the layout
            for (int i = 0; i < 64 * 4; ++i) {
                ha[(size_t)t * 256 + i] = (uint16_t)(0x3f80 + ((rnd32() % 3) << 6));   // 1, 1.5, 2 (0x4000) approx
                hb[(size_t)t * 256 + i] = (uint16_t)(0x3f80 + ((rnd32() % 3) << 6));
            }
    }
    s16x4 *a, *b;
    f32x4 *c, *d;
    CK(hipMalloc(&a, ha.size() * 2)); CK(hipMalloc(&b, hb.size() * 2));
    CK(hipMalloc(&c, hc.size() * 4)); CK(hipMalloc(&d, hc.size() * 4));
    CK(hipMemcpy(a, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(b, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(c, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_cases<0>, dim3(1), dim3(64), 0, 0, a, b, c, d, T);
    CK(hipMemcpy(hd0.data(), d, hd0.size() * 4, hipMemcpyDeviceToHost));
    hipLaunchKernelGGL(k_cases<4>, dim3(1), dim3(64), 0, 0, a, b, c, d, T);
    CK(hipMemcpy(hd4.data(), d, hd4.size() * 4, hipMemcpyDeviceToHost));
    FILE* f = fopen(path, "wb");
    if (!f) { printf("cannot open %s\n", path); return 1; }
    const int hdr[2] = {T, 64};
    fwrite(hdr, 4, 2, f);
    fwrite(ha.data(), 2, ha.size(), f);
    fwrite(hb.data(), 2, hb.size(), f);
    fwrite(hc.data(), 4, hc.size(), f);
    fwrite(hd0.data(), 4, hd0.size(), f);
    fwrite(hd4.data(), 4, hd4.size(), f);
    fclose(f);
    printf("== wrote %d x 64 lanes of (A, B, C, D[blgp0], D[blgp4]) to %s\n", T, path);

    // ---- v_cvt_pk_bf16_f32 vs integer RNE
    {
        const int n = 1 << 20;
        std::vector<float> hx(n);
        std::vector<unsigned short> hh(n);
        for (int i = 0; i < n; ++i) {
            hx[i] = rnd_f32(30);
            if (i % 16 == 0) {  // exact ties: low 16 bits = 0x8000
                uint32_t u;
                memcpy(&u, &hx[i], 4);
                u = (u & 0xffff0000u) | 0x8000u;
                memcpy(&hx[i], &u, 4);
            }
        }
        float* x;
        unsigned short* hw;
        CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&hw, n * 2));
        CK(hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_cvt, dim3(n / 2 / 256), dim3(256), 0, 0, x, hw, n);
        CK(hipMemcpy(hh.data(), hw, n * 2, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < n; ++i) {
            uint32_t u;
            memcpy(&u, &hx[i], 4);
            const uint16_t want = (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
            bad += want != hh[i];
        }
        printf("== float -> bf16 (compiler's conversion on gfx950) vs integer RNE: %d / %d mismatches\n", bad, n);
    }
    // ---- latency / throughput
    {
        float* out; long long* cyc; long long h;
        CK(hipMalloc(&out, 1024)); CK(hipMalloc(&cyc, 8));
        const int n = 1000;
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k_tput<1>, dim3(1), dim3(64), 0, 0, out, cyc, n); CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
            printf("== 4x4x4 bf16 dependent chain: %.2f cycles/mfma\n", (double)h / (16.0 * n));
            hipLaunchKernelGGL(k_tput<2>, dim3(1), dim3(64), 0, 0, out, cyc, n); CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
            printf("== 4x4x4 bf16 2 chains: %.2f cycles/mfma\n", (double)h / (32.0 * n));
            hipLaunchKernelGGL(k_tput<4>, dim3(1), dim3(64), 0, 0, out, cyc, n); CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
            printf("== 4x4x4 bf16 4 chains: %.2f cycles/mfma\n", (double)h / (64.0 * n));
        }
    }
    return 0;
}
