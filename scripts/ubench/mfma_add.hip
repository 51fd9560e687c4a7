// Micro-benchmark: can the FP64 matrix pipe serve as a second FP64 ADDER next to the vector ALU, bit-exactly?
//   D = A x B + C with v_mfma_f64_4x4x4_4b_f64 and A = the 4x4 identity in every block gives D[lane] = B[lane] + C[lane]
//   (one fused multiply-add with multiplier 1.0 per element = one IEEE addition, all other terms are exact zeros).
// Part 1 checks that bit for bit against v_add_f64 on adversarial operands; part 2 measures issue rates of MFMA-adds, vector
// FP64 adds and mixtures of both on one SIMD (s_memtime ticks per instruction, 1 and 2 waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <random>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ double ident_a(int lane) { const int l = lane & 15; return ((l & 3) == (l >> 2)) ? 1.0 : 0.0; }

__global__ void k_check(const double* b, const double* c, double* d_add, double* d_sub, int n) {
    const int lane = threadIdx.x & 63;
    const double a = ident_a(lane);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        d_add[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b[i], c[i], 0, 0, 0);
        d_sub[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b[i], c[i], 0, 0, 2);     // blgp bit 1: negate B
    }
}

// MODE 0: MFMA adds only; 1: vector adds only; 2: per iteration 8 MFMA adds + 24 vector adds interleaved (independent chains)
template <int MODE>
__global__ void k_rate(double* out, long long* cyc, int iters, double seed) {
    const int lane = threadIdx.x & 63;
    const double a = ident_a(lane);
    double m[8], v[24];
#pragma unroll
    for (int i = 0; i < 8; i++) m[i] = seed + lane * 1e-9 + i;
#pragma unroll
    for (int i = 0; i < 24; i++) v[i] = seed * 0.5 + lane * 1e-7 + i;
    const double c1 = 1.0000001;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 8; i++) m[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, c1, m[i], 0, 0, 0);
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 24; i++) v[i] = v[i] + c1;
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                m[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, c1, m[i], 0, 0, 0);
                v[3 * i] = v[3 * i] + c1; v[3 * i + 1] = v[3 * i + 1] + c1; v[3 * i + 2] = v[3 * i + 2] + c1;
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += m[i];
#pragma unroll
    for (int i = 0; i < 24; i++) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

// two roles in one workgroup: waves 0..3 (one per SIMD) run MFMA adds only, waves 4..7 (their SIMD partners) vector adds only
__global__ void k_split(double* out, long long* cyc, int iters, double seed) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double a = ident_a(lane);
    double m[8], v[24];
#pragma unroll
    for (int i = 0; i < 8; i++) m[i] = seed + lane * 1e-9 + i;
#pragma unroll
    for (int i = 0; i < 24; i++) v[i] = seed * 0.5 + lane * 1e-7 + i;
    const double c1 = 1.0000001;
    long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 8; i++) m[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, c1, m[i], 0, 0, 0);
        }
    } else {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 24; i++) v[i] = v[i] + c1;
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += m[i];
#pragma unroll
    for (int i = 0; i < 24; i++) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + wave] = t1 - t0;
}

int main() {
    const int n = 1 << 20;
    std::vector<double> hb(n), hc(n);
    std::mt19937_64 rng(7);
    auto bits = [](uint64_t u) { double d; memcpy(&d, &u, 8); return d; };
    for (int i = 0; i < n; i++) {
        const int kind = i % 8;
        uint64_t x = rng(), y = rng();
        if (kind == 0) { hb[i] = bits(x & 0x7fefffffffffffffull | (x & 0x8000000000000000ull)); hc[i] = bits(y & 0x7fefffffffffffffull | (y & 0x8000000000000000ull)); }
        else if (kind == 1) { hb[i] = (double)(int64_t)(x >> 14) * 0.37; hc[i] = -hb[i] * (1.0 + 1e-15 * (double)(y & 7)); }          // cancellation
        else if (kind == 2) { hb[i] = bits(x & 0x800fffffffffffffull); hc[i] = bits(y & 0x800fffffffffffffull); }                    // denormals
        else if (kind == 3) { hb[i] = (x & 1) ? 0.0 : -0.0; hc[i] = (y & 1) ? 0.0 : -0.0; }                                          // signed zeros
        else if (kind == 4) { hb[i] = ldexp((double)(x >> 11), (int)(y % 80) - 40); hc[i] = ldexp((double)(y >> 11), (int)(x % 80) - 40) * ((x & 2) ? -1 : 1); }
        else if (kind == 5) { hb[i] = (double)(int32_t)x; hc[i] = 6755399441055744.0; }                                               // the truncation magic constant
        else if (kind == 6) { hb[i] = ldexp(1.0, 52) + (double)(x & 0xfffff); hc[i] = 0.5 + (double)(y & 3) * 0.25; }                 // ties
        else { hb[i] = (double)(int64_t)x; hc[i] = (double)(int64_t)y; }
    }
    double *b, *c, *da, *ds;
    CHECK(hipMalloc(&b, n * 8)); CHECK(hipMalloc(&c, n * 8)); CHECK(hipMalloc(&da, n * 8)); CHECK(hipMalloc(&ds, n * 8));
    CHECK(hipMemcpy(b, hb.data(), n * 8, hipMemcpyHostToDevice)); CHECK(hipMemcpy(c, hc.data(), n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_check, dim3(512), dim3(256), 0, 0, b, c, da, ds, n);
    CHECK(hipDeviceSynchronize());
    std::vector<double> ha(n), hs(n);
    CHECK(hipMemcpy(ha.data(), da, n * 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hs.data(), ds, n * 8, hipMemcpyDeviceToHost));
    long bad_add = 0, bad_sub = 0, zero_sign = 0, shown = 0;
    for (int i = 0; i < n; i++) {
        const double ea = hc[i] + hb[i], es = hc[i] - hb[i];
        uint64_t ga, gs, xa, xs;
        memcpy(&ga, &ha[i], 8); memcpy(&gs, &hs[i], 8); memcpy(&xa, &ea, 8); memcpy(&xs, &es, 8);
        const bool nan_ok_a = (ea != ea) && (ha[i] != ha[i]), nan_ok_s = (es != es) && (hs[i] != hs[i]);
        if (ga != xa && !nan_ok_a) { if (ea == 0.0 && ha[i] == 0.0) zero_sign++; else { bad_add++; if (shown++ < 5) printf("add mismatch kind %d: b=%a c=%a got %a want %a\n", i % 8, hb[i], hc[i], ha[i], ea); } }
        if (gs != xs && !nan_ok_s) { if (es == 0.0 && hs[i] == 0.0) zero_sign++; else { bad_sub++; if (shown++ < 10) printf("sub mismatch kind %d: b=%a c=%a got %a want %a\n", i % 8, hb[i], hc[i], hs[i], es); } }
    }
    printf("bit-exactness of MFMA add/sub vs host IEEE add/sub over %d operand pairs: add mismatches %ld, sub mismatches %ld, sign-of-zero-only differences %ld\n", n, bad_add, bad_sub, zero_sign);

    double* out; long long* cyc;
    CHECK(hipMalloc(&out, 256 * 1024 * 8)); CHECK(hipMalloc(&cyc, 8192 * 8));
    const int iters = 20000;
    const char* names[] = {"MFMA adds only (8 per iteration)", "vector adds only (24 per iteration)", "8 MFMA adds + 24 vector adds interleaved in one wave"};
    for (int mode = 0; mode < 3; mode++)
        for (int wps = 1; wps <= 2; wps++) {
            const int threads = 256 * wps;
            hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            auto launch = [&]() {
                if (mode == 0) hipLaunchKernelGGL(k_rate<0>, dim3(256), dim3(threads), 0, 0, out, cyc, iters, 1.0);
                if (mode == 1) hipLaunchKernelGGL(k_rate<1>, dim3(256), dim3(threads), 0, 0, out, cyc, iters, 1.0);
                if (mode == 2) hipLaunchKernelGGL(k_rate<2>, dim3(256), dim3(threads), 0, 0, out, cyc, iters, 1.0);
            };
            launch(); CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0)); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double n_mfma = mode == 1 ? 0 : 8.0 * iters, n_vec = mode == 0 ? 0 : 24.0 * iters;
            printf("%-55s waves/SIMD %d: %.3f ms; per SIMD: %.2f ns per iteration = %.1f cycles @2.4GHz for %g MFMA + %g vector adds per wave-iteration\n",
                   names[mode], wps, ms, ms * 1e6 / iters, ms * 1e6 / iters * 2.4, n_mfma / iters, n_vec / iters);
        }
    {
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_split, dim3(256), dim3(512), 0, 0, out, cyc, iters, 1.0); CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(k_split, dim3(256), dim3(512), 0, 0, out, cyc, iters, 1.0); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        long long hcyc[8]; CHECK(hipMemcpy(hcyc, cyc, 64, hipMemcpyDeviceToHost));
        printf("one MFMA-only wave + one vector-only wave per SIMD: %.3f ms total; ticks per iteration: MFMA waves %.1f (8 MFMA adds), vector waves %.1f (24 vector adds)\n",
               ms, (double)hcyc[0] / iters, (double)hcyc[4] / iters);
    }
    return 0;
}
