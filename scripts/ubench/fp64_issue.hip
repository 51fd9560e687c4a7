// Micro-benchmark: issue rate of v_add_f64 / v_mul_f64 / v_fma_f64 / v_add_f32 / int ops per SIMD at 1, 2, 4 waves per SIMD,
// and of LDS b64 writes/reads.  Prints cycles per wave-instruction per SIMD (wall-clock based, assumes 2.4 GHz nominal; also prints s_memtime cycles).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int OP>
__global__ void k_issue(double* out, long long* cyc, int iters, double seed) {
    double a[16];
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = seed + threadIdx.x * 1e-9 + i;
    float f[16];
#pragma unroll
    for (int i = 0; i < 16; i++) f[i] = (float)a[i];
    int q[16];
#pragma unroll
    for (int i = 0; i < 16; i++) q[i] = threadIdx.x + i;
    const double c1 = 1.0000001, c2 = 0.9999999;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (OP == 0) a[i] = a[i] + c1;
            if (OP == 1) a[i] = a[i] * c2;
            if (OP == 2) a[i] = __builtin_fma(a[i], c2, c1);
            if (OP == 3) f[i] = f[i] + 1.0000001f;
            if (OP == 4) q[i] = q[i] * 3 + 1;            // v_mad_u32_u24 / mul_lo
            if (OP == 5) q[i] = (q[i] ^ 0x55) + i;       // 2 int ops
            if (OP == 6) { a[i] = a[i] + c1; q[i] = (q[i] ^ 0x55) + i; }   // f64 add + 2 int
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; float sf = 0; int sq = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) { s += a[i]; sf += f[i]; sq += q[i]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + sf + sq;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
__global__ void k_lds(double* out, long long* cyc, int iters) {
    extern __shared__ double sm[];
    double* my = sm + (threadIdx.x >> 6) * 1024;
    const int lane = threadIdx.x & 63;
    double v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = lane + i;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if (OP == 0 || OP == 2) {
#pragma unroll
            for (int i = 0; i < 8; i++) my[lane + 72 * i] = v[i];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (OP == 1 || OP == 2) {
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] += my[9 * lane + i];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    double* out; long long* cyc;
    CHECK(hipMalloc(&out, 256 * 1024 * 8)); CHECK(hipMalloc(&cyc, 1024 * 8));
    const int iters = 20000;
    const char* names[] = {"v_add_f64", "v_mul_f64", "v_fma_f64", "v_add_f32", "int mad", "int xor+add (2 ops)", "f64 add + 2 int (3 ops)"};
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int op = 0; op < 7; op++)
        for (int wps = 1; wps <= 4; wps *= 2) {
            const int threads = 256 * wps;   // wps waves per SIMD, one block per CU
            auto launch = [&]() {
                switch (op) {
                    case 0: hipLaunchKernelGGL(k_issue<0>, dim3(256), dim3(threads), 0, 0, out, cyc, iters, 1.0); break;
                    case 1: hipLaunchKernelGGL(k_issue<1>, dim3(256), dim3(threads), 0, 0, out, cyc, iters, 1.0); break;
                    case 2: hipLaunchKernelGGL(k_issue<2>, dim3(256), dim3(threads), 0, 0, out, cyc, iters, 1.0); break;
                    case 3: hipLaunchKernelGGL(k_issue<3>, dim3(256), dim3(threads), 0, 0, out, cyc, iters, 1.0); break;
                    case 4: hipLaunchKernelGGL(k_issue<4>, dim3(256), dim3(threads), 0, 0, out, cyc, iters, 1.0); break;
                    case 5: hipLaunchKernelGGL(k_issue<5>, dim3(256), dim3(threads), 0, 0, out, cyc, iters, 1.0); break;
                    default: hipLaunchKernelGGL(k_issue<6>, dim3(256), dim3(threads), 0, 0, out, cyc, iters, 1.0); break;
                }
            };
            launch(); CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0)); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            long long c; CHECK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
            const double ops_per_wave = (double)iters * 16 * (op == 5 ? 2 : op == 6 ? 3 : 1);
            printf("%-26s waves/SIMD %d: %.2f ms, memtime %.2f ticks per wave-instr, wall %.2f ns per instr per SIMD (= %.2f cyc @2.4GHz)\n", names[op], wps, ms,
                   (double)c / ops_per_wave, ms * 1e6 / (ops_per_wave * wps), ms * 1e6 / (ops_per_wave * wps) * 2.4);
        }
    const char* ln[] = {"ds_write_b64 x8 (stride-9KB map)", "ds_read_b64 x8", "write x8 + read x8"};
    for (int op = 0; op < 3; op++)
        for (int w = 1; w <= 8; w *= 2) {     // w waves per CU
            auto launch = [&]() {
                switch (op) {
                    case 0: hipLaunchKernelGGL(k_lds<0>, dim3(256), dim3(64 * w), w * 8192, 0, out, cyc, iters); break;
                    case 1: hipLaunchKernelGGL(k_lds<1>, dim3(256), dim3(64 * w), w * 8192, 0, out, cyc, iters); break;
                    default: hipLaunchKernelGGL(k_lds<2>, dim3(256), dim3(64 * w), w * 8192, 0, out, cyc, iters); break;
                }
            };
            launch(); CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0)); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double n_inst = (double)iters * 8 * (op == 2 ? 2 : 1);
            printf("%-34s waves/CU %d: %.2f ms, %.1f ns per LDS wave-instr per wave, CU total %.2f cyc@2.4GHz per wave-instr\n", ln[op], w, ms,
                   ms * 1e6 / n_inst, ms * 1e6 / (n_inst * w) * 2.4);
        }
    return 0;
}
