// Micro-benchmark: does a wavefront whose upper 32 lanes are masked off issue FP64 VALU instructions faster?
// (v_add_f64 loop; full EXEC vs EXEC = lanes 0..31 vs EXEC = lanes 0..15), 1 and 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int ACTIVE>
__global__ void k_issue(double* out, int iters, double seed) {
    double a[16];
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = seed + threadIdx.x * 1e-9 + i;
    const double c1 = 1.0000001;
    if ((threadIdx.x & 63) < ACTIVE) {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 16; i++) a[i] = a[i] + c1;
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    double* out;
    CHECK(hipMalloc(&out, 256 * 1024 * 8));
    const int iters = 20000;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int wps = 1; wps <= 2; wps++)
        for (int act = 64; act >= 16; act /= 2) {
            const int threads = 256 * wps;
            auto launch = [&]() {
                if (act == 64) hipLaunchKernelGGL(k_issue<64>, dim3(256), dim3(threads), 0, 0, out, iters, 1.0);
                else if (act == 32) hipLaunchKernelGGL(k_issue<32>, dim3(256), dim3(threads), 0, 0, out, iters, 1.0);
                else hipLaunchKernelGGL(k_issue<16>, dim3(256), dim3(threads), 0, 0, out, iters, 1.0);
            };
            launch(); CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0)); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double ops = (double)iters * 16;
            printf("v_add_f64, %2d active lanes, waves/SIMD %d: %.3f ms, %.2f ns per wave-instr per SIMD (= %.2f cyc @2.4GHz)\n", act, wps, ms,
                   ms * 1e6 / (ops * wps), ms * 1e6 / (ops * wps) * 2.4);
        }
    return 0;
}
