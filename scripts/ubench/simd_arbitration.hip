// Micro-benchmark: how does a SIMD arbitrate between two resident waves that both have FP64 work?
// One 8-wave workgroup per CU (waves w and w + 4 share SIMD w); every wave runs the same v_add_f64 loop and records its
// own elapsed time.  Cases: equal priorities; waves 4..7 raised with s_setprio 3.  Printed per wave: time / time alone.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>   // 0: equal priorities, 1: waves 4..7 at priority 3, 2: only waves 0..3 run (baseline "alone")
__global__ void k_arb(double* out, long long* t, int iters, double seed) {
    const int wave = threadIdx.x >> 6;
    double a[16];
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = seed + threadIdx.x * 1e-9 + i;
    const double c1 = 1.0000001;
    if (MODE == 1 && wave >= 4) __builtin_amdgcn_s_setprio(3);
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (!(MODE == 2 && wave >= 4)) {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 16; i++) a[i] = a[i] + c1;
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += a[i];
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) t[wave] = t1 - t0;
}

int main() {
    double* out; long long* t;
    CHECK(hipMalloc(&out, 256 * 512 * 8)); CHECK(hipMalloc(&t, 64));
    const int iters = 20000;
    long long h[3][8];
    for (int mode = 0; mode < 3; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            if (mode == 0) hipLaunchKernelGGL(k_arb<0>, dim3(256), dim3(512), 0, 0, out, t, iters, 1.0);
            if (mode == 1) hipLaunchKernelGGL(k_arb<1>, dim3(256), dim3(512), 0, 0, out, t, iters, 1.0);
            if (mode == 2) hipLaunchKernelGGL(k_arb<2>, dim3(256), dim3(512), 0, 0, out, t, iters, 1.0);
            CHECK(hipDeviceSynchronize());
        }
        CHECK(hipMemcpy(h[mode], t, 64, hipMemcpyDeviceToHost));
    }
    const double alone = (double)h[2][0];
    const char* names[] = {"equal priorities            ", "waves 4..7 at s_setprio 3   "};
    for (int mode = 0; mode < 2; mode++) {
        printf("%s time / time alone, waves 0..7:", names[mode]);
        for (int w = 0; w < 8; w++) printf(" %.2f", (double)h[mode][w] / alone);
        printf("\n");
    }
    printf("(alone: %.2f memtime ticks per FP64 wave-instruction)\n", alone / ((double)iters * 16));
    return 0;
}
