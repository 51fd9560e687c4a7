// Micro-benchmark (round 5): does ANY form of an FP64 vector instruction issue faster than the 4.8 cycles per wave-instruction that v_add_f64 /
// v_mul_f64 streams reach on gfx950 (fp64_issue.hip)?  Straight-line streams of 512 instructions per loop iteration (loop overhead < 1 %), 16
// independent destinations, W waves per SIMD (1, 2, 4, 8), one workgroup of 4 W waves per CU.  Forms: operands in the same / in different VGPR banks,
// one scalar operand, an inline constant, the 4-byte VOP2 encoding (v_fmac_f64_e32), FMA, 64-bit moves.
// Prints cycles per wave-instruction and SIMD from the launch's wall time at the nominal 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int FORM>
__global__ void k_form(double* out, int iters, double seed) {
    double a[16], b[8];
#pragma unroll
    for (int i = 0; i < 16; i++) { a[i] = seed + threadIdx.x * 1e-9 + i; b[i & 7] = 1.0 + 1e-9 * (i + 1); }
    double sc = seed * 0.5 + 1.0;       // wave-uniform -> SGPR pair
    sc = __longlong_as_double(((long long)__builtin_amdgcn_readfirstlane((int)(__double_as_longlong(sc) >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)__double_as_longlong(sc)));
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 32; rep++) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (FORM == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b[i & 7]));             // a[i], b[i & 7]: an even number of register pairs apart
                if (FORM == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 15]));      // neighbours: 2 registers apart (other banks)
                if (FORM == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "s"(sc));                   // one scalar operand
                if (FORM == 3) asm volatile("v_add_f64 %0, %0, 1.0" : "+v"(a[i]));                            // inline constant
                if (FORM == 4) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b[i & 7]));
                if (FORM == 5) asm volatile("v_fmac_f64_e32 %0, 1.0, %1" : "+v"(a[i]) : "v"(b[i & 7]));           // VOP2 (4 bytes): a += 1.0 * b  (an exact addition)
                if (FORM == 6) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i & 7]), "v"(b[(i + 3) & 7]));
                if (FORM == 7) asm volatile("v_mov_b64 %0, %1" : "=v"(a[i]) : "v"(b[i & 7]));
                if (FORM == 8) asm volatile("v_add_f64 %0, %1, %2" : "=v"(a[i]) : "v"(b[i & 7]), "v"(b[(i + 5) & 7]));    // no read of the destination
                if (FORM == 9) asm volatile("v_add_f64 %0, -%0, %1" : "+v"(a[i]) : "v"(b[i & 7]));               // a source modifier
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += a[i] + b[i & 7];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int FORM>
int run(const char* name, double* out, int cus) {
    const int iters = 400;
    for (int w = 1; w <= 8; w *= 2) {
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        const int block = 256 * w > 1024 ? 1024 : 256 * w, grid = cus * (256 * w / block);     // 4 w waves per CU
        hipLaunchKernelGGL(k_form<FORM>, dim3(grid), dim3(block), 0, 0, out, 10, 1.0);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_form<FORM>, dim3(grid), dim3(block), 0, 0, out, iters, 1.0);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double instr_per_simd = (double)iters * 512 * w;
        printf("%-44s waves/SIMD %d: %7.3f ms  %.2f cycles@2.4GHz per wave-instruction and SIMD\n", name, w, ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
    }
    return 0;
}

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    double* out; CHECK(hipMalloc(&out, (size_t)cus * 4096 * 8));
    if (run<0>("v_add_f64 v, v, v (same banks)", out, cus)) return 1;
    if (run<1>("v_add_f64 v, v, v (other banks)", out, cus)) return 1;
    if (run<2>("v_add_f64 v, v, s", out, cus)) return 1;
    if (run<3>("v_add_f64 v, v, 1.0", out, cus)) return 1;
    if (run<4>("v_mul_f64 v, v, v", out, cus)) return 1;
    if (run<5>("v_fmac_f64_e32 v, 1.0, v (VOP2)", out, cus)) return 1;
    if (run<6>("v_fma_f64 v, v, v, v", out, cus)) return 1;
    if (run<7>("v_mov_b64 v, v", out, cus)) return 1;
    if (run<8>("v_add_f64 d, v, v (destination not read)", out, cus)) return 1;
    if (run<9>("v_add_f64 v, -v, v (source modifier)", out, cus)) return 1;
    return 0;
}
