// Micro-benchmark (round 3): aggregate throughput of the 512-point FP64 transform (the path's device functions, rtfhe_device.hpp)
// per CU by residency: W waves per CU, each transforming private register data in a loop (forward + inverse, LDS exchanges
// included, no synchronisation between waves).  Question: how much would 3 or 4 resident waves per SIMD (<= 168 / 128 VGPRs each)
// buy over the 2 the two-waves-per-gate kernel has?  Prints transforms per microsecond per CU and the VGPR use of each variant.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../rustfhe_amd/csrc/rtfhe_device.hpp"

using namespace rtfhe;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int WAVES, bool DUAL, int MINW>
__global__ __launch_bounds__(64 * WAVES, MINW) void k_loop(const cplx* gtw, double* out, int iters) {
    typedef Geo<10> G;
    extern __shared__ __align__(16) unsigned char smem[];
    cplx* tw = reinterpret_cast<cplx*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int idx = tid; idx < G::TW_TOTAL; idx += 64 * WAVES) tw[idx] = gtw[idx];
    __syncthreads();
    double* xbuf = reinterpret_cast<double*>(tw + G::TW_TOTAL) + (size_t)wave * G::XSLOTS * (DUAL ? 2 : 1);
    double re[G::R], im[G::R];
#pragma unroll
    for (int m = 0; m < G::R; m++) { re[m] = lane + m; im[m] = lane - m; }
    for (int it = 0; it < iters; it++) {
        fft_forward<10, DUAL>(re, im, tw, xbuf, lane);
        fft_inverse<10, DUAL>(re, im, tw + G::TW_DIR, tw + G::TW_DIR, xbuf, lane);
    }
    double s = 0;
#pragma unroll
    for (int m = 0; m < G::R; m++) s += re[m] + im[m];
    out[blockIdx.x * 64 * WAVES + tid] = s;
}

// The batch kernel's per-step transform mix per wave: NR forward rows side by side (interleaved exchanges) + INV_OF of every INV_PER waves one inverse.
// 2 waves/SIMD with NR = 3 and an inverse in every wave is what k_bootstrap_pair runs; 3 waves/SIMD with NR = 2 and an inverse in 2 of 3 waves is the
// same work per gate on three waves (<= 168 registers each).
template <int WAVES, int NR, int INV_OF, int INV_PER, int MINW>
__global__ __launch_bounds__(64 * WAVES, MINW) void k_loop_multi(const cplx* gtw, double* out, int iters) {
    typedef Geo<10> G;
    extern __shared__ __align__(16) unsigned char smem[];
    cplx* tw = reinterpret_cast<cplx*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int idx = tid; idx < G::TW_TOTAL; idx += 64 * WAVES) tw[idx] = gtw[idx];
    __syncthreads();
    double* xbuf = reinterpret_cast<double*>(tw + G::TW_TOTAL) + (size_t)wave * G::XSLOTS * 2;
    double re[NR][G::R], im[NR][G::R];
#pragma unroll
    for (int j = 0; j < NR; j++)
#pragma unroll
        for (int m = 0; m < G::R; m++) { re[j][m] = lane + m + j; im[j][m] = lane - m - j; }
    const bool inv = (wave % INV_PER) < INV_OF;
    for (int it = 0; it < iters; it++) {
        fft_forward_multi_a<10, NR, true, NoHook, true>(re, im, tw, xbuf, xbuf + G::XSLOTS, lane);
        fft_forward_multi_b<10, NR, true>(re, im, tw);
        if (inv) fft_inverse<10, true, true>(re[0], im[0], tw + G::TW_DIR, tw + G::TW_DIR, xbuf, lane, xbuf + G::XSLOTS);
    }
    double s = 0;
#pragma unroll
    for (int j = 0; j < NR; j++)
#pragma unroll
        for (int m = 0; m < G::R; m++) s += re[j][m] + im[j][m];
    out[blockIdx.x * 64 * WAVES + tid] = s;
}

template <int WAVES, int NR, int INV_OF, int INV_PER, int MINW>
int run_multi(const cplx* tw, double* out, const char* name) {
    typedef Geo<10> G;
    const size_t lds = (size_t)G::TW_TOTAL * sizeof(cplx) + (size_t)WAVES * G::XSLOTS * 8 * 2;
    auto k = k_loop_multi<WAVES, NR, INV_OF, INV_PER, MINW>;
    CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipFuncAttributes fa; CHECK(hipFuncGetAttributes(&fa, (const void*)k));
    const int iters = 1500;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(256), dim3(64 * WAVES), lds, 0, tw, out, iters); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(k, dim3(256), dim3(64 * WAVES), lds, 0, tw, out, iters); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double tr = (double)iters * WAVES * (NR + (double)INV_OF / INV_PER);     // transforms per CU
    printf("%-60s regs %3d  LDS %6zu B: %.3f ms, %.2f transforms/us/CU, %.2f cycles@2.4GHz per FP64 instruction per SIMD (360 per transform)\n",
           name, fa.numRegs, lds, ms, tr / (ms * 1e3), ms * 1e-3 * 2.4e9 / (tr / 4 * 360));
    return 0;
}

template <int WAVES, bool DUAL, int MINW>
int run(const cplx* tw, double* out, const char* name) {
    typedef Geo<10> G;
    const size_t lds = (size_t)G::TW_TOTAL * sizeof(cplx) + (size_t)WAVES * G::XSLOTS * 8 * (DUAL ? 2 : 1);
    auto k = k_loop<WAVES, DUAL, MINW>;
    CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipFuncAttributes fa; CHECK(hipFuncGetAttributes(&fa, (const void*)k));
    const int iters = 2000;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(256), dim3(64 * WAVES), lds, 0, tw, out, iters); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(k, dim3(256), dim3(64 * WAVES), lds, 0, tw, out, iters); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double tr = 2.0 * iters * WAVES;     // transforms per CU
    printf("%-44s regs %3d  LDS %6zu B: %.3f ms, %.2f transforms/us/CU, %.2f cycles@2.4GHz per FP64 instruction per SIMD (360 per transform)\n",
           name, fa.numRegs, lds, ms, tr / (ms * 1e3), ms * 1e-3 * 2.4e9 / (tr / 4 * 360));
    return 0;
}

int main() {
    typedef Geo<10> G;
    cplx* tw; double* out;
    CHECK(hipMalloc(&tw, G::TW_TOTAL * sizeof(cplx))); CHECK(hipMalloc(&out, 256 * 1024 * 8));
    cplx* h = new cplx[G::TW_TOTAL];
    for (int i = 0; i < G::TW_TOTAL; i++) h[i] = make_double2(0.7 + 1e-4 * (i % 97), 0.7 - 1e-4 * (i % 89));
    CHECK(hipMemcpy(tw, h, G::TW_TOTAL * sizeof(cplx), hipMemcpyHostToDevice));
    if (run<4, true, 1>(tw, out, "1 wave/SIMD, two exchange buffers")) return 1;
    if (run<8, true, 2>(tw, out, "2 waves/SIMD, two exchange buffers")) return 1;
    if (run<8, false, 2>(tw, out, "2 waves/SIMD, one exchange buffer")) return 1;
    if (run<12, false, 3>(tw, out, "3 waves/SIMD, one exchange buffer")) return 1;
    if (run<16, false, 4>(tw, out, "4 waves/SIMD, one exchange buffer")) return 1;
    if (run<12, true, 3>(tw, out, "3 waves/SIMD, two exchange buffers")) return 1;
    if (run_multi<8, 3, 1, 1, 2>(tw, out, "2 waves/SIMD, 3 forward rows + 1 inverse per wave (pair)")) return 1;
    if (run_multi<12, 2, 2, 3, 3>(tw, out, "3 waves/SIMD, 2 forward rows + inverse in 2 of 3 waves")) return 1;
    if (run_multi<12, 2, 1, 1, 3>(tw, out, "3 waves/SIMD, 2 forward rows + 1 inverse per wave")) return 1;
    if (run_multi<8, 2, 1, 1, 2>(tw, out, "2 waves/SIMD, 2 forward rows + 1 inverse per wave")) return 1;
    return 0;
}
