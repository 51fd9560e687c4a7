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
    return 0;
}
