// rtfhe_kernels_anyn.hpp -- the reference's two transforms for ANY power of two 16 <= N <= 2048.
//
// The gate path runs at N = 1024 / 2048 on the register-resident wave transforms of rtfhe_device.hpp.  The reference's FFT FFI, however,
// accepts every power of two >= 16 (Spqlios::new, utils/src/spqlios.rs:40-50) and its own unit test uses N = 16 (spqlios.rs:243-276), so
// the library must answer those too.  Speed is irrelevant off the gate path: ONE workgroup per polynomial, the N/2 complex points in LDS,
// every radix-2 stage of the reference network as one sweep of the workgroup over the stage's N/4 butterflies with a barrier behind it.
// The butterflies, their operand order and their individually rounded products and sums (-ffp-contract=off) are the reference's:
//   forward = ifft_model  utils/src/spqlios/spqlios-fft-impl.cpp:469-641 (asm `ifft`, spqlios-ifft-avx.s:64-272):
//             twist by (cos, sin)(2 pi j / 2N) :496-518; stages halfnn = N/4 ... 4: x0 + x1, (x0 - x1) w :526-572; size 4 :575-603; size 2 :606-634
//   inverse = fft_model   spqlios-fft-impl.cpp:204-397 (asm `fft`, spqlios-fft-avx.s:79-280):
//             input times 2/N (fft_processor_spqlios.cpp:158); size 2 :248-269; size 4 :289-310; stages halfnn = 4 ... N/4: t = x1 w, x0 +- t
//             :315-363; untwist by (cos, sin)(-2 pi j / 2N) :374-396; Torus32(int64_t(x)) fft_processor_spqlios.cpp:182
// Tables: the host's natural-order arrays (HostTw in rtfhe_host.hpp / rtfhe_twiddles.hip: same values as new_ifft_table / new_fft_table), eight arrays of P = N/2
// doubles: twist cos / sin, untwist cos / sin, forward stages cos / sin (stage halfnn at P - 2 halfnn), inverse stages cos / sin (at halfnn - 4).
#pragma once

#include "rtfhe_device.hpp"

namespace rtfhe {

enum AnyNMode : int32_t {
    ANYN_IFFT_I32 = 0,   // int32[N]  -> double[N]   Spqlios_ifft_i32 / _ifft_u32 (execute_reverse_int, fft_processor_spqlios.cpp:58-106)
    ANYN_IFFT_F64 = 1,   // double[N] -> double[N]   Spqlios_ifft     (execute_reverse)
    ANYN_FFT_U32 = 2,    // double[N] -> uint32[N]   Spqlios_fft_u32  (execute_direct_torus32, :156-183)
    ANYN_FFT_F64 = 3,    // double[N] -> double[N]   Spqlios_fft      (execute_direct, :108-154)
    ANYN_POLY_MUL = 4,   // uint32[N] x uint32[N] -> uint32[N]   Spqlios_poly_mul (spqlios-wrapper.cpp:38-53)
};

struct AnyNArgs {
    const double* tab;   // [8][P]
    const void* in;      // [count][N]
    const void* in2;     // [count][N], ANYN_POLY_MUL only
    void* out;           // [count][N]
    int32_t N, count, mode;
};

constexpr int ANYN_THREADS = 256;
constexpr int ANYN_MAXP = 1024;

// all threads of the workgroup; re / im hold P points; ends with a barrier
__device__ __forceinline__ void anyn_forward(double* re, double* im, const double* __restrict__ tab, int P, int tid) {
    const double* tc = tab;
    const double* ts = tab + P;
    const double* fc = tab + 4 * P;
    const double* fs = tab + 5 * P;
    for (int j = tid; j < P; j += ANYN_THREADS) {
        const double r = re[j], i = im[j], c = tc[j], s = ts[j];
        const double rc = r * c, ic = i * c, rs = r * s, is = i * s;
        re[j] = rc - is;
        im[j] = ic + rs;
    }
    __syncthreads();
    for (int halfnn = P / 2; halfnn >= 4; halfnn >>= 1) {
        const double* c = fc + (P - 2 * halfnn);
        const double* s = fs + (P - 2 * halfnn);
        for (int b = tid; b < P / 2; b += ANYN_THREADS) {
            const int k = b & (halfnn - 1), i0 = ((b - k) << 1) + k, i1 = i0 + halfnn;
            const double sr = re[i0] + re[i1], si = im[i0] + im[i1];
            const double dr = re[i0] - re[i1], di = im[i0] - im[i1];
            re[i0] = sr; im[i0] = si;
            double p = dr * c[k], q = di * s[k];
            re[i1] = p - q;
            p = dr * s[k]; q = di * c[k];
            im[i1] = p + q;
        }
        __syncthreads();
    }
    for (int g = tid; g < P / 4; g += ANYN_THREADS) {
        const int m = 4 * g;
        const double r0 = re[m], r1 = re[m + 1], r2 = re[m + 2], r3 = re[m + 3];
        const double j0 = im[m], j1 = im[m + 1], j2 = im[m + 2], j3 = im[m + 3];
        re[m] = r0 + r2; re[m + 1] = r1 + r3; re[m + 2] = r0 + (-r2); re[m + 3] = (-j1) + j3;
        im[m] = j0 + j2; im[m + 1] = j1 + j3; im[m + 2] = j0 + (-j2); im[m + 3] = r1 + (-r3);
    }
    __syncthreads();
    for (int g = tid; g < P / 2; g += ANYN_THREADS) {
        const int m = 2 * g;
        const double r0 = re[m], r1 = re[m + 1], j0 = im[m], j1 = im[m + 1];
        re[m] = r0 + r1; re[m + 1] = r0 + (-r1);
        im[m] = j0 + j1; im[m + 1] = j0 + (-j1);
    }
    __syncthreads();
}

// in: re / im already scaled by 2/N
__device__ __forceinline__ void anyn_inverse(double* re, double* im, const double* __restrict__ tab, int P, int tid) {
    const double* uc = tab + 2 * P;
    const double* us = tab + 3 * P;
    const double* ic_ = tab + 6 * P;
    const double* is_ = tab + 7 * P;
    for (int g = tid; g < P / 2; g += ANYN_THREADS) {
        const int m = 2 * g;
        const double r0 = re[m], r1 = re[m + 1], j0 = im[m], j1 = im[m + 1];
        re[m] = r0 + r1; re[m + 1] = r0 + (-r1);
        im[m] = j0 + j1; im[m + 1] = j0 + (-j1);
    }
    __syncthreads();
    for (int g = tid; g < P / 4; g += ANYN_THREADS) {
        const int m = 4 * g;
        const double r0 = re[m], r1 = re[m + 1], r2 = re[m + 2], r3 = re[m + 3];
        const double j0 = im[m], j1 = im[m + 1], j2 = im[m + 2], j3 = im[m + 3];
        re[m] = r0 + r2; re[m + 1] = r1 + j3;    re[m + 2] = r0 + (-r2); re[m + 3] = r1 + (-j3);
        im[m] = j0 + j2; im[m + 1] = j1 + (-r3); im[m + 2] = j0 + (-j2); im[m + 3] = j1 + r3;
    }
    __syncthreads();
    for (int halfnn = 4; halfnn < P; halfnn <<= 1) {
        const double* c = ic_ + (halfnn - 4);
        const double* s = is_ + (halfnn - 4);
        for (int b = tid; b < P / 2; b += ANYN_THREADS) {
            const int k = b & (halfnn - 1), i0 = ((b - k) << 1) + k, i1 = i0 + halfnn;
            const double t0 = re[i1] * c[k], t1 = re[i1] * s[k], t2 = im[i1] * c[k], t3 = im[i1] * s[k];
            const double tr = t0 - t3, ti = t1 + t2;
            const double ar = re[i0], ai = im[i0];
            re[i0] = ar + tr; im[i0] = ai + ti;
            re[i1] = ar - tr; im[i1] = ai - ti;
        }
        __syncthreads();
    }
    for (int j = tid; j < P; j += ANYN_THREADS) {
        const double r = re[j], i = im[j], c = uc[j], s = us[j];
        const double rc = r * c, ic = i * c, rs = r * s, is = i * s;
        re[j] = rc - is;
        im[j] = ic + rs;
    }
    __syncthreads();
}

__global__ __launch_bounds__(ANYN_THREADS) void k_fft_anyn(const AnyNArgs a) {
    __shared__ double re[ANYN_MAXP], im[ANYN_MAXP], re2[ANYN_MAXP], im2[ANYN_MAXP];
    const int tid = threadIdx.x, N = a.N, P = N / 2;
    const double two_over_n = 2.0 / (double)N;       // fft_processor_spqlios.cpp:110,158 (a power of two: the product is exact)
    for (int g = blockIdx.x; g < a.count; g += gridDim.x) {
        const size_t base = (size_t)g * N;
        if (a.mode == ANYN_IFFT_I32 || a.mode == ANYN_POLY_MUL) {
            const int32_t* s = static_cast<const int32_t*>(a.in) + base;     // torus words are reinterpreted as signed (:100-106)
            for (int j = tid; j < P; j += ANYN_THREADS) { re[j] = (double)s[j]; im[j] = (double)s[j + P]; }
        } else {
            const double* s = static_cast<const double*>(a.in) + base;
            const double f = (a.mode == ANYN_IFFT_F64) ? 1.0 : two_over_n;
            for (int j = tid; j < P; j += ANYN_THREADS) { re[j] = s[j] * f; im[j] = s[j + P] * f; }
        }
        __syncthreads();
        if (a.mode == ANYN_IFFT_I32 || a.mode == ANYN_IFFT_F64) {
            anyn_forward(re, im, a.tab, P, tid);
        } else if (a.mode == ANYN_POLY_MUL) {
            const int32_t* s2 = static_cast<const int32_t*>(a.in2) + base;
            for (int j = tid; j < P; j += ANYN_THREADS) { re2[j] = (double)s2[j]; im2[j] = (double)s2[j + P]; }
            __syncthreads();
            anyn_forward(re, im, a.tab, P, tid);
            anyn_forward(re2, im2, a.tab, P, tid);
            for (int j = tid; j < P; j += ANYN_THREADS) {        // spqlios-wrapper.cpp:45-50, then execute_direct_torus32's input scaling
                const double aimbim = im[j] * im2[j], arebim = re[j] * im2[j], p = re[j] * re2[j], q = im[j] * re2[j];
                re[j] = (p - aimbim) * two_over_n;
                im[j] = (q + arebim) * two_over_n;
            }
            __syncthreads();
            anyn_inverse(re, im, a.tab, P, tid);
        } else {
            anyn_inverse(re, im, a.tab, P, tid);
        }
        if (a.mode == ANYN_FFT_U32 || a.mode == ANYN_POLY_MUL) {
            uint32_t* o = static_cast<uint32_t*>(a.out) + base;
            for (int j = tid; j < P; j += ANYN_THREADS) { o[j] = trunc_to_torus_wide(re[j]); o[j + P] = trunc_to_torus_wide(im[j]); }
        } else {
            double* o = static_cast<double*>(a.out) + base;
            for (int j = tid; j < P; j += ANYN_THREADS) { o[j] = re[j]; o[j + P] = im[j]; }
        }
        __syncthreads();
    }
}

}  // namespace rtfhe
