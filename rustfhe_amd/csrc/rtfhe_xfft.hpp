// rtfhe_xfft.hpp -- device building blocks of the SPLIT-FFT EXACT backend (RTFHE_BACKEND_FFT_SPLIT_EXACT): exact negacyclic products
// (the semantics of the reference's Polynomial::cross, utils/src/math.rs:238-257, KAT :761-843) through an FMA-contracted FP64 FFT.
//
// An FP64 FFT cannot return  S = sum_r D_r (*) K_r  exactly (|S| < 2^48.6 against a transform error of ~0.3: the reference's own +-1 LSB,
// SURVEY App. B P1).  The key is therefore split ONCE, at load time, into signed 16-bit halves  K = 2^16 hi + lo,  and
//       S = 2^16 * (sum_r D_r (*) hi_r) + (sum_r D_r (*) lo_r)         each sum below 6 N 32 2^15 = 2^33.6 (N = 1024)
// comes out of two transforms whose error is proven below 2^-8 for EVERY input (scripts/xfft/model.py: l2 error analysis, worst-case
// patterns 2^-17, random inputs 2^-24); a magic-number addition rounds each to the nearest integer and the halves are recombined mod 2^32
// in integer registers.  Exact arithmetic carries NO rounding DAG to mirror, so everything the mirror backend may not do is done here:
//   * every butterfly is  a' = a + w b, b' = a - w b = 2 a - a'  in 6 v_fma_f64 (the mirror: 10 separately rounded instructions);
//   * the forward transform runs natural-order-in -> bit-reversed-out with the negacyclic twist MERGED into its twiddles (ring C[X]/(X^n - i),
//     stage s / block B twiddle exp(i theta/2), theta_root = pi/2, children theta/2 and theta/2 + pi): no twist pass, and the twiddles of the
//     first in-register pass are seven wave-uniform constants (scalar operands);
//   * the inverse is a plain radix-2 DIT on the bit-reversed spectrum: its first in-register pass multiplies by 1, -i and (+-1 - i)/sqrt 2
//     only; ONE untwist multiply by psi^-j / n is fused with the rounding (two FMAs per real output);
//   * the fold over the six key rows has no order: the two waves of a gate accumulate their own three rows for BOTH output polynomials and
//     swap partial sums (rtfhe_kernels_xfft.hpp).
// Layouts and exchanges are the mirror's (Geo<10>, exchange<>, XAffine: lane / register <-> point maps L1, L2, L3 of rtfhe_device.hpp).
#pragma once

#include "rtfhe_device.hpp"

namespace rtfhe {
namespace xfft {

// device twiddle table, cplx units (512-point wave transform; host builder: xfft_device_table, rtfhe_dispatch_xfft.hip)
//   forward entry e of a pass: e = 0: the stage pairing register bit 2 (one twiddle per lane), e = 1 + q: register bit 1 (q = m >> 2),
//                              e = 3 + q: register bit 0 (q = m >> 1)                          -- block twiddles, q = the register bits ABOVE the pair
//   inverse entry e of a pass: e = 0: register bit 0, e = 1 + q: bit 1 (q = m & 1), e = 3 + q: bit 2 (q = m & 3)   -- q = the register bits BELOW
struct XTw {
    static constexpr int F1 = 0;              // [7] (+ 1 pad)   forward pass 1: wave-uniform
    static constexpr int F2 = 8;              // [7][8]          forward pass 2: by lane >> 3
    static constexpr int F3 = F2 + 7 * 8;     // [7][64]         forward pass 3: by lane
    static constexpr int I2 = F3 + 7 * 64;    // [7][8]          inverse pass 2: by lane & 7
    static constexpr int I3 = I2 + 7 * 8;     // [7][64]         inverse pass 3: by lane
    static constexpr int UT = I3 + 7 * 64;    // [8][64]         untwist psi^-(lane + 64 m) / n
    static constexpr int TOTAL = UT + 8 * 64;
};

constexpr int R = 8;
constexpr double SQRT_HALF = 0.70710678118654752440;
// 1.5 * 2^36 + 0.5: x + MAGIC has its unit in the last place at 2^-16 for |x| < 2^35, so mantissa bits [16, 48) hold floor(x + 1/2) mod 2^32
constexpr double MAGIC = 103079215104.5;

// a' = a + w b ; b' = a - w b = 2 a - a'
__device__ __forceinline__ void bfly(double& ar, double& ai, double& br, double& bi, double wr, double wi) {
    const double tr = __builtin_fma(wr, br, __builtin_fma(-wi, bi, ar));
    const double ti = __builtin_fma(wr, bi, __builtin_fma(wi, br, ai));
    br = __builtin_fma(2.0, ar, -tr);
    bi = __builtin_fma(2.0, ai, -ti);
    ar = tr; ai = ti;
}

// forward stage on register bit MB: pairs (m, m | h), twiddle w[q], q = m >> (MB + 1)
template <int MB>
__device__ __forceinline__ void fwd_stage(double (&re)[R], double (&im)[R], const cplx* w) {
    constexpr int h = 1 << MB;
#pragma unroll
    for (int m = 0; m < R; m++) {
        if (m & h) continue;
        const int q = m >> (MB + 1);
        bfly(re[m], im[m], re[m | h], im[m | h], w[q].x, w[q].y);
    }
}
// the three stages of one forward pass; w: 7 entries (XTw)
__device__ __forceinline__ void fwd_pass(double (&re)[R], double (&im)[R], const cplx* w) {
    fwd_stage<2>(re, im, w);
    fwd_stage<1>(re, im, w + 1);
    fwd_stage<0>(re, im, w + 3);
}

// inverse stage on register bit MB: pairs (m, m | h), twiddle w[q], q = m & (h - 1)
template <int MB>
__device__ __forceinline__ void inv_stage(double (&re)[R], double (&im)[R], const cplx* w) {
    constexpr int h = 1 << MB;
#pragma unroll
    for (int m = 0; m < R; m++) {
        if (m & h) continue;
        const int q = m & (h - 1);
        bfly(re[m], im[m], re[m | h], im[m | h], w[q].x, w[q].y);
    }
}
__device__ __forceinline__ void inv_pass(double (&re)[R], double (&im)[R], const cplx* w) {
    inv_stage<0>(re, im, w);
    inv_stage<1>(re, im, w + 1);
    inv_stage<2>(re, im, w + 3);
}

// first inverse pass (DIT stages of half-size 1, 2, 4 on the register index): twiddles 1; 1, -i; 1, (1 - i)/sqrt 2, -i, (-1 - i)/sqrt 2
__device__ __forceinline__ void inv_pass1(double (&re)[R], double (&im)[R]) {
    auto plain = [&](int a, int b) {       // w = 1
        const double ar = re[a], ai = im[a], br = re[b], bi = im[b];
        re[a] = ar + br; im[a] = ai + bi; re[b] = ar - br; im[b] = ai - bi;
    };
    auto minus_i = [&](int a, int b) {     // w = -i: w b = (bi, -br)
        const double ar = re[a], ai = im[a], br = re[b], bi = im[b];
        re[a] = ar + bi; im[a] = ai - br; re[b] = ar - bi; im[b] = ai + br;
    };
#pragma unroll
    for (int m = 0; m < R; m += 2) plain(m, m + 1);
#pragma unroll
    for (int m = 0; m < R; m += 4) { plain(m, m + 2); minus_i(m + 1, m + 3); }
    plain(0, 4);
    {   // w = (1 - i)/sqrt 2: w b = c ((br + bi) + i (bi - br))
        const double s = re[5] + im[5], d = im[5] - re[5], ar = re[1], ai = im[1];
        re[1] = __builtin_fma(SQRT_HALF, s, ar); im[1] = __builtin_fma(SQRT_HALF, d, ai);
        re[5] = __builtin_fma(-SQRT_HALF, s, ar); im[5] = __builtin_fma(-SQRT_HALF, d, ai);
    }
    minus_i(2, 6);
    {   // w = (-1 - i)/sqrt 2: w b = c ((bi - br) - i (br + bi))
        const double d = im[7] - re[7], s = re[7] + im[7], ar = re[3], ai = im[3];
        re[3] = __builtin_fma(SQRT_HALF, d, ar); im[3] = __builtin_fma(-SQRT_HALF, s, ai);
        re[7] = __builtin_fma(-SQRT_HALF, d, ar); im[7] = __builtin_fma(SQRT_HALF, s, ai);
    }
}

// untwist (x * (c, s), 1/n folded in) fused with the rounding: the 64-bit patterns of x' + MAGIC
__device__ __forceinline__ void untwist_round(double (&re)[R], double (&im)[R], const cplx* ut) {
#pragma unroll
    for (int m = 0; m < R; m++) {
        const double xr = re[m], xi = im[m];
        re[m] = __builtin_fma(xr, ut[m].x, __builtin_fma(-xi, ut[m].y, MAGIC));
        im[m] = __builtin_fma(xi, ut[m].x, __builtin_fma(xr, ut[m].y, MAGIC));
    }
}
// floor(x + 1/2) mod 2^32 of a value that went through untwist_round
__device__ __forceinline__ uint32_t rounded_u32(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return __builtin_amdgcn_alignbit((uint32_t)(b >> 32), (uint32_t)b, 16);
}
// (floor(x + 1/2) mod 2^16) << 16: the weight of the high key half, in place
__device__ __forceinline__ uint32_t rounded_hi16(double v) {
    return (uint32_t)__double_as_longlong(v) & 0xFFFF0000u;
}

using rtfhe::flag_arrive;      // (rtfhe_device.hpp: flags in LDS addressed by scalar registers, sleepy waits)
using rtfhe::flag_wait;

// s (+)= b * x, 4 FMA per point; FIRST: s = b * x
template <bool FIRST>
__device__ __forceinline__ void mac(double (&sre)[R], double (&sim)[R], const cplx (&b)[R], const double (&xr)[R], const double (&xi)[R]) {
#pragma unroll
    for (int m = 0; m < R; m++) {
        const double r0 = FIRST ? b[m].x * xr[m] : __builtin_fma(b[m].x, xr[m], sre[m]);
        const double i0 = FIRST ? b[m].x * xi[m] : __builtin_fma(b[m].x, xi[m], sim[m]);
        sre[m] = __builtin_fma(-b[m].y, xi[m], r0);
        sim[m] = __builtin_fma(b[m].y, xr[m], i0);
    }
}

// the same over one half of a lane's points (h = 0: points 0..3, 1: points 4..7): the key arrives in half rows (rtfhe_kernels_xfft.hpp)
__device__ __forceinline__ void mac_half(double (&sre)[R], double (&sim)[R], const cplx (&b)[R / 2], const double (&xr)[R], const double (&xi)[R], int h, bool first) {
#pragma unroll
    for (int k = 0; k < R / 2; k++) {
        const int m = h * (R / 2) + k;
        const double r0 = first ? b[k].x * xr[m] : __builtin_fma(b[k].x, xr[m], sre[m]);
        const double i0 = first ? b[k].x * xi[m] : __builtin_fma(b[k].x, xi[m], sim[m]);
        sre[m] = __builtin_fma(-b[k].y, xi[m], r0);
        sim[m] = __builtin_fma(b[k].y, xr[m], i0);
    }
}

// NR forward transforms side by side in one wave.  in: layout L1 (re[j][m], im[j][m] = folded point lane + 64 m of row j); out: layout L3
// (position (lane << 3) | m, bit-reversed frequency order -- the order the key is stored in).  tw: LDS table (XTw); the seven pass-1 twiddles
// come as w1 (registers / scalars of the caller).  A row's exchange reads are issued between the stages of the NEXT row's pass (as
// fft_forward_multi_a's interleaved form, rtfhe_device.hpp).
struct NoPoint { __device__ __forceinline__ void operator()(int) const {} };
// (forward_multi_t: the pass-2 / pass-3 tables given explicitly -- [7][8] by lane >> 3 and [7][64] by lane; the N = 2048 kernel runs this
// 512-point transform with one of two table sets, rtfhe_kernels_xfft2.hpp)
template <int NR, typename HOOK = NoPoint>
__device__ __forceinline__ void forward_multi_t(double (&re)[NR][R], double (&im)[NR][R], const cplx* __restrict__ f2, const cplx* __restrict__ f3,
                                                const cplx (&w1)[7], double* __restrict__ xbuf, double* __restrict__ xim, int lane, HOOK point = HOOK()) {
    typedef XAffine<10, 1, 2> X1;
    typedef XAffine<10, 2, 3> X2;
    auto pass = [&](int j, const cplx* w, auto&& between0, auto&& between1) {
        fwd_stage<2>(re[j], im[j], w);
        __builtin_amdgcn_sched_barrier(0); between0(); __builtin_amdgcn_sched_barrier(0);
        fwd_stage<1>(re[j], im[j], w + 1);
        __builtin_amdgcn_sched_barrier(0); between1(); __builtin_amdgcn_sched_barrier(0);
        fwd_stage<0>(re[j], im[j], w + 3);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto nothing = [] {};
    if constexpr (NR == 1) {       // a single row has no neighbour to hide its exchanges under
        fwd_pass(re[0], im[0], w1);
        exchange<10, 1, 2, 1>(re[0], im[0], xbuf, lane, xim);
        Tw<7> w2;
        w2.load(f2 + (lane >> 3), 8);
        fwd_pass(re[0], im[0], w2.w);
        exchange<10, 2, 3, 1>(re[0], im[0], xbuf, lane, xim);
        Tw<7> w3;
        w3.load(f3 + lane, 64);
        fwd_pass(re[0], im[0], w3.w);
        return;
    }
#pragma unroll
    for (int j = 0; j < NR; j++) {
        if (j == 0) pass(0, w1, nothing, nothing);
        else pass(j, w1, [&] { X1::template read_half<0>(re[j - 1], im[j - 1], xbuf, xim, lane); }, [&] { X1::template read_half<1>(re[j - 1], im[j - 1], xbuf, xim, lane); wave_lds_sync(); });
        X1::write(re[j], im[j], xbuf, xim, lane);
        wave_lds_sync();
    }
    point(1);
    Tw<7> w2;
    w2.load(f2 + (lane >> 3), 8);
#pragma unroll
    for (int j = 0; j < NR; j++) {
        if (j == 0) pass(0, w2.w, [&] { X1::template read_half<0>(re[NR - 1], im[NR - 1], xbuf, xim, lane); }, [&] { X1::template read_half<1>(re[NR - 1], im[NR - 1], xbuf, xim, lane); wave_lds_sync(); });
        else pass(j, w2.w, [&] { X2::template read_half<0>(re[j - 1], im[j - 1], xbuf, xim, lane); }, [&] { X2::template read_half<1>(re[j - 1], im[j - 1], xbuf, xim, lane); wave_lds_sync(); });
        X2::write(re[j], im[j], xbuf, xim, lane);
        wave_lds_sync();
    }
    point(2);
    Tw<7> w3;
    w3.load(f3 + lane, 64);
#pragma unroll
    for (int j = 0; j < NR; j++) {
        if (j == 0) pass(0, w3.w, [&] { X2::template read_half<0>(re[NR - 1], im[NR - 1], xbuf, xim, lane); }, [&] { X2::template read_half<1>(re[NR - 1], im[NR - 1], xbuf, xim, lane); wave_lds_sync(); });
        else pass(j, w3.w, nothing, nothing);
    }
}
template <int NR, typename HOOK = NoPoint>
__device__ __forceinline__ void forward_multi(double (&re)[NR][R], double (&im)[NR][R], const cplx* __restrict__ tw, const cplx (&w1)[7],
                                              double* __restrict__ xbuf, double* __restrict__ xim, int lane, HOOK point = HOOK()) {
    forward_multi_t<NR>(re, im, tw + XTw::F2, tw + XTw::F3, w1, xbuf, xim, lane, point);
}

// NR inverse transforms side by side.  in: layout L3 (the multiply-accumulate's sums); out: layout L1, untwisted, scaled and carrying MAGIC
// (re[j][m] <-> coefficient lane + 64 m, im[j][m] <-> coefficient lane + 64 m + N/2; read with rounded_u32 / rounded_hi16).
// (inverse_core: the nine stages alone, out: layout L1, NOT untwisted; pass-2 / pass-3 tables given explicitly -- [7][8] by lane & 7, [7][64] by lane)
template <int NR, typename HOOK = NoPoint>
__device__ __forceinline__ void inverse_core(double (&re)[NR][R], double (&im)[NR][R], const cplx* __restrict__ i2, const cplx* __restrict__ i3,
                                             double* __restrict__ xbuf, double* __restrict__ xim, int lane, HOOK point = HOOK()) {
    typedef XAffine<10, 3, 2> X1;
    typedef XAffine<10, 2, 1> X2;
#pragma unroll
    for (int j = 0; j < NR; j++) {
        inv_pass1(re[j], im[j]);
        if (j > 0) { X1::template read_half<0>(re[j - 1], im[j - 1], xbuf, xim, lane); X1::template read_half<1>(re[j - 1], im[j - 1], xbuf, xim, lane); wave_lds_sync(); }
        X1::write(re[j], im[j], xbuf, xim, lane);
        wave_lds_sync();
    }
    point(1);
    Tw<7> w2;
    w2.load(i2 + (lane & 7), 8);
#pragma unroll
    for (int j = 0; j < NR; j++) {
        if (j == 0) { X1::template read_half<0>(re[NR - 1], im[NR - 1], xbuf, xim, lane); X1::template read_half<1>(re[NR - 1], im[NR - 1], xbuf, xim, lane); wave_lds_sync(); }
        inv_pass(re[j], im[j], w2.w);
        if (j > 0) { X2::template read_half<0>(re[j - 1], im[j - 1], xbuf, xim, lane); X2::template read_half<1>(re[j - 1], im[j - 1], xbuf, xim, lane); wave_lds_sync(); }
        X2::write(re[j], im[j], xbuf, xim, lane);
        wave_lds_sync();
    }
    point(2);
    Tw<7> w3;
    w3.load(i3 + lane, 64);
#pragma unroll
    for (int j = 0; j < NR; j++) {
        if (j == 0) { X2::template read_half<0>(re[NR - 1], im[NR - 1], xbuf, xim, lane); X2::template read_half<1>(re[NR - 1], im[NR - 1], xbuf, xim, lane); wave_lds_sync(); }
        inv_pass(re[j], im[j], w3.w);
    }
}
template <int NR, typename HOOK = NoPoint>
__device__ __forceinline__ void inverse_multi(double (&re)[NR][R], double (&im)[NR][R], const cplx* __restrict__ tw,
                                              double* __restrict__ xbuf, double* __restrict__ xim, int lane, HOOK point = HOOK()) {
    inverse_core<NR>(re, im, tw + XTw::I2, tw + XTw::I3, xbuf, xim, lane, point);
    Tw<8> ut;
    ut.load(tw + XTw::UT + lane, 64);
#pragma unroll
    for (int j = 0; j < NR; j++) untwist_round(re[j], im[j], ut.w);
}

}  // namespace xfft
}  // namespace rtfhe
