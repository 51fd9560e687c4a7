// rtfhe_device.hpp -- gfx950 device building blocks for the HomNAND hot path.
//
// One 64-lane wavefront owns one polynomial transform (and, in the bootstrap kernel, one whole
// gate): the N/2-point complex FP64 transform is held R = N/128 points per lane and runs as three
// in-register radix-R passes separated by two wave-private LDS exchanges.  No workgroup barrier is
// ever needed inside a transform.
//
// Parity: the arithmetic DAG (which two values meet in which butterfly, which twiddle multiplies
// which difference, every product and sum rounded on its own -- the file is compiled with
// -ffp-contract=off) is exactly the one of the reference's native FFT:
//   forward  = `ifft`  spqlios-ifft-avx.s:64-272  (readable spec ifft_model, spqlios-fft-impl.cpp:469-641)
//   inverse  = `fft`   spqlios-fft-avx.s:79-280   (readable spec fft_model,  spqlios-fft-impl.cpp:204-397)
// Only the schedule differs (which lane/register holds which point, and when).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rtfhe {

typedef double2 cplx;  // .x = re / cos, .y = im / sin

// The bootstrap kernels (outputs = torus words) skip the multiplies of the one butterfly per transform whose twiddle is exactly (1, 0) and the
// "+0.0 +" of a fold's first row -- see fwd_stage_tw (TRIV0) for why no torus word can change; the host refuses a twiddle table whose entry is
// not exactly (1, +-0) (unit_twiddles_ok, rtfhe_twiddles.hip).
constexpr bool BOOT_TRIV = true;

__host__ __device__ constexpr int ilog2(int v) { return v <= 1 ? 0 : 1 + ilog2(v >> 1); }

// Geometry of the N/2-point transform on one wavefront.
//   layout L1: lane t,           register m <-> point  t + 64 m                    (m = top LR bits)
//   layout L2: lane (B,r),       register m <-> point (B << 6) | (m << LOW) | r    (m = middle LR bits)
//   layout L3: lane v,           register m <-> point (v << LR) | m                (m = low LR bits)
template <int LOGN>
struct Geo {
    static constexpr int N = 1 << LOGN;
    static constexpr int P = N / 2;
    static constexpr int LOGP = LOGN - 1;
    static constexpr int R = P / 64;
    static constexpr int LR = ilog2(R);
    static constexpr int LOW = LOGP - 2 * LR;
    static constexpr int NLOW = 1 << LOW;
    static_assert(LOW >= 2 && LOW <= LR, "supported: N = 1024, 2048");
    // twiddle table (cplx units), one per direction
    static constexpr int TW_TWIST = 0;                         // [R][64]
    static constexpr int TW_P1 = TW_TWIST + R * 64;            // [R-1][64]
    static constexpr int TW_P2 = TW_P1 + (R - 1) * 64;         // [R-1][NLOW]
    static constexpr int TW_P3 = TW_P2 + (R - 1) * NLOW;       // [NLOW-4]  (stages with halfnn >= 4 left for pass 3)
    static constexpr int TW_DIR = TW_P3 + (NLOW - 4);          // cplx per direction
    static constexpr int TW_TOTAL = 2 * TW_DIR;                // forward then inverse
    // LDS exchange buffer, slots of one double (real and imaginary halves take turns)
    static constexpr int XSLOTS = P + 64;
    __host__ __device__ static constexpr int f1(int pos) { return pos + NLOW * (pos >> 6); }   // L1 <-> L2
    __host__ __device__ static constexpr int f2(int pos) { return pos + (pos >> LR); }         // L2 <-> L3
    __host__ __device__ static constexpr int pos1(int lane, int m) { return lane + 64 * m; }
    __host__ __device__ static constexpr int pos2(int lane, int m) {
        return ((lane >> LOW) << 6) | (m << LOW) | (lane & (NLOW - 1));
    }
    __host__ __device__ static constexpr int pos3(int lane, int m) { return (lane << LR) | m; }
};

// A wave-private LDS hand-off: DS instructions of one wave execute in order, so no s_barrier is
// needed; this only stops the compiler from moving LDS accesses across the point.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One double to / from an exchange buffer as its OWN ds_write_b64 / ds_read_b64.  Plain accesses with constant offsets from
// one base are paired by the compiler into ds_write2_b64 / ds_read2_b64, and on gfx950 a ds_read2_b64 occupies the LDS for
// 8 cycles where two ds_read_b64 take 2 + 2 (ds_write2_b64: 13 against 6 + 6) -- MI355X_MICROARCH.md, LDS table; measured
// here: one 16-double exchange at two waves per SIMD costs the CU 81 cycles unpaired against 101 paired
// (scripts/ubench/lds_forms.hip, profiles/r03/lds_forms.log), the headline kernel 7.35 -> 7.12 ms per 1024 gates.  A relaxed
// wavefront-scope atomic access lowers to exactly the plain instruction and is never paired.
__device__ __forceinline__ void lds_st(double* p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}
__device__ __forceinline__ double lds_ld(const double* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}
// ... and one complex value as ONE 16-byte access (ds_write_b128 / ds_read_b128: 13 / 4 LDS cycles against 6 + 6 / 2 + 2 for its two halves, i.e.
// the same LDS time, in half the instructions -- scripts/ubench/issue_mix.hip: at two waves per SIMD an instruction of any kind costs the SIMD's
// issue about as much as an FP64 one).  p: 16-byte aligned.
__device__ __forceinline__ void lds_st128(cplx* p, double re, double im) { *p = make_double2(re, im); }
__device__ __forceinline__ cplx lds_ld128(const cplx* p) { return *p; }

// ---------------------------------------------------------------------------------------------
// butterflies (one register-index bit MB at a time; h = 1 << MB; q = m & (h-1) selects the twiddle)
// ---------------------------------------------------------------------------------------------

// All twiddles of one in-register pass, loaded from LDS ahead of use so that their latency overlaps the
// previous pass / exchange.  Entry R - 2h + q belongs to the stage with h = 1 << MB, q = m & (h - 1).
template <int CNT>
struct Tw {
    cplx w[CNT > 0 ? CNT : 1];
    __device__ __forceinline__ void load(const cplx* __restrict__ tw, int stride) {
#pragma unroll
        for (int e = 0; e < CNT; e++) w[e] = tw[e * stride];
    }
};

// DIF, twiddled: x0' = x0 + x1 ; x1' = (x0 - x1) * w      (spqlios-fft-impl.cpp:546-569)
// TRIV0: the caller vouches that w[0] is exactly (1.0, +0.0) -- true of every table the reference builders produce for a
// stage's first entry, cos(0) / sin(0) (a build that turns it on must refuse any other table in rtfhe_set_twiddles) -- and the
// butterfly with q = 0 then skips its six multiply / add instructions: d * 1.0 == d and d - e * 0.0 == d for every finite d, e
// EXCEPT in the sign of a zero result.  The sign of a zero never reaches a torus word: zero + x == x, zero * w is a zero, there
// is no division or comparison on the path and Torus32(int64_t(+-0.0)) == 0.  Only kernels whose outputs are torus words use it;
// the stage-level transform kernels, whose outputs ARE spectra, keep TRIV0 = false and stay byte-identical to the reference.
template <int R, int MB, bool TRIV0 = false>
__device__ __forceinline__ void fwd_stage_tw(double (&re)[R], double (&im)[R], const cplx* w) {
    constexpr int h = 1 << MB;
#pragma unroll
    for (int m = 0; m < R; m++) {
        if (m & h) continue;
        const int m1 = m | h, q = m & (h - 1);
        const double sr = re[m] + re[m1], si = im[m] + im[m1];
        const double dr = re[m] - re[m1], di = im[m] - im[m1];
        re[m] = sr; im[m] = si;
        if (TRIV0 && q == 0) { re[m1] = dr; im[m1] = di; continue; }
        double a = dr * w[q].x, b = di * w[q].y;
        re[m1] = a - b;
        a = dr * w[q].y; b = di * w[q].x;
        im[m1] = a + b;
    }
}

// DIT, twiddled: t = x1 * w ; x0' = x0 + t ; x1' = x0 - t   (spqlios-fft-impl.cpp:346-359)
// CONJ (UNUSED by every shipped kernel, kept for the experiment recorded in DESIGN.md 5.4): w holds the FORWARD table's (c, s) and the
// stage multiplies by its conjugate (c, -s).  x * (-s) == -(x * s) and a - (-b) == a + b exactly, so t0 + t3 / t2 - t1 are the bits
// t0 - t3' / t1' + t2 give with (c, s') = (c, -s).  NOTE: the reference's inverse table is NOT the exact conjugate of its forward table (at
// the quarter turn of every stage cos comes out as -6.1e-17 forward and +6.1e-17 inverse), and no host code checks for it: a kernel
// that wanted CONJ = true would first have to patch those entries, as the N = 2048 experiment did.
template <int R, int MB, bool CONJ = false, bool TRIV0 = false>
__device__ __forceinline__ void inv_stage_tw(double (&re)[R], double (&im)[R], const cplx* w) {
    constexpr int h = 1 << MB;
#pragma unroll
    for (int m = 0; m < R; m++) {
        if (m & h) continue;
        const int m1 = m | h, q = m & (h - 1);
        if (TRIV0 && q == 0) {     // t = x1 * (1, 0) = x1 up to the sign of a zero (see fwd_stage_tw)
            const double ar = re[m], ai = im[m], tr = re[m1], ti = im[m1];
            re[m] = ar + tr; im[m] = ai + ti;
            re[m1] = ar - tr; im[m1] = ai - ti;
            continue;
        }
        const double t0 = re[m1] * w[q].x, t1 = re[m1] * w[q].y, t2 = im[m1] * w[q].x, t3 = im[m1] * w[q].y;
        const double tr = CONJ ? t0 + t3 : t0 - t3, ti = CONJ ? t2 - t1 : t1 + t2;
        const double ar = re[m], ai = im[m];
        re[m] = ar + tr; im[m] = ai + ti;
        re[m1] = ar - tr; im[m1] = ai - ti;
    }
}

// halfnn = 1 (both directions): (x0 + x1, x0 + (-x1))   (spqlios-fft-impl.cpp:248-269, 612-633)
template <int R>
__device__ __forceinline__ void stage_size2(double (&re)[R], double (&im)[R]) {
#pragma unroll
    for (int m = 0; m < R; m += 2) {
        const double r0 = re[m], r1 = re[m + 1], j0 = im[m], j1 = im[m + 1];
        re[m] = r0 + r1; re[m + 1] = r0 + (-r1);
        im[m] = j0 + j1; im[m + 1] = j0 + (-j1);
    }
}

// forward halfnn = 2: x0+x2, x1+x3, x0-x2, i*(x1-x3)   (spqlios-fft-impl.cpp:581-602)
template <int R>
__device__ __forceinline__ void fwd_stage_size4(double (&re)[R], double (&im)[R]) {
#pragma unroll
    for (int m = 0; m < R; m += 4) {
        const double r0 = re[m], r1 = re[m + 1], r2 = re[m + 2], r3 = re[m + 3];
        const double j0 = im[m], j1 = im[m + 1], j2 = im[m + 2], j3 = im[m + 3];
        re[m] = r0 + r2; re[m + 1] = r1 + r3; re[m + 2] = r0 + (-r2); re[m + 3] = (-j1) + j3;
        im[m] = j0 + j2; im[m + 1] = j1 + j3; im[m + 2] = j0 + (-j2); im[m + 3] = r1 + (-r3);
    }
}

// inverse halfnn = 2: x0+x2, x1-i*x3, x0-x2, x1+i*x3   (spqlios-fft-impl.cpp:289-310)
template <int R>
__device__ __forceinline__ void inv_stage_size4(double (&re)[R], double (&im)[R]) {
#pragma unroll
    for (int m = 0; m < R; m += 4) {
        const double r0 = re[m], r1 = re[m + 1], r2 = re[m + 2], r3 = re[m + 3];
        const double j0 = im[m], j1 = im[m + 1], j2 = im[m + 2], j3 = im[m + 3];
        re[m] = r0 + r2; re[m + 1] = r1 + j3;    re[m + 2] = r0 + (-r2); re[m + 3] = r1 + (-j3);
        im[m] = j0 + j2; im[m + 1] = j1 + (-r3); im[m + 2] = j0 + (-j2); im[m + 3] = j1 + r3;
    }
}

// (re,im) * (c,s):  re*c - im*s , im*c + re*s   (spqlios-fft-impl.cpp:512-517, 390-395)
template <int R>
__device__ __forceinline__ void twist_mul(double (&re)[R], double (&im)[R], const cplx* w) {
#pragma unroll
    for (int m = 0; m < R; m++) {
        const double rc = re[m] * w[m].x, ic = im[m] * w[m].x, rs = re[m] * w[m].y, is = im[m] * w[m].y;
        re[m] = rc - is;
        im[m] = ic + rs;
    }
}

template <int R, int MBTOP>
struct P12 {   // passes 1 and 2: all LR register bits, twiddled; w = the pass's R-1 twiddles
    __device__ __forceinline__ static void fwd(double (&re)[R], double (&im)[R], const cplx* w) {
        if constexpr (MBTOP >= 0) {
            constexpr int h = 1 << MBTOP;
            fwd_stage_tw<R, MBTOP>(re, im, w + (R - 2 * h));
            P12<R, MBTOP - 1>::fwd(re, im, w);
        }
    }
    template <bool CONJ = false>
    __device__ __forceinline__ static void inv(double (&re)[R], double (&im)[R], const cplx* w) {
        if constexpr (MBTOP >= 0) {
            constexpr int h = 1 << MBTOP;
            P12<R, MBTOP - 1>::template inv<CONJ>(re, im, w);
            inv_stage_tw<R, MBTOP, CONJ>(re, im, w + (R - 2 * h));
        }
    }
};

template <int R, int NLOW, int MBTOP, bool TRIV = false>
struct P3 {    // pass 3: register bits LOW-1 .. 0; bits >= 2 twiddled (wave-uniform twiddles), bits 1, 0 special
    __device__ __forceinline__ static void fwd(double (&re)[R], double (&im)[R], const cplx* w) {
        if constexpr (MBTOP >= 2) {
            constexpr int h = 1 << MBTOP;
            fwd_stage_tw<R, MBTOP, TRIV>(re, im, w + (NLOW - 2 * h));
            P3<R, NLOW, MBTOP - 1, TRIV>::fwd(re, im, w);
        } else {
            fwd_stage_size4<R>(re, im);
            stage_size2<R>(re, im);
        }
    }
    template <bool CONJ = false>
    __device__ __forceinline__ static void inv(double (&re)[R], double (&im)[R], const cplx* w) {
        if constexpr (MBTOP >= 2) {
            constexpr int h = 1 << MBTOP;
            P3<R, NLOW, MBTOP - 1, TRIV>::template inv<CONJ>(re, im, w);
            inv_stage_tw<R, MBTOP, CONJ, TRIV>(re, im, w + (NLOW - 2 * h));
        } else {
            stage_size2<R>(re, im);
            inv_stage_size4<R>(re, im);
        }
    }
};

// Wave-private exchange of the 2R doubles a lane holds (R real parts, then R imaginary parts) through a
// buffer of Geo::XSLOTS doubles: slot maps FROM (written layout) and TO (read layout).  Real and imaginary
// halves go through the same buffer one after the other (half the LDS footprint of a cplx buffer at the same
// LDS cycle count: ds_write_b64 / ds_read_b64 move 8 B per lane per 6 / 2 cycles vs 16 B per 13 / 4).
// DUAL = true gives the real and the imaginary halves a buffer each (2 x XSLOTS doubles): both are written first and
// then read back pair by pair in the order the next pass consumes them, so that its first butterflies start while
// the later reads are still in flight (matters when one wave has the SIMD to itself).
// DUAL = 2 (N = 1024 geometry only): the two buffers are ONE array of XSLOTS complex values at xbuf (16-byte aligned; ximbuf unused), every slot one
// 16-byte access.
template <int LOGN, int FROM, int TO, int DUAL = 0>
__device__ __forceinline__ void exchange(double (&re)[Geo<LOGN>::R], double (&im)[Geo<LOGN>::R],
                                         double* __restrict__ xbuf, int lane, double* __restrict__ ximbuf = nullptr) {
    typedef Geo<LOGN> G;
    constexpr int R = G::R;
    auto slot = [&](int layout, int m) {
        // the pad map is the one of the exchange (f1 for L1<->L2, f2 for L2<->L3), the position the one of the layout
        const int pos = layout == 1 ? G::pos1(lane, m) : layout == 2 ? G::pos2(lane, m) : G::pos3(lane, m);
        return (FROM + TO == 3) ? G::f1(pos) : G::f2(pos);
    };
    if constexpr (DUAL && G::LR == G::LOW) {
        // N = 1024 geometry: every slot map is affine in the register index m, slot(layout, m) = base(lane) + stride * m
        //   L1 through f1:  lane                                  + (64 + NLOW) m
        //   L2 through f1:  (64 + NLOW)(lane >> LOW) + (lane & (NLOW-1)) + NLOW m
        //   L2 through f2:  the same base                         + (NLOW + 1) m
        //   L3 through f2:  (R + 1) lane                          + m
        // so an exchange needs two address registers and immediates (the generic form below costs ~6 integer ops per slot)
        double* xim = ximbuf ? ximbuf : xbuf + G::XSLOTS;
        constexpr bool X12 = (FROM + TO == 3);
        auto base = [&](int layout) {
            return layout == 1 ? lane : layout == 2 ? (G::NLOW + 64) * (lane >> G::LOW) + (lane & (G::NLOW - 1)) : (R + 1) * lane;
        };
        auto stride = [](int layout) constexpr { return layout == 1 ? 64 + G::NLOW : layout == 2 ? (X12 ? G::NLOW : G::NLOW + 1) : 1; };
        static_assert(G::f1(G::pos1(5, 3)) == 5 + (64 + G::NLOW) * 3 && G::f2(G::pos3(5, 3)) == (R + 1) * 5 + 3, "affine slot maps");
        const int bw = base(FROM), br = base(TO);
        constexpr int sw = stride(FROM), sr = stride(TO);
        if constexpr (DUAL == 2) {
            cplx* xc = reinterpret_cast<cplx*>(xbuf);
#pragma unroll
            for (int m = 0; m < R; m++) lds_st128(&xc[bw + sw * m], re[m], im[m]);
            wave_lds_sync();
#pragma unroll
            for (int m = 0; m < R / 2; m++) {
                const cplx a = lds_ld128(&xc[br + sr * m]), b = lds_ld128(&xc[br + sr * (m + R / 2)]);
                re[m] = a.x; im[m] = a.y; re[m + R / 2] = b.x; im[m + R / 2] = b.y;
            }
            wave_lds_sync();
            return;
        }
#pragma unroll
        for (int m = 0; m < R; m++) lds_st(&xbuf[bw + sw * m], re[m]);
#pragma unroll
        for (int m = 0; m < R; m++) lds_st(&xim[bw + sw * m], im[m]);
        wave_lds_sync();
#pragma unroll
        for (int m = 0; m < R / 2; m++) {
            re[m] = lds_ld(&xbuf[br + sr * m]); re[m + R / 2] = lds_ld(&xbuf[br + sr * (m + R / 2)]);
            im[m] = lds_ld(&xim[br + sr * m]);  im[m + R / 2] = lds_ld(&xim[br + sr * (m + R / 2)]);
        }
        wave_lds_sync();
    } else if constexpr (DUAL) {
        double* xim = ximbuf ? ximbuf : xbuf + G::XSLOTS;   // second buffer: caller's, or right behind the first
#pragma unroll
        for (int m = 0; m < R; m++) lds_st(&xbuf[slot(FROM, m)], re[m]);
#pragma unroll
        for (int m = 0; m < R; m++) lds_st(&xim[slot(FROM, m)], im[m]);
        wave_lds_sync();
        // first stage of the next pass pairs m with m + R/2
#pragma unroll
        for (int m = 0; m < R / 2; m++) {
            re[m] = lds_ld(&xbuf[slot(TO, m)]); re[m + R / 2] = lds_ld(&xbuf[slot(TO, m + R / 2)]);
            im[m] = lds_ld(&xim[slot(TO, m)]);  im[m + R / 2] = lds_ld(&xim[slot(TO, m + R / 2)]);
        }
        wave_lds_sync();
    } else {
#pragma unroll
        for (int m = 0; m < R; m++) lds_st(&xbuf[slot(FROM, m)], re[m]);
        wave_lds_sync();
#pragma unroll
        for (int m = 0; m < R; m++) re[m] = lds_ld(&xbuf[slot(TO, m)]);
        wave_lds_sync();
#pragma unroll
        for (int m = 0; m < R; m++) lds_st(&xbuf[slot(FROM, m)], im[m]);
        wave_lds_sync();
#pragma unroll
        for (int m = 0; m < R; m++) im[m] = lds_ld(&xbuf[slot(TO, m)]);
        wave_lds_sync();
    }
}

// Forward transform in two parts so that a caller can issue global loads between them.
// part A: twist, pass 1, exchange, pass 2.  in: layout L1 (re[m], im[m] = point lane + 64 m), not yet twisted.
// part B: exchange, pass 3.                  out: layout L3 (point (lane << LR) | m) = the reference's FrrSeries order.
// tw: LDS, forward table.  xbuf: LDS, wave-private, Geo::XSLOTS doubles.
template <int LOGN, int DUAL = 0>
__device__ __forceinline__ void fft_forward_a(double (&re)[Geo<LOGN>::R], double (&im)[Geo<LOGN>::R],
                                              const cplx* __restrict__ tw, double* __restrict__ xbuf, int lane,
                                              double* __restrict__ xim = nullptr) {
    typedef Geo<LOGN> G;
    constexpr int R = G::R;
    Tw<R> wt; Tw<R - 1> w1, w2;
    wt.load(tw + G::TW_TWIST + lane, 64);
    w1.load(tw + G::TW_P1 + lane, 64);
    twist_mul<R>(re, im, wt.w);
    P12<R, G::LR - 1>::fwd(re, im, w1.w);
    w2.load(tw + G::TW_P2 + (lane & (G::NLOW - 1)), G::NLOW);     // in flight during the exchange
    exchange<LOGN, 1, 2, DUAL>(re, im, xbuf, lane, xim);
    P12<R, G::LR - 1>::fwd(re, im, w2.w);
}
template <int LOGN, int DUAL = 0, bool TRIV = false>
__device__ __forceinline__ void fft_forward_b(double (&re)[Geo<LOGN>::R], double (&im)[Geo<LOGN>::R],
                                              const cplx* __restrict__ tw, double* __restrict__ xbuf, int lane,
                                              double* __restrict__ xim = nullptr) {
    typedef Geo<LOGN> G;
    Tw<G::NLOW - 4> w3;
    w3.load(tw + G::TW_P3, 1);
    exchange<LOGN, 2, 3, DUAL>(re, im, xbuf, lane, xim);
    P3<G::R, G::NLOW, G::LOW - 1, TRIV>::fwd(re, im, w3.w);
}
template <int LOGN, int DUAL = 0, bool TRIV = false>
__device__ __forceinline__ void fft_forward(double (&re)[Geo<LOGN>::R], double (&im)[Geo<LOGN>::R],
                                            const cplx* __restrict__ tw, double* __restrict__ xbuf, int lane,
                                            double* __restrict__ xim = nullptr) {
    fft_forward_a<LOGN, DUAL>(re, im, tw, xbuf, lane, xim);
    fft_forward_b<LOGN, DUAL, TRIV>(re, im, tw, xbuf, lane, xim);
}

// NR forward transforms side by side in one wave (the digit rows of one polynomial): every pass loads its twiddles ONCE for
// all rows (a third of the LDS twiddle reads), and row j's exchange is in flight while rows j+1.. compute -- DS instructions
// of a wave execute in order, so the rows may share the exchange buffers back to back without waiting for each other's reads.
// Same butterflies, same operands, same order per row as fft_forward.
// part A: twist, pass 1, first exchange, pass 2, second exchange (issued).  part B: pass 3.  in: layout L1, out: layout L3.
// TWIST = false: the rows come already twisted (the two-waves-per-transform kernel twists before its first, cross-wave stage).
// The two halves of the N = 1024 geometry's affine exchange as separate calls (see exchange<>: slot = base(lane) + stride * m), so that a caller
// can place the 16 writes and the two groups of 8 reads between blocks of arithmetic instead of issuing 32 DS instructions in one burst (a wave's
// LDS queue holds 16: a burst stalls the wave at issue, and with it its FP64 stream).
template <int LOGN, int FROM, int TO, bool B128 = false>
struct XAffine {
    typedef Geo<LOGN> G;
    static_assert(G::LR == G::LOW, "N = 1024 geometry");
    static constexpr int R = G::R;
    static constexpr bool X12 = (FROM + TO == 3);
    __device__ __forceinline__ static int base(int layout, int lane) {
        return layout == 1 ? lane : layout == 2 ? (G::NLOW + 64) * (lane >> G::LOW) + (lane & (G::NLOW - 1)) : (R + 1) * lane;
    }
    __host__ __device__ static constexpr int stride(int layout) { return layout == 1 ? 64 + G::NLOW : layout == 2 ? (X12 ? G::NLOW : G::NLOW + 1) : 1; }
    __device__ __forceinline__ static void write(const double (&re)[R], const double (&im)[R], double* __restrict__ xbuf, double* __restrict__ xim, int lane) {
        const int bw = base(FROM, lane);
        constexpr int sw = stride(FROM);
        if constexpr (B128) {      // one array of XSLOTS complex values at xbuf (xim unused)
            cplx* xc = reinterpret_cast<cplx*>(xbuf);
#pragma unroll
            for (int m = 0; m < R; m++) lds_st128(&xc[bw + sw * m], re[m], im[m]);
            return;
        }
#pragma unroll
        for (int m = 0; m < R; m++) lds_st(&xbuf[bw + sw * m], re[m]);
#pragma unroll
        for (int m = 0; m < R; m++) lds_st(&xim[bw + sw * m], im[m]);
    }
    // HALF 0: the operands of the next pass's first two butterflies (m = 0, 1 with m + R/2), HALF 1: of the other two
    template <int HALF>
    __device__ __forceinline__ static void read_half(double (&re)[R], double (&im)[R], const double* __restrict__ xbuf, const double* __restrict__ xim, int lane) {
        const int br = base(TO, lane);
        constexpr int sr = stride(TO);
        if constexpr (B128) {
            const cplx* xc = reinterpret_cast<const cplx*>(xbuf);
#pragma unroll
            for (int m = HALF * (R / 4); m < (HALF + 1) * (R / 4); m++) {
                const cplx a = lds_ld128(&xc[br + sr * m]), b = lds_ld128(&xc[br + sr * (m + R / 2)]);
                re[m] = a.x; im[m] = a.y; re[m + R / 2] = b.x; im[m + R / 2] = b.y;
            }
            return;
        }
#pragma unroll
        for (int m = HALF * (R / 4); m < (HALF + 1) * (R / 4); m++) {
            re[m] = lds_ld(&xbuf[br + sr * m]); re[m + R / 2] = lds_ld(&xbuf[br + sr * (m + R / 2)]);
            im[m] = lds_ld(&xim[br + sr * m]);  im[m + R / 2] = lds_ld(&xim[br + sr * (m + R / 2)]);
        }
    }
};

struct NoHook { __device__ __forceinline__ void operator()() const {} };
template <int LOGN, int NR, bool TWIST = true, typename HOOK = NoHook, bool INTERLEAVE = false, bool B128 = false>
__device__ __forceinline__ void fft_forward_multi_a(double (&re)[NR][Geo<LOGN>::R], double (&im)[NR][Geo<LOGN>::R],
                                                    const cplx* __restrict__ tw, double* __restrict__ xbuf, double* __restrict__ xim, int lane,
                                                    HOOK after_pass1 = HOOK()) {
    typedef Geo<LOGN> G;
    constexpr int R = G::R;
#pragma unroll
    for (int half = 0; half < (TWIST ? 2 : 0); half++) {        // twist twiddles in two halves: R/2 of them live at a time
        cplx wt[R / 2];
#pragma unroll
        for (int m = 0; m < R / 2; m++) wt[m] = tw[G::TW_TWIST + lane + 64 * (half * (R / 2) + m)];
#pragma unroll
        for (int j = 0; j < NR; j++)
#pragma unroll
            for (int m = 0; m < R / 2; m++) {
                const int k = half * (R / 2) + m;
                const double rc = re[j][k] * wt[m].x, ic = im[j][k] * wt[m].x, rs = re[j][k] * wt[m].y, is = im[j][k] * wt[m].y;
                re[j][k] = rc - is;
                im[j][k] = ic + rs;
            }
    }
  if constexpr (INTERLEAVE) {
    // Interleaved form (N = 1024, R = 8): a row's 16 exchange writes follow its pass; its 16 reads are issued in two groups of 8 BETWEEN the three
    // stages of the NEXT row's pass (the last row's between the stages of the next pass's first row), each group pinned by scheduling barriers.  No
    // burst is longer than 16 DS instructions and every read has a stage of arithmetic to land under.
    // Measured: k_bootstrap_pair 6.73 -> 6.69 ms per 1024 gates; the N = 2048 kernel, whose waves also meet at barriers inside a step, 17.13 -> 17.39
    // (slower): on by template argument where it pays.
    static_assert(G::LR == 3, "three stages per pass");
    typedef XAffine<LOGN, 1, 2, B128> X1;
    typedef XAffine<LOGN, 2, 3, B128> X2;
    auto pass = [&](int j, const cplx* w, auto&& between0, auto&& between1) {
        fwd_stage_tw<R, 2>(re[j], im[j], w + (R - 8));
        __builtin_amdgcn_sched_barrier(0); between0(); __builtin_amdgcn_sched_barrier(0);
        fwd_stage_tw<R, 1>(re[j], im[j], w + (R - 4));
        __builtin_amdgcn_sched_barrier(0); between1(); __builtin_amdgcn_sched_barrier(0);
        fwd_stage_tw<R, 0>(re[j], im[j], w + (R - 2));
        __builtin_amdgcn_sched_barrier(0);
    };
    auto nothing = [] {};
    Tw<R - 1> w1;
    w1.load(tw + G::TW_P1 + lane, 64);
#pragma unroll
    for (int j = 0; j < NR; j++) {
        if (j == 0) pass(0, w1.w, nothing, nothing);
        else pass(j, w1.w, [&] { X1::template read_half<0>(re[j - 1], im[j - 1], xbuf, xim, lane); }, [&] { X1::template read_half<1>(re[j - 1], im[j - 1], xbuf, xim, lane); wave_lds_sync(); });
        X1::write(re[j], im[j], xbuf, xim, lane);
        wave_lds_sync();
    }
    after_pass1();
    Tw<R - 1> w2;
    w2.load(tw + G::TW_P2 + (lane & (G::NLOW - 1)), G::NLOW);
    // the last row of the first exchange is read under pass 2 of row 0 -- but row 0's own pass-2 inputs were read long ago
#pragma unroll
    for (int j = 0; j < NR; j++) {
        if (j == 0) pass(0, w2.w, [&] { X1::template read_half<0>(re[NR - 1], im[NR - 1], xbuf, xim, lane); }, [&] { X1::template read_half<1>(re[NR - 1], im[NR - 1], xbuf, xim, lane); wave_lds_sync(); });
        else pass(j, w2.w, [&] { X2::template read_half<0>(re[j - 1], im[j - 1], xbuf, xim, lane); }, [&] { X2::template read_half<1>(re[j - 1], im[j - 1], xbuf, xim, lane); wave_lds_sync(); });
        X2::write(re[j], im[j], xbuf, xim, lane);
        wave_lds_sync();
    }
    X2::template read_half<0>(re[NR - 1], im[NR - 1], xbuf, xim, lane);
    X2::template read_half<1>(re[NR - 1], im[NR - 1], xbuf, xim, lane);
    wave_lds_sync();
  } else {
    {
        Tw<R - 1> w1;
        w1.load(tw + G::TW_P1 + lane, 64);
#pragma unroll
        for (int j = 0; j < NR; j++) {
            P12<R, G::LR - 1>::fwd(re[j], im[j], w1.w);
            exchange<LOGN, 1, 2, B128 ? 2 : 1>(re[j], im[j], xbuf, lane, xim);
        }
    }
    after_pass1();        // a caller's scheduling point (e.g. a priority change) between the two batched passes
    Tw<R - 1> w2;
    w2.load(tw + G::TW_P2 + (lane & (G::NLOW - 1)), G::NLOW);
#pragma unroll
    for (int j = 0; j < NR; j++) {
        P12<R, G::LR - 1>::fwd(re[j], im[j], w2.w);
        exchange<LOGN, 2, 3, B128 ? 2 : 1>(re[j], im[j], xbuf, lane, xim);
    }
  }
}
template <int LOGN, int NR, bool TRIV = false>
__device__ __forceinline__ void fft_forward_multi_b(double (&re)[NR][Geo<LOGN>::R], double (&im)[NR][Geo<LOGN>::R],
                                                    const cplx* __restrict__ tw) {
    typedef Geo<LOGN> G;
    Tw<G::NLOW - 4> w3;
    w3.load(tw + G::TW_P3, 1);
#pragma unroll
    for (int j = 0; j < NR; j++) P3<G::R, G::NLOW, G::LOW - 1, TRIV>::fwd(re[j], im[j], w3.w);
}

// Inverse transform.  in: layout L3, unscaled (the 2/N factor lives in the untwist twiddles).  out: layout L1, untwisted (natural
// coefficient order: re[m] = coefficient lane + 64 m, im[m] = coefficient lane + 64 m + N/2).
// tw_small holds the pass-2/3 entries (always LDS), tw_big the pass-1 and untwist entries (LDS, or -- where the LDS
// budget is better spent on resident gates, N = 2048 -- the global table; both pointers use the per-direction offsets).
template <int LOGN, int DUAL = 0, bool TRIV = false>
__device__ __forceinline__ void fft_inverse(double (&re)[Geo<LOGN>::R], double (&im)[Geo<LOGN>::R],
                                            const cplx* __restrict__ tw_small, const cplx* __restrict__ tw_big,
                                            double* __restrict__ xbuf, int lane, double* __restrict__ xim = nullptr) {
    typedef Geo<LOGN> G;
    constexpr int R = G::R;
    Tw<G::NLOW - 4> w3; Tw<R - 1> w2, w1; Tw<R> wt;
    w3.load(tw_small + G::TW_P3, 1);
    P3<R, G::NLOW, G::LOW - 1, TRIV>::template inv<false>(re, im, w3.w);
    w2.load(tw_small + G::TW_P2 + (lane & (G::NLOW - 1)), G::NLOW);
    exchange<LOGN, 3, 2, DUAL>(re, im, xbuf, lane, xim);
    P12<R, G::LR - 1>::inv(re, im, w2.w);
    w1.load(tw_big + G::TW_P1 + lane, 64);          // in flight during the exchange
    exchange<LOGN, 2, 1, DUAL>(re, im, xbuf, lane, xim);
    wt.load(tw_big + G::TW_TWIST + lane, 64);
    P12<R, G::LR - 1>::inv(re, im, w1.w);
    twist_mul<R>(re, im, wt.w);
}

// Twiddle staging: the forward table and the inverse table's small part always live in LDS; the inverse table's
// big part (untwist + pass 1) stays in global memory when SPLIT (N = 2048: 31 KiB of LDS = one more resident gate).
template <int LOGN>
struct TwStage {
    typedef Geo<LOGN> G;
    static constexpr bool SPLIT = (LOGN >= 11);
    static constexpr int LDS_CPLX = SPLIT ? G::TW_DIR + (G::TW_DIR - G::TW_P2) : G::TW_TOTAL;
    // copies into `lds` (caller barriers afterwards)
    __device__ __forceinline__ static void stage(cplx* __restrict__ lds, const cplx* __restrict__ global, int tid, int nthreads) {
        if constexpr (SPLIT) {
            for (int idx = tid; idx < G::TW_DIR; idx += nthreads) lds[idx] = global[idx];
            for (int idx = tid + G::TW_P2; idx < G::TW_DIR; idx += nthreads) lds[G::TW_DIR + idx - G::TW_P2] = global[G::TW_DIR + idx];
        } else {
            for (int idx = tid; idx < G::TW_TOTAL; idx += nthreads) lds[idx] = global[idx];
        }
    }
    __device__ __forceinline__ static const cplx* fwd(const cplx* lds) { return lds; }
    __device__ __forceinline__ static const cplx* inv_small(const cplx* lds) { return SPLIT ? lds + G::TW_DIR - G::TW_P2 : lds + G::TW_DIR; }
    __device__ __forceinline__ static const cplx* inv_big(const cplx* lds, const cplx* global) { return SPLIT ? global + G::TW_DIR : lds + G::TW_DIR; }
};

// Torus32(int64_t(x)): truncate toward zero, keep the low 32 bits (fft_processor_spqlios.cpp:182).
// trunc(x) + 1.5*2^52 is exact for |x| < 2^51 (the path's values stay below 2^50: 2l*N*(Bg/2)*2^31),
// and the low word of its mantissa is trunc(x) mod 2^32.
__device__ __forceinline__ uint32_t trunc_to_torus(double x) {
    const double t = __builtin_trunc(x);
    const double y = t + 6755399441055744.0;
    return (uint32_t)__double_as_longlong(y);
}

// The same conversion over the reference's whole range: Torus32(int64_t(x)) is defined for every |x| < 2^63 (cvttsd2si), the
// magic-add form above only below 2^51.  The stage-level entry point rtfhe_fft_u32_batch takes arbitrary spectra and uses
// this one: t = trunc(x) splits exactly into hi * 2^32 + lo (power-of-two scaling, floor and the fused multiply-add below
// are all exact for an integer-valued double), lo in [0, 2^32) is t mod 2^32.  Out of range / NaN gives 0, the low word of
// the 0x8000000000000000 the x86 conversion returns.
__device__ __forceinline__ uint32_t trunc_to_torus_wide(double x) {
    const double t = __builtin_trunc(x);
    if (!(__builtin_fabs(t) < 9223372036854775808.0)) return 0u;
    const double hi = __builtin_floor(t * 2.3283064365386963e-10);       // 2^-32
    const double lo = __builtin_fma(hi, -4294967296.0, t);
    return (uint32_t)lo;
}

// make_decomp_mask(l, bits), utils/src/math.rs:542-560
__host__ __device__ constexpr uint32_t decomp_mask(int l, int bits) {
    uint32_t u = 0;
    if (32 - l * bits != 0) {
        u += 1u << (32 - l * bits - 1);
        for (int i = l; i >= 1; i--) u += 1u << (32 - i * bits - 1);
    } else {
        for (int i = l - 1; i >= 1; i--) u += 1u << (32 - i * bits - 1);
    }
    return u;
}

// digit j of the pre-masked word u = (x + M) ^ M, sign-extended from `bits` (utils/src/math.rs:314-322):
// (v & half) * 0xfffffffe + v  ==  v - 2 (v & half)  ==  the two's-complement value of the `bits`-wide field
__device__ __forceinline__ int32_t decomp_digit(uint32_t u, int bits, int j) {
    return __builtin_amdgcn_sbfe((int32_t)u, (uint32_t)(32 - bits * (j + 1)), (uint32_t)bits);
}

// Flags and counters in LDS, addressed by SCALAR registers.  The step holds 256 live vector registers at its peak and every vector register that
// lives across it (an LDS address of a flag, say) costs a cascade of spills (measured: three such addresses, 2 -> 181 spilled registers); the
// address therefore travels in an SGPR and is moved into a temporary inside the statement.  Each wait is ONE opaque statement (a loop the
// compiler could see would become a block boundary in the middle of the step, with the same effect on the allocator); it sleeps between polls
// so that the poll costs the SIMD's other wave next to nothing.

// publishes k in the flag at LDS address `flag`
__device__ __forceinline__ void flag_arrive(unsigned flag, unsigned k) {
    unsigned t0, t1;
    asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3\n\tds_write_b32 %0, %1" : "=&v"(t0), "=&v"(t1) : "s"(flag), "s"(k) : "memory");
}
// waits until the flag / counter at LDS address `flag` has reached k
__device__ __forceinline__ void flag_wait(unsigned flag, unsigned k) {
    unsigned v, t;
    asm volatile(
        "v_mov_b32 %0, %2\n"
        "1:\n\t"
        "ds_read_b32 %0, %0\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_readfirstlane_b32 %1, %0\n\t"
        "s_sub_i32 %1, %1, %3\n\t"
        "s_cmp_lt_i32 %1, 0\n\t"
        "s_cbranch_scc0 2f\n\t"
        "s_sleep 1\n\t"
        "v_mov_b32 %0, %2\n\t"
        "s_branch 1b\n"
        "2:"
        : "=&v"(v), "=&s"(t)
        : "s"(flag), "s"(k)
        : "memory", "scc");
}
// negacyclic rotate read: coefficient c of X^r * p, p in LDS (utils/src/math.rs:85-132)
template <int LOGN>
__device__ __forceinline__ uint32_t rotated_coef(const uint32_t* __restrict__ p, int c, int r) {
    constexpr int N = 1 << LOGN;
    const int e = (c - r) & (2 * N - 1);
    const uint32_t v = p[e & (N - 1)];
    return (e >> LOGN) ? (0u - v) : v;      // (a branch-free (v ^ s) - s form measured 0.6 % slower in the pair kernel)
}

}  // namespace rtfhe
