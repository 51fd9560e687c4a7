// rtfhe_sub256.hpp -- one PARITY of the 512-point transform (N = 1024) on one wave: 4 points per lane.
//
// The latency kernel (rtfhe_kernels_wg.hpp) has eight waves for one gate and, per CMUX step, phases in which only some of them have a
// whole transform to run.  A transform splits over two waves the way k_bootstrap_eo splits the 1024-point one (rtfhe_kernels_eo.hpp):
// wave H owns the points of parity H, i = 2 j + H.  Every radix-2 stage pairs i with i + halfnn, of the same parity for halfnn >= 2, so
// the twist, the seven twiddled stages halfnn = 256 .. 4 and this parity's half of the size-4 stage (spqlios-fft-impl.cpp:526-603 forward,
// :289-363 inverse) stay inside the wave, as a 256-point network in j; the twiddle of pair (i, i + halfnn) is entry i mod halfnn =
// 2 (j mod halfnn / 2) + H of the reference's stage table.  Only the untwiddled size-2 stage (x0 + x1, x0 + (-x1); :606-634 / :248-269)
// pairs the parities: the caller does it where both values already meet (the spectrum buffer in LDS).
//
// Four points per lane = two stages per in-register pass = four passes and three wave-private exchanges:
//   layout L1: lane t, register m <-> j = t + 64 m                                      (m = bits 7..6)   stages j-halfnn 128, 64
//   layout L2: lane (a, r),  m   <-> j = (a << 6) | (m << 4) | r,  a = lane >> 4, r = lane & 15  (bits 5..4)          32, 16
//   layout L3: lane (b, c),  m   <-> j = (b << 4) | (m << 2) | c,  b = lane >> 2, c = lane & 3   (bits 3..2)           8,  4
//   layout L4: lane v,       m   <-> j = (v << 2) | m                                    (bits 1..0)   j-halfnn 2 and the size-4 half
// Same butterflies (fwd_stage_tw / inv_stage_tw), same operands, same order per point as the one-wave transform: bit-identical.
#pragma once

#include "rtfhe_device.hpp"

namespace rtfhe {

// tables, cplx units: [direction 0 = forward, 1 = inverse][parity][ONE]
struct Q4Tw {
    static constexpr int TW = 0;                 // [4][64]  forward: twist of point i = 2 (lane + 64 m) + H; inverse: untwist times 2/N
    static constexpr int P1 = TW + 4 * 64;       // [3][64]  entries of Tw<3>: e = 0, 1: i-halfnn 256, q = e; e = 2: i-halfnn 128
    static constexpr int P2 = P1 + 3 * 64;       // [3][16]  i-halfnn 64 (q = 0, 1), 32; by r = lane & 15
    static constexpr int P3 = P2 + 3 * 16;       // [3][4]   i-halfnn 16 (q = 0, 1), 8;  by c = lane & 3
    static constexpr int P4 = P3 + 3 * 4;        // [2]      i-halfnn 4, q = 0, 1 (wave-uniform)
    static constexpr int ONE = 512;
    static constexpr int TOTAL = 4 * ONE;
    __host__ __device__ static constexpr int off(int dir, int H) { return (dir * 2 + H) * ONE; }
};

// the twiddles one wave uses for one direction and parity: 15 complex values, loaded once and kept in registers over the whole blind rotation
struct Q4Regs {
    cplx wt[4], w1[3], w2[3], w3[3], w4[2];
    __device__ __forceinline__ void load(const cplx* __restrict__ t, int lane) {
#pragma unroll
        for (int m = 0; m < 4; m++) wt[m] = t[Q4Tw::TW + m * 64 + lane];
#pragma unroll
        for (int e = 0; e < 3; e++) {
            w1[e] = t[Q4Tw::P1 + e * 64 + lane];
            w2[e] = t[Q4Tw::P2 + e * 16 + (lane & 15)];
            w3[e] = t[Q4Tw::P3 + e * 4 + (lane & 3)];
        }
        w4[0] = t[Q4Tw::P4]; w4[1] = t[Q4Tw::P4 + 1];
    }
    __device__ __forceinline__ void get_wt(cplx (&o)[4]) const { for (int m = 0; m < 4; m++) o[m] = wt[m]; }
    __device__ __forceinline__ void get_w1(cplx (&o)[3]) const { for (int e = 0; e < 3; e++) o[e] = w1[e]; }
    __device__ __forceinline__ void get_w2(cplx (&o)[3]) const { for (int e = 0; e < 3; e++) o[e] = w2[e]; }
    __device__ __forceinline__ void get_w3(cplx (&o)[3]) const { for (int e = 0; e < 3; e++) o[e] = w3[e]; }
    __device__ __forceinline__ void get_w4(cplx (&o)[2]) const { o[0] = w4[0]; o[1] = w4[1]; }
};
// ... or read pass by pass from a copy of the table in LDS (k_bootstrap_pair4 at three gates per workgroup: 168 registers per wave leave no room
// for 30 resident complex values); a pass's loads are issued before the exchange in front of it
struct Q4Lds {
    const cplx* t;       // LDS: Q4Tw::ONE entries of this wave's direction and parity
    int lane;
    __device__ __forceinline__ void get_wt(cplx (&o)[4]) const {
#pragma unroll
        for (int m = 0; m < 4; m++) o[m] = t[Q4Tw::TW + m * 64 + lane];
    }
    __device__ __forceinline__ void get_w1(cplx (&o)[3]) const {
#pragma unroll
        for (int e = 0; e < 3; e++) o[e] = t[Q4Tw::P1 + e * 64 + lane];
    }
    __device__ __forceinline__ void get_w2(cplx (&o)[3]) const {
#pragma unroll
        for (int e = 0; e < 3; e++) o[e] = t[Q4Tw::P2 + e * 16 + (lane & 15)];
    }
    __device__ __forceinline__ void get_w3(cplx (&o)[3]) const {
#pragma unroll
        for (int e = 0; e < 3; e++) o[e] = t[Q4Tw::P3 + e * 4 + (lane & 3)];
    }
    __device__ __forceinline__ void get_w4(cplx (&o)[2]) const { o[0] = t[Q4Tw::P4]; o[1] = t[Q4Tw::P4 + 1]; }
};

struct Q4 {
    static constexpr int R = 4;
    static constexpr int XS = 320;               // complex slots of an exchange buffer (the largest padded slot is 318); 16-byte aligned
    // Slot of register m in the buffer of the exchange between layouts X and X + 1: base(lane) + stride * m, with the pads
    //   L1 <-> L2: slot(j) = j + 16 (j >> 6)     L2 <-> L3: j + 4 (j >> 4)     L3 <-> L4: j + (j >> 2)
    // chosen conflict-free on the write AND on the read side.
    template <int X, int LAYOUT>
    __device__ __forceinline__ static int base(int lane) {
        if constexpr (X == 1) return LAYOUT == 1 ? lane : 80 * (lane >> 4) + (lane & 15);
        else if constexpr (X == 2) return LAYOUT == 2 ? 80 * (lane >> 4) + (lane & 15) : 20 * (lane >> 2) + (lane & 3);
        else return LAYOUT == 3 ? 20 * (lane >> 2) + (lane & 3) : 5 * lane;
    }
    template <int X, int LAYOUT>
    __host__ __device__ static constexpr int stride() {
        return X == 1 ? (LAYOUT == 1 ? 80 : 16) : X == 2 ? (LAYOUT == 2 ? 20 : 4) : (LAYOUT == 3 ? 5 : 1);
    }
    // PLANES = false: one 16-byte LDS access per complex value (what the latency kernel wants: its waves wait on their own LDS round trips, and half
    // the instructions are worth 2 %); true: the real and the imaginary parts as 8-byte accesses to two planes of XS doubles in the same buffer (what
    // k_bootstrap_pair4 wants at two waves per SIMD: 4.05 -> 3.89 ms per 512 gates).  profiles/r04/pair4_ab.log
    template <int FROM, int TO, bool PLANES = false>
    __device__ __forceinline__ static void exchange(double (&re)[R], double (&im)[R], cplx* __restrict__ xc, int lane) {
        constexpr int X = FROM < TO ? FROM : TO;
        static_assert(FROM + TO == 2 * X + 1, "neighbouring layouts");
        const int bw = base<X, FROM>(lane), br = base<X, TO>(lane);
        constexpr int sw = stride<X, FROM>(), sr = stride<X, TO>();
      if constexpr (PLANES) {
        double* xre = reinterpret_cast<double*>(xc);
        double* xim = xre + XS;
#pragma unroll
        for (int m = 0; m < R; m++) lds_st(&xre[bw + sw * m], re[m]);
#pragma unroll
        for (int m = 0; m < R; m++) lds_st(&xim[bw + sw * m], im[m]);
        wave_lds_sync();
#pragma unroll
        for (int m = 0; m < R / 2; m++) {
            re[m] = lds_ld(&xre[br + sr * m]); re[m + 2] = lds_ld(&xre[br + sr * (m + 2)]);
            im[m] = lds_ld(&xim[br + sr * m]); im[m + 2] = lds_ld(&xim[br + sr * (m + 2)]);
        }
        wave_lds_sync();
        return;
      }
#pragma unroll
        for (int m = 0; m < R; m++) lds_st128(&xc[bw + sw * m], re[m], im[m]);
        wave_lds_sync();
#pragma unroll
        for (int m = 0; m < R / 2; m++) {        // the next pass's first stage pairs m with m + 2
            const cplx a = lds_ld128(&xc[br + sr * m]), b = lds_ld128(&xc[br + sr * (m + 2)]);
            re[m] = a.x; im[m] = a.y; re[m + 2] = b.x; im[m + 2] = b.y;
        }
        wave_lds_sync();
    }
};

// Forward: in = this parity's points in layout L1 (digits, not yet twisted); out = the sub-network's outputs out_H[j] in layout L4,
// i.e. what the size-2 stage across the parities still has to combine (spectrum point 2 j = out_0[j] + out_1[j], 2 j + 1 = out_0[j] + (-out_1[j])).
struct Q4NoHook { __device__ __forceinline__ void operator()(int) const {} };
// part A: twist and the three passes that need the exchange buffer; part B: the last pass, registers only (a caller with several rows and one
// buffer runs A for all of them first)
template <typename W, typename HOOK = Q4NoHook, bool PLANES = false>
__device__ __forceinline__ void sub256_forward_a(double (&re)[4], double (&im)[4], const W& w, cplx* __restrict__ xc, int lane, HOOK after_exchange = HOOK()) {
    cplx wt[4], w1[3], w2[3], w3[3];
    w.get_wt(wt); w.get_w1(w1);
    twist_mul<4>(re, im, wt);
    P12<4, 1>::fwd(re, im, w1);
    w.get_w2(w2);
    Q4::exchange<1, 2, PLANES>(re, im, xc, lane);
    after_exchange(1);
    P12<4, 1>::fwd(re, im, w2);
    w.get_w3(w3);
    Q4::exchange<2, 3, PLANES>(re, im, xc, lane);
    after_exchange(2);
    P12<4, 1>::fwd(re, im, w3);
    Q4::exchange<3, 4, PLANES>(re, im, xc, lane);
    after_exchange(3);
}
// ... for NR rows side by side, pass by pass: a row's exchange is in flight while the other rows compute (DS instructions of a wave execute in
// order, so the rows share the buffer back to back), and every pass loads nothing (the twiddles are the caller's registers)
template <int NR, bool PLANES, typename W>
__device__ __forceinline__ void sub256_forward_a_multi(double (&re)[NR][4], double (&im)[NR][4], const W& w, cplx* __restrict__ xc, int lane) {
    cplx wt[4], w1[3], w2[3], w3[3];
    w.get_wt(wt); w.get_w1(w1);
#pragma unroll
    for (int j = 0; j < NR; j++) { twist_mul<4>(re[j], im[j], wt); P12<4, 1>::fwd(re[j], im[j], w1); Q4::exchange<1, 2, PLANES>(re[j], im[j], xc, lane); }
    w.get_w2(w2);
#pragma unroll
    for (int j = 0; j < NR; j++) { P12<4, 1>::fwd(re[j], im[j], w2); Q4::exchange<2, 3, PLANES>(re[j], im[j], xc, lane); }
    w.get_w3(w3);
#pragma unroll
    for (int j = 0; j < NR; j++) { P12<4, 1>::fwd(re[j], im[j], w3); Q4::exchange<3, 4, PLANES>(re[j], im[j], xc, lane); }
}
template <bool ODD, bool TRIV, typename W>
__device__ __forceinline__ void sub256_forward_b(double (&re)[4], double (&im)[4], const W& w) {
    cplx w4[2];
    w.get_w4(w4);
    fwd_stage_tw<4, 1, TRIV && !ODD>(re, im, w4);       // i-halfnn 4; entry 0 of parity 0 is the reference's (1, 0): see fwd_stage_tw
    // this parity's half of the size-4 stage (spqlios-fft-impl.cpp:581-602): even points x0, x2 -> x0 + x2, x0 + (-x2); odd points x1, x3 ->
    // x1 + x3, i (x1 - x3) = ((-j1) + j3, r1 + (-r3))
#pragma unroll
    for (int m = 0; m < 4; m += 2) {
        const double ra = re[m], rb = re[m + 1], ja = im[m], jb = im[m + 1];
        if (!ODD) { re[m] = ra + rb; re[m + 1] = ra + (-rb); im[m] = ja + jb; im[m + 1] = ja + (-jb); }
        else      { re[m] = ra + rb; re[m + 1] = (-ja) + jb; im[m] = ja + jb; im[m + 1] = ra + (-rb); }
    }
}
template <bool ODD, bool TRIV, typename W, typename HOOK = Q4NoHook>
__device__ __forceinline__ void sub256_forward(double (&re)[4], double (&im)[4], const W& w, cplx* __restrict__ xc, int lane, HOOK after_exchange = HOOK()) {
    sub256_forward_a(re, im, w, xc, lane, after_exchange);
    sub256_forward_b<ODD, TRIV>(re, im, w);
}

// Inverse: in = in_H[j] in layout L4 (the size-2 stage across the parities already applied: in_0[j] = s[2j] + s[2j + 1], in_1[j] = s[2j] + (-s[2j + 1]));
// out = this parity's coefficients, untwisted (the 2/N of fft_processor_spqlios.cpp:158 is in the table), layout L1.
template <bool ODD, bool TRIV, bool PLANES = false, typename W>
__device__ __forceinline__ void sub256_inverse(double (&re)[4], double (&im)[4], const W& w, cplx* __restrict__ xc, int lane) {
    cplx wt[4], w1[3], w2[3], w3[3], w4[2];
    w.get_w4(w4); w.get_w3(w3);
    // this parity's half of the size-4 stage (:289-310): even x0, x2 -> x0 + x2, x0 + (-x2); odd x1, x3 -> x1 - i x3 = (r1 + j3, j1 + (-r3)),
    // x1 + i x3 = (r1 + (-j3), j1 + r3)
#pragma unroll
    for (int m = 0; m < 4; m += 2) {
        const double ra = re[m], rb = re[m + 1], ja = im[m], jb = im[m + 1];
        if (!ODD) { re[m] = ra + rb; re[m + 1] = ra + (-rb); im[m] = ja + jb; im[m + 1] = ja + (-jb); }
        else      { re[m] = ra + jb; re[m + 1] = ra + (-jb); im[m] = ja + (-rb); im[m + 1] = ja + rb; }
    }
    inv_stage_tw<4, 1, false, TRIV && !ODD>(re, im, w4);
    Q4::exchange<4, 3, PLANES>(re, im, xc, lane);
    P12<4, 1>::inv(re, im, w3);
    w.get_w2(w2);
    Q4::exchange<3, 2, PLANES>(re, im, xc, lane);
    P12<4, 1>::inv(re, im, w2);
    w.get_w1(w1); w.get_wt(wt);
    Q4::exchange<2, 1, PLANES>(re, im, xc, lane);
    P12<4, 1>::inv(re, im, w1);
    twist_mul<4>(re, im, wt);
}

}  // namespace rtfhe
