// rtfhe_ntt.hpp -- exact-integer multiply backend: negacyclic NTT over F_P, P = 2^50 - 16383, N = 1024.
//
// This is the second multiply backend (the one BASELINE's north_star names).  It computes
// the external product's polynomial products EXACTLY (reference semantics of Polynomial::cross, utils/src/math.rs:238-257),
// so its outputs are bit-identical to the oracle's `exact_int` backend -- and therefore NOT to the reference CPU path,
// whose FP64 FFT + truncation deviates from exact arithmetic by +-1 LSB per external product.  Against
// the reference it gives identical decrypted bits and phases within the noise margin.
//
// Arithmetic: doubles holding exact integers (|x| < 2^53), modular product by an error-free FMA split:
//   h = a*w ; l = fma(a, w, -h) ; q = rint(h / P) ; r = fma(-q, P, h) + l        (6 FP64 ops, r == a*w (mod P), |r| < 2.2 P)
// Transforms with the negacyclic twist merged into per-block twiddles (zeta_k = psi^bitrev(k)):
//   forward  Cooley-Tukey butterflies (t = zeta x1 ; x0 +- t), stride N/2 .. 1, natural in  -> bit-reversed out
//   inverse  Gentleman-Sande butterflies (x0 + x1 ; zeta^-1 (x0 - x1)), stride 1 .. N/2, bit-reversed in -> natural out
// Bounds: doubles hold the values exactly while every |value| < 2^53 = 8.0000001 P.  A modular product returns |r| <= P/2 + 3 |a| |w| 2^-53,
// a Cooley-Tukey stage therefore grows a bound B to 1.1875 B + P/2, a Gentleman-Sande stage doubles it.  The LEAN schedule (N = 1024 kernels,
// round 5) renormalises to |x| <= P/2 ONCE inside a forward transform (after stage 7; outputs <= 2.64 P go to the multiply-accumulate as they
// are, six products sum to <= 5.97 P) and on entry, after stages 4 and 8 and at the end of an inverse (four stages from P/2 reach 8 P exactly
// below 2^53).  scripts/ntt/model.py proves these bounds for every input (worst_case_bounds) and runs the operation sequence on exact integers.
// The full schedule (after stages 5 and 10 forward; entry, 3, 6, 9, 10 inverse: 384 FP64-rate instructions per CMUX more) stays for the
// key transform -- whose outputs must be normalised -- and the N = 2048 kernels (one more stage across the halves).  The true result of a gate's
// sum of 2l products is below 2^48.6 < P/2, so the centred residue IS the integer result.  N^-1 is folded into the key's transformed rows.
//
// One wave per transform, 16 points per lane, three in-register passes (4 + 4 + 2 stages) with two wave-private LDS
// exchanges: the index geometry (layouts L1/L2/L3, conflict-free slot maps f1/f2) is Geo<11>'s (1024 points, R = 16).
#pragma once

#include "rtfhe_device.hpp"

namespace rtfhe {
namespace ntt {

typedef Geo<11> GN;                       // 1024 points on 64 lanes: R = 16, LR = 4, LOW = 2
constexpr int N = 1024;
constexpr int R = 16;
constexpr double P = 1125899906826241.0;  // 2^50 - 16383, prime, 2^14 | P - 1
constexpr unsigned long long P_U64 = 1125899906826241ull;
constexpr double PINV = 1.0 / 1125899906826241.0;

// twiddle table (doubles), one per direction: [pass 1: 15 wave-uniform][pass 2: 15 x 16 lane groups][pass 3: 12 x 64 lanes]
constexpr int TW_P1 = 0;
constexpr int TW_P2 = TW_P1 + 15;
constexpr int TW_P3 = TW_P2 + 15 * 16;
constexpr int TW_DIR = TW_P3 + 12 * 64;   // 1023 = N - 1 zetas
constexpr int TW_DIR_PAD = 1024;          // keep the inverse table 16-byte aligned
// digit tables [5][64]: entry e of table k = (e as a signed 6-bit digit) * c_k mod P, centred, for c = zeta_1, zeta_2, zeta_3, zeta_2 zeta_1,
// zeta_3 zeta_1.  The first TWO forward stages act on decomposition DIGITS (64 possible values, Bgbit = 6) and involve three twiddles only
// (zeta_1; zeta_2 and zeta_3 for the two blocks of stage 2): every product of a digit with a twiddle (or a product of two) is one LDS read
// instead of a conversion and 6-instruction modular products (first_two_stages_digits).  Table 0 alone serves the N = 2048 kernels' first stage.
constexpr int DIGITS = 64;
constexpr int DIG_TABLES = 5;
constexpr int TW_DIG = 2 * TW_DIR_PAD;
constexpr int TW_TOTAL = TW_DIG + DIG_TABLES * DIGITS;
constexpr int XSLOTS = GN::XSLOTS;        // 1088 doubles

__device__ __forceinline__ double modmul(double a, double w) {
    const double h = a * w;
    const double l = __builtin_fma(a, w, -h);
    const double q = __builtin_rint(h * PINV);
    return __builtin_fma(-q, P, h) + l;
}
__device__ __forceinline__ double normalize(double x) {
    return __builtin_fma(-__builtin_rint(x * PINV), P, x);
}
template <int CNT>
__device__ __forceinline__ void normalize_all(double (&x)[CNT]) {
#pragma unroll
    for (int m = 0; m < CNT; m++) x[m] = normalize(x[m]);
}

// Cooley-Tukey stage on register bit MB; zeta index = (nb - 1) + (m >> (MB + 1)), nb = R >> (MB + 1)
template <int MB>
__device__ __forceinline__ void fwd_stage(double (&x)[R], const double* z) {
    constexpr int h = 1 << MB, nb = R >> (MB + 1);
#pragma unroll
    for (int m = 0; m < R; m++) {
        if (m & h) continue;
        const double t = modmul(x[m | h], z[nb - 1 + (m >> (MB + 1))]);
        x[m | h] = x[m] - t;
        x[m] = x[m] + t;
    }
}
// Gentleman-Sande stage
template <int MB>
__device__ __forceinline__ void inv_stage(double (&x)[R], const double* z) {
    constexpr int h = 1 << MB, nb = R >> (MB + 1);
#pragma unroll
    for (int m = 0; m < R; m++) {
        if (m & h) continue;
        const double u = x[m], v = x[m | h];
        x[m] = u + v;
        x[m | h] = modmul(u - v, z[nb - 1 + (m >> (MB + 1))]);
    }
}

// (plain accesses: the compiler pairs them into ds_read2_b64 / ds_write2_b64.  Unpaired -- lds_st / lds_ld of rtfhe_device.hpp, which
// gains 1-3 % in the FFT kernels -- measured 1.3 % (N = 1024) and 2.8 % (N = 2048) SLOWER here: 16 values per lane are twice as many DS
// instructions per exchange and the wave's 16-deep LDS queue fills; profiles/r03/lds_paired_vs_unpaired_all_kernels.log)
template <int FROM, int TO>
__device__ __forceinline__ void exchange(double (&x)[R], double* __restrict__ xbuf, int lane) {
    auto slot = [&](int layout, int m) {
        const int pos = layout == 1 ? GN::pos1(lane, m) : layout == 2 ? GN::pos2(lane, m) : GN::pos3(lane, m);
        return (FROM + TO == 3) ? GN::f1(pos) : GN::f2(pos);
    };
#pragma unroll
    for (int m = 0; m < R; m++) xbuf[slot(FROM, m)] = x[m];
    wave_lds_sync();
#pragma unroll
    for (int m = 0; m < R; m++) x[m] = xbuf[slot(TO, m)];
    wave_lds_sync();
}

// Forward transform in two parts so that a caller can issue its global loads between them (their registers are then
// not live through passes 1 and 2).  in: layout L1 (x[m] = coefficient lane + 64 m), small integers or |x| < 2^32.
// out: layout L3 (point (lane << 4) | m), normalised.  tw: LDS forward table.
template <int STAGES_DONE = 0, bool LEAN = false>
__device__ __forceinline__ void forward_a(double (&x)[R], const double* __restrict__ tw, double* __restrict__ xbuf, int lane) {
    double z1[15], z2[15];
#pragma unroll
    for (int e = 0; e < 15; e++) z1[e] = tw[TW_P1 + e];
    if constexpr (STAGES_DONE < 1) fwd_stage<3>(x, z1);
    if constexpr (STAGES_DONE < 2) fwd_stage<2>(x, z1);
    fwd_stage<1>(x, z1); fwd_stage<0>(x, z1);
#pragma unroll
    for (int e = 0; e < 15; e++) z2[e] = tw[TW_P2 + e * 16 + (lane >> 2)];
    exchange<1, 2>(x, xbuf, lane);
    fwd_stage<3>(x, z2);
    if constexpr (!LEAN) normalize_all(x);     // full schedule: after stage 5 (and after stage 10)
    fwd_stage<2>(x, z2); fwd_stage<1>(x, z2);
    if constexpr (LEAN) normalize_all(x);      // lean schedule: after stage 7 only
    fwd_stage<0>(x, z2);
}
template <bool LEAN = false>
__device__ __forceinline__ void forward_b(double (&x)[R], const double* __restrict__ tw, double* __restrict__ xbuf, int lane) {
    double z3[12];
#pragma unroll
    for (int e = 0; e < 12; e++) z3[e] = tw[TW_P3 + e * 64 + lane];
    exchange<2, 3>(x, xbuf, lane);
    // pass 3 touches register bits 1 and 0 only: bit 1 uses entries 0..3 (index m >> 2), bit 0 entries 4..11 (index m >> 1)
#pragma unroll
    for (int m = 0; m < R; m++) {
        if (m & 2) continue;
        const double t = modmul(x[m | 2], z3[m >> 2]);
        x[m | 2] = x[m] - t;
        x[m] = x[m] + t;
    }
#pragma unroll
    for (int m = 0; m < R; m += 2) {
        const double t = modmul(x[m + 1], z3[4 + (m >> 1)]);
        x[m + 1] = x[m] - t;
        x[m] = x[m] + t;
    }
    if constexpr (!LEAN) normalize_all(x);
}
__device__ __forceinline__ void forward(double (&x)[R], const double* __restrict__ tw, double* __restrict__ xbuf, int lane) {
    forward_a(x, tw, xbuf, lane);
    forward_b(x, tw, xbuf, lane);
}

// forward_b cut at its exchange, for a hand-off between two waves (k_bootstrap_ntt_wg): the first wave writes the exchange buffer
// (forward_b_send), the second reads it and runs pass 3 (forward_b_receive).  The caller orders the two (release / acquire).
__device__ __forceinline__ void forward_b_send(const double (&x)[R], double* __restrict__ xbuf, int lane) {
#pragma unroll
    for (int m = 0; m < R; m++) xbuf[GN::f2(GN::pos2(lane, m))] = x[m];
}
template <bool LEAN = false>
__device__ __forceinline__ void forward_b_receive(double (&x)[R], const double* __restrict__ tw, const double* __restrict__ xbuf, int lane) {
    double z3[12];
#pragma unroll
    for (int e = 0; e < 12; e++) z3[e] = tw[TW_P3 + e * 64 + lane];
#pragma unroll
    for (int m = 0; m < R; m++) x[m] = xbuf[GN::f2(GN::pos3(lane, m))];
#pragma unroll
    for (int m = 0; m < R; m++) {
        if (m & 2) continue;
        const double t = modmul(x[m | 2], z3[m >> 2]);
        x[m | 2] = x[m] - t;
        x[m] = x[m] + t;
    }
#pragma unroll
    for (int m = 0; m < R; m += 2) {
        const double t = modmul(x[m + 1], z3[4 + (m >> 1)]);
        x[m + 1] = x[m] - t;
        x[m] = x[m] + t;
    }
    if constexpr (!LEAN) normalize_all(x);
}

// The first TWO stages on decomposition digits.  u[m] = the pre-masked decomposition word of coefficient lane + 64 m; with a, b, c, d the digits
// j of coefficients lane + 64 g + {0, 256, 512, 768} (registers g, g + 4, g + 8, g + 12), the Cooley-Tukey stages 1 (stride 512, zeta_1) and 2
// (stride 256, zeta_2 for the lower block, zeta_3 for the upper) give
//   x[g]     = (a + z1 c) + (z2 b + z2 z1 d)        x[g + 4]  = (a + z1 c) - (z2 b + z2 z1 d)
//   x[g + 8] = (a - z1 c) + (z3 b - z3 z1 d)        x[g + 12] = (a - z1 c) - (z3 b - z3 z1 d)
// with every product read from a digit table (centred residues, |.| <= P/2: |x| <= 1.5 P + 32): 1 conversion + 8 sums per four points where
// a table for stage 1 and modular products for stage 2 took 22 FP64-rate instructions.  forward_a<2> continues with stage 3.
__device__ __forceinline__ int digit_entry(uint32_t u, int bits, int j);
__device__ __forceinline__ void first_two_stages_digits(double (&x)[R], const uint32_t (&u)[R], int bits, int jj, const double* __restrict__ dig) {
#pragma unroll
    for (int g = 0; g < R / 4; g++) {
        const int ib = digit_entry(u[g + 4], bits, jj), ic = digit_entry(u[g + 8], bits, jj), id = digit_entry(u[g + 12], bits, jj);
        const double a = (double)decomp_digit(u[g], bits, jj);
        const double zc = dig[ic], z2b = dig[DIGITS + ib], z3b = dig[2 * DIGITS + ib], z2d = dig[3 * DIGITS + id], z3d = dig[4 * DIGITS + id];
        const double s = a + zc, t = a - zc, p = z2b + z2d, q = z3b - z3d;
        x[g] = s + p; x[g + 4] = s - p; x[g + 8] = t + q; x[g + 12] = t - q;
    }
}
// byte offset of a digit's entry in the digit table: field j (from the top) of the pre-masked word u = (x + M) ^ M, times 8
__device__ __forceinline__ int digit_entry(uint32_t u, int bits, int j) {
    return (int)__builtin_amdgcn_ubfe(u, (uint32_t)(32 - bits * (j + 1)), (uint32_t)bits);
}

// in: layout L3, |x| < 2^52.  out: layout L1, the centred residue (= the exact integer when |true value| < P/2).
// tw3: the table the per-lane pass-3 entries are read from (the N = 2048 kernel keeps that part in global memory).
template <bool LEAN = false>
__device__ __forceinline__ void inverse(double (&x)[R], const double* __restrict__ tw, const double* __restrict__ tw3,
                                        double* __restrict__ xbuf, int lane) {
    double z1[15], z2[15], z3[12];
#pragma unroll
    for (int e = 0; e < 12; e++) z3[e] = tw3[TW_P3 + e * 64 + lane];
    normalize_all(x);
#pragma unroll
    for (int m = 0; m < R; m += 2) {
        const double u = x[m], v = x[m + 1];
        x[m] = u + v;
        x[m + 1] = modmul(u - v, z3[4 + (m >> 1)]);
    }
#pragma unroll
    for (int m = 0; m < R; m++) {
        if (m & 2) continue;
        const double u = x[m], v = x[m | 2];
        x[m] = u + v;
        x[m | 2] = modmul(u - v, z3[m >> 2]);
    }
#pragma unroll
    for (int e = 0; e < 15; e++) z2[e] = tw[TW_P2 + e * 16 + (lane >> 2)];
    exchange<3, 2>(x, xbuf, lane);
    if constexpr (LEAN) {             // sums double per stage: four stages from P/2 stay below 2^53 -- renormalise after stages 4, 8 and 10
        inv_stage<0>(x, z2); inv_stage<1>(x, z2);
        normalize_all(x);
        inv_stage<2>(x, z2); inv_stage<3>(x, z2);
    } else {                          // full schedule: after stages 3, 6, 9 and 10
        inv_stage<0>(x, z2);
        normalize_all(x);
        inv_stage<1>(x, z2); inv_stage<2>(x, z2); inv_stage<3>(x, z2);
        normalize_all(x);
    }
#pragma unroll
    for (int e = 0; e < 15; e++) z1[e] = tw[TW_P1 + e];
    exchange<2, 1>(x, xbuf, lane);
    if constexpr (LEAN) {
        inv_stage<0>(x, z1); inv_stage<1>(x, z1);
        normalize_all(x);
        inv_stage<2>(x, z1); inv_stage<3>(x, z1);
    } else {
        inv_stage<0>(x, z1); inv_stage<1>(x, z1); inv_stage<2>(x, z1);
        normalize_all(x);
        inv_stage<3>(x, z1);
    }
    normalize_all(x);
}

template <bool LEAN = false>
__device__ __forceinline__ void inverse(double (&x)[R], const double* __restrict__ tw, double* __restrict__ xbuf, int lane) {
    inverse<LEAN>(x, tw, tw, xbuf, lane);
}

// The inverse transform read from a FORWARD table.  psi^N = -1 gives zeta_{nb + b}^-1 = -zeta_{nb + (nb - 1 - b)} on every level
// (nb blocks): the inverse twiddles are the forward ones with the block index reversed inside each level, negated -- and
// (u - v) (-z) = (v - u) z, so the negation costs nothing.  `twm` is the forward table of the mirrored transform: the transform's
// own forward table for a whole N = 1024 transform, the OTHER half's for a half of the N = 2048 transform (block b of half H
// is block H nb + b of the level, its mirror lies in half 1 - H).  Same values bit for bit as inverse() with an inverse table.
__host__ __device__ constexpr int rev15(int e) {      // entry nb - 1 + idx of a 15-entry level list -> nb - 1 + (nb - 1 - idx)
    int nb = 1;
    while (2 * nb <= e + 1) nb *= 2;
    return nb - 1 + (nb - 1 - (e + 1 - nb));
}
template <int MB>
__device__ __forceinline__ void inv_stage_rev(double (&x)[R], const double* z) {
    constexpr int h = 1 << MB, nb = R >> (MB + 1);
#pragma unroll
    for (int m = 0; m < R; m++) {
        if (m & h) continue;
        const double u = x[m], v = x[m | h];
        x[m] = u + v;
        x[m | h] = modmul(v - u, z[nb - 1 + (m >> (MB + 1))]);
    }
}
__device__ __forceinline__ void inverse_rev(double (&x)[R], const double* __restrict__ twm, double* __restrict__ xbuf, int lane) {
    double z1[15], z2[15], z3[12];
    const int rl = 63 - lane;
#pragma unroll
    for (int e = 0; e < 4; e++) z3[e] = twm[TW_P3 + (3 - e) * 64 + rl];
#pragma unroll
    for (int e = 0; e < 8; e++) z3[4 + e] = twm[TW_P3 + (4 + 7 - e) * 64 + rl];
    normalize_all(x);
#pragma unroll
    for (int m = 0; m < R; m += 2) {
        const double u = x[m], v = x[m + 1];
        x[m] = u + v;
        x[m + 1] = modmul(v - u, z3[4 + (m >> 1)]);
    }
#pragma unroll
    for (int m = 0; m < R; m++) {
        if (m & 2) continue;
        const double u = x[m], v = x[m | 2];
        x[m] = u + v;
        x[m | 2] = modmul(v - u, z3[m >> 2]);
    }
#pragma unroll
    for (int e = 0; e < 15; e++) z2[e] = twm[TW_P2 + rev15(e) * 16 + 15 - (lane >> 2)];
    exchange<3, 2>(x, xbuf, lane);
    inv_stage_rev<0>(x, z2);
    normalize_all(x);
    inv_stage_rev<1>(x, z2); inv_stage_rev<2>(x, z2); inv_stage_rev<3>(x, z2);
    normalize_all(x);
#pragma unroll
    for (int e = 0; e < 15; e++) z1[e] = twm[TW_P1 + rev15(e)];
    exchange<2, 1>(x, xbuf, lane);
    inv_stage_rev<0>(x, z1); inv_stage_rev<1>(x, z1); inv_stage_rev<2>(x, z1);
    normalize_all(x);
    inv_stage_rev<3>(x, z1);
    normalize_all(x);
}

// exact integer (|x| < 2^51) -> torus word: the low 32 bits of the two's-complement value
__device__ __forceinline__ uint32_t to_torus(double x) {
    return (uint32_t)__double_as_longlong(x + 6755399441055744.0);
}

}  // namespace ntt
}  // namespace rtfhe
