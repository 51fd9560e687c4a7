// rtfhe_kernels.hpp -- gfx950 kernels of the HomNAND hot path (one wavefront = one gate).
//
// Reference functions restated here (paths relative to the reference repo root):
//   gate pre-step            hom_nand/src/tfhe.rs:27-71
//   blind rotate             hom_nand/src/tfhe.rs:89-113
//   CMUX / external product  hom_nand/src/trgsw.rs:264-321, utils/src/spqlios.rs:204-222
//   gadget decomposition     utils/src/math.rs:300-326 (mask :542-560)
//   negacyclic rotate        utils/src/math.rs:85-132
//   sample extract           hom_nand/src/trlwe.rs:110-121
//   identity key switch      hom_nand/src/tlwe.rs:43-73
#pragma once

#include "rtfhe_device.hpp"

namespace rtfhe {

enum { OP_NAND = 0, OP_AND = 1, OP_OR = 2, OP_XOR = 3, OP_NOT = 4, OP_COPY = 5, OP_ANDNY = 6 };
enum { MODE_GATE = 0, MODE_BLIND_ROTATE = 1,
       MODE_EXTRACT = 2 };   // blind rotate + sample extract into `ext`, for the batch key switch that follows (rtfhe_kernels_ksmm.hpp)

struct BootstrapArgs {
    const cplx* tw;          // [Geo::TW_TOTAL] forward table then inverse table
    const cplx* bk;          // device layout [n][2l][2][R][64]
    const uint32_t* ksk;     // device layout [N*(t/2)*(base^2-1) + 1][ksw] (pairs of levels pre-summed, see ks_accumulate); last row all zero
    const uint32_t* in0;     // [count][n+1]
    const uint32_t* in1;     // [count][n+1] (may alias in0)
    uint32_t* out;           // MODE_GATE / MODE_EXTRACT: [count][n+1] (or the wire table);  MODE_BLIND_ROTATE: [count][2][N]
    uint32_t* ext;           // MODE_EXTRACT: lvl1 samples (a'[0..N), b') of the whole batch in the batch key switch's operand order (ext_slot
                             // below), gate number = ext_first + gate number within the launch; the gate's `out` row is ZEROED (the batch key
                             // switch adds its K-slices into it with wrapping atomics)
    int32_t count;
    int32_t op;
    int32_t n;
    int32_t steps;           // CMUX steps to run (= n for a real gate)
    int32_t mode;
    int32_t ksw;             // padded KSK row width in u32 (multiple of 4)
    int32_t npad;            // per-wave LDS words reserved for the mod-switched mask (>= n+1)
    int32_t ext_first;       // MODE_EXTRACT: the batch-wide number of this launch's first gate (segments of a batch share one sample buffer)
    // netlist mode (all null for a plain batch): gate g reads rows idx0[g], idx1[g] of in0 (the wire table), applies
    // ops[g] and writes row idx_out[g] of out
    const int32_t* ops;
    const int32_t* idx0;
    const int32_t* idx1;
    const int32_t* idx_out;
    int32_t num_wires;       // netlist mode: rows in the wire table; indices and opcodes are checked against it on the device
    int32_t* fault;          // netlist mode: set to 1 when a gate was skipped for an out-of-range index / unknown opcode
    unsigned long long* dbg;   // diagnostic builds only (RTFHE_WG_STAMPS): per-phase cycle sums of workgroup 0
};

// Where the split path keeps a batch's lvl1 samples between its two launches: the operand order of k_key_switch_mm (rtfhe_kernels_ksmm.hpp).
// Tiles of 16 gates, 16 N + 16 words each: [coefficient group kk / 4 (N / 16)][lane = 16 q + gate % 16][kk % 4] for coefficient
// c = q N/4 + kk, then the 16 b' words -- a wave of the key switch reads the four coefficients of its 64 (gate, q) rows as ONE contiguous
// KiB.  (Rows of N + 1 words, as the reference stores a TLWE sample, made that read 64 pieces of 16 B in 64 cache lines, re-read by every
// one of the 40 column groups: 10 GB through the L2 per 8,192 gates, the launch's bound until round 5.)  c == N addresses b'.
__device__ __forceinline__ size_t ext_tile_words(int N) { return 16 * (size_t)N + 16; }
__device__ __forceinline__ uint32_t* ext_slot(uint32_t* ext, int g, int c, int N) {
    uint32_t* tile = ext + (size_t)(g >> 4) * ext_tile_words(N);
    if (c == N) return tile + 16 * N + (g & 15);
    const int quarter = N >> 2, q = c / quarter, kk = c & (quarter - 1);
    return tile + ((((kk >> 2) * 64 + q * 16 + (g & 15)) << 2) | (kk & 3));
}

// gate pre-step on one TLWE word (a-part: isb = false, b-part: isb = true), hom_nand/src/tfhe.rs:27-71
__device__ __forceinline__ uint32_t gate_linear(int op, uint32_t x0, uint32_t x1, bool isb) {
    const uint32_t c8 = 0x20000000u;   // torus!(1/8), utils/src/math.rs:691-696
    const uint32_t c4 = 0x40000000u;   // torus!(2 * 1/8)
    switch (op) {
        case OP_NAND: return (isb ? c8 : 0u) - (x0 + x1);
        case OP_AND:  return (x0 + x1) - (isb ? c8 : 0u);
        case OP_OR:   return (x0 + x1) + (isb ? c8 : 0u);
        case OP_XOR:  return (x0 + x1) * 2u + (isb ? c4 : 0u);
        case OP_NOT:  return 0u - x0;
        case OP_ANDNY: return (x1 - x0) - (isb ? c8 : 0u);   // hom_and(-x0, x1), the second AND of hom_mux (tfhe.rs:34)
        default:      return x0;
    }
}

// Where gate g reads and writes.  Netlist gates with a wire index outside [0, num_wires) or an unknown opcode are
// skipped (ok = false: they run on row 0 and store nothing) and reported through *fault -- never dereferenced.
struct GateIo { const uint32_t* p0; const uint32_t* p1; uint32_t* out; int op; bool ok; };
__device__ __forceinline__ GateIo gate_io(const BootstrapArgs& a, int g) {
    const size_t w = (size_t)a.n + 1;
    GateIo io;
    if (a.idx0) {
        int i0 = a.idx0[g], i1 = a.idx1[g], o = a.idx_out[g];
        io.op = a.ops[g];
        const unsigned nw = (unsigned)a.num_wires;
        io.ok = (unsigned)i0 < nw && (unsigned)i1 < nw && (unsigned)o < nw && (unsigned)io.op <= (unsigned)OP_ANDNY;
        if (!io.ok) { i0 = 0; i1 = 0; o = 0; io.op = OP_COPY; if (a.fault) *a.fault = 1; }
        io.p0 = a.in0 + (size_t)i0 * w; io.p1 = a.in0 + (size_t)i1 * w; io.out = a.out + (size_t)o * w;
    } else {
        io.p0 = a.in0 + (size_t)g * w; io.p1 = a.in1 + (size_t)g * w; io.out = a.out + (size_t)g * w;
        io.op = a.op; io.ok = true;
    }
    return io;
}

// One external product / CMUX on the wave-private accumulator in LDS.
//   CMUX = true : acc <- cross(bk_i, X^r * acc - acc) + acc      (trgsw.rs:319-321, tfhe.rs:103-110)
//   CMUX = false: acc <- cross(bk_i, acc)                          (trgsw.rs:264-306)
// accbuf: LDS u32 [2][N] (b then a).  bk_i: this TRGSW in device layout [2l][2][R][64] cplx.
template <int LOGN, int L, int BGBIT, bool CMUX, bool DUAL = false>
__device__ __forceinline__ void cmux_step(uint32_t* __restrict__ accbuf, int r, const cplx* __restrict__ bk_i,
                                          const cplx* __restrict__ twf, const cplx* __restrict__ twi, const cplx* __restrict__ twi_big,
                                          double* __restrict__ xbuf, int lane) {
    typedef Geo<LOGN> G;
    constexpr int N = G::N, P = G::P, R = G::R;
    constexpr uint32_t M = decomp_mask(L, BGBIT);

    double s0re[R], s0im[R], s1re[R], s1im[R];
#pragma unroll
    for (int m = 0; m < R; m++) { s0re[m] = 0.0; s0im[m] = 0.0; s1re[m] = 0.0; s1im[m] = 0.0; }

#pragma unroll 1
    for (int h = 0; h < 2; h++) {
        const uint32_t* poly = accbuf + h * N;
        uint32_t u[2 * R];
#pragma unroll
        for (int mm = 0; mm < 2 * R; mm++) {
            const int c = lane + 64 * mm;
            const uint32_t own = poly[c];
            const uint32_t d = CMUX ? (rotated_coef<LOGN>(poly, c, r) - own) : own;
            u[mm] = (d + M) ^ M;
        }
#pragma unroll 1
        for (int jj = 0; jj < L; jj++) {
            const cplx* bkj = bk_i + (size_t)((h * L + jj) * 2) * R * 64 + lane;
            double re[R], im[R];
#pragma unroll
            for (int m = 0; m < R; m++) {
                re[m] = (double)decomp_digit(u[m], BGBIT, jj);
                im[m] = (double)decomp_digit(u[R + m], BGBIT, jj);
            }
            fft_forward_a<LOGN, DUAL>(re, im, twf, xbuf, lane);
            // the two BK rows of this digit are requested here, not earlier: their 64 VGPRs would otherwise be
            // live through the whole transform; the last exchange + in-register pass cover the L2 latency
            __builtin_amdgcn_sched_barrier(0);
            cplx b0[R], b1[R];
#pragma unroll
            for (int m = 0; m < R; m++) { b0[m] = bkj[m * 64]; b1[m] = bkj[(R + m) * 64]; }
            __builtin_amdgcn_sched_barrier(0);
            fft_forward_b<LOGN, DUAL>(re, im, twf, xbuf, lane);
            // hadamard + fold-add from zero, utils/src/spqlios.rs:204-222, hom_nand/src/trgsw.rs:290-299
#pragma unroll
            for (int m = 0; m < R; m++) {
                {
                    const double ii = b0[m].y * im[m], rr = b0[m].x * re[m], ri = b0[m].x * im[m], ir = b0[m].y * re[m];
                    s0re[m] = s0re[m] + (rr - ii);
                    s0im[m] = s0im[m] + (ir + ri);
                }
                {
                    const double ii = b1[m].y * im[m], rr = b1[m].x * re[m], ri = b1[m].x * im[m], ir = b1[m].y * re[m];
                    s1re[m] = s1re[m] + (rr - ii);
                    s1im[m] = s1im[m] + (ir + ri);
                }
            }
        }
    }

    // the 2/N input scaling of the reference (fft_processor_spqlios.cpp:158) is folded into the untwist twiddles
#pragma unroll 1
    for (int comp = 0; comp < 2; comp++) {
        double re[R], im[R];
#pragma unroll
        for (int m = 0; m < R; m++) {
            re[m] = comp ? s1re[m] : s0re[m];
            im[m] = comp ? s1im[m] : s0im[m];
        }
        fft_inverse<LOGN, DUAL>(re, im, twi, twi_big, xbuf, lane);
        uint32_t* poly = accbuf + comp * N;
#pragma unroll
        for (int m = 0; m < R; m++) {
            const int c = lane + 64 * m;
            const uint32_t x0 = trunc_to_torus(re[m]), x1 = trunc_to_torus(im[m]);
            if (CMUX) { poly[c] += x0; poly[c + P] += x1; }
            else      { poly[c] = x0;  poly[c + P] = x1; }
        }
    }
    wave_lds_sync();
}

// identity key switch of the lvl1 sample held as a'[0..N) in LDS (+ b'), hom_nand/src/tlwe.rs:43-73.
// ks_accumulate sums the selected rows for coefficients [i_begin, i_end) into per-lane uint4 accumulators (lane holds
// columns 4 (lane + 64 q) .. +3).  Rows are added in a different order than the reference's (i, l) loop: wrapping u32
// addition is commutative and associative, so the result is bit-identical.
//
// For the same reason the device copy of the key holds, per coefficient, the rows of KS_GROUP = 2 adjacent levels already
// summed: row (i, p, c) = KS[i][2p][d0] + KS[i][2p+1][d1] with c = d0 * base + d1 in 1 .. base^2 - 1 (k_ksk_combine).  A gate
// then gathers t/2 rows per coefficient of which 15/16 are non-zero instead of t rows of which 3/4 are: 9.8 MB instead of
// 15.6 MB per gate through the L2 (the key switch is L2-bandwidth bound: all gates of a launch reach it together), for a
// 2.5x larger table (156 MB, still resident in the Infinity Cache next to the bootstrapping key).
constexpr int KS_GROUP = 2;
__host__ __device__ constexpr int ks_dev_rows(int N, int t, int basebit) { return N * (t / KS_GROUP) * ((1 << (basebit * KS_GROUP)) - 1); }

template <int LOGN, int KS_T, int KS_BB, int KSQ, int KS_UI = 2 * KS_GROUP>
__device__ __forceinline__ void ks_accumulate(const uint32_t* __restrict__ aprime, int i_begin, int i_end,
                                              const uint32_t* __restrict__ ksk, int ksw, uint4 (&sum)[KSQ], int lane) {
    constexpr int N = 1 << LOGN;
    static_assert(KS_T % KS_GROUP == 0, "levels are grouped in pairs");
    constexpr int PT = KS_T / KS_GROUP, PBB = KS_BB * KS_GROUP, PBASE1 = (1 << PBB) - 1;
    constexpr uint32_t ROUND = (32 - KS_T * KS_BB) != 0 ? (1u << (32 - KS_T * KS_BB - 1)) : 0u;
    const int zero_row = ks_dev_rows(N, KS_T, KS_BB);
    // lanes past the end of a row re-read its last 16 bytes (branch-free); the columns they accumulate are never stored
    int idx[KSQ];
#pragma unroll
    for (int q = 0; q < KSQ; q++) { sum[q] = make_uint4(0, 0, 0, 0); idx[q] = min(lane + 64 * q, ksw / 4 - 1); }
    // KS_UI coefficients per iteration: KS_UI * PT rows (3 x 16 B per lane each) in flight per wave (4 by default: 192 VGPRs)
#pragma unroll 1
    for (int i = i_begin; i < i_end; i += KS_UI) {
        uint4 v[KS_UI][PT][KSQ];
#pragma unroll
        for (int k = 0; k < KS_UI; k++) {
            const uint32_t u = (uint32_t)__builtin_amdgcn_readfirstlane((int)(aprime[i + k] + ROUND));
#pragma unroll
            for (int l = 0; l < PT; l++) {
                const uint32_t c = (u >> (32 - PBB * (l + 1))) & ((1u << PBB) - 1u);          // digits 2l, 2l+1 side by side
                const int row = c ? (((i + k) * PT + l) * PBASE1 + (int)c - 1) : zero_row;   // both digits 0 -> the shared all-zero row
                const uint4* p = reinterpret_cast<const uint4*>(ksk + (size_t)row * ksw);
#pragma unroll
                for (int q = 0; q < KSQ; q++) v[k][l][q] = p[idx[q]];
            }
        }
#pragma unroll
        for (int k = 0; k < KS_UI; k++)
#pragma unroll
            for (int l = 0; l < PT; l++)
#pragma unroll
                for (int q = 0; q < KSQ; q++) {
                    sum[q].x += v[k][l][q].x; sum[q].y += v[k][l][q].y; sum[q].z += v[k][l][q].z; sum[q].w += v[k][l][q].w;
                }
    }
}

// device key-switching key from the reference's [N][t][base-1] rows (padded to ksw words, + the all-zero row at `zero_src`):
// one block per output row (i, p, c)
struct KskCombineArgs {
    const uint32_t* raw;     // [N*t*(base-1) + 1][ksw]
    uint32_t* out;           // [ks_dev_rows + 1][ksw]; the last row is all zero
    int32_t N, ksw;
};
template <int KS_T, int KS_BB>
__global__ __launch_bounds__(256) void k_ksk_combine(const KskCombineArgs a) {
    constexpr int BASE1 = (1 << KS_BB) - 1, PT = KS_T / KS_GROUP, PBB = KS_BB * KS_GROUP, PBASE1 = (1 << PBB) - 1;
    const int rows = a.N * PT * PBASE1, zero_src = a.N * KS_T * BASE1;
    for (int r = blockIdx.x; r <= rows; r += gridDim.x) {
        uint32_t* dst = a.out + (size_t)r * a.ksw;
        if (r == rows) { for (int w = threadIdx.x; w < a.ksw; w += blockDim.x) dst[w] = 0u; continue; }
        const int c = r % PBASE1 + 1, ip = r / PBASE1, p = ip % PT, i = ip / PT;
        const int d0 = c >> KS_BB, d1 = c & BASE1;
        const int s0 = d0 ? ((i * KS_T + 2 * p) * BASE1 + d0 - 1) : zero_src;
        const int s1 = d1 ? ((i * KS_T + 2 * p + 1) * BASE1 + d1 - 1) : zero_src;
        const uint32_t* r0 = a.raw + (size_t)s0 * a.ksw;
        const uint32_t* r1 = a.raw + (size_t)s1 * a.ksw;
        for (int w = threadIdx.x; w < a.ksw; w += blockDim.x) dst[w] = r0[w] + r1[w];
    }
}

template <int LOGN, int KS_T, int KS_BB, int KSQ>
__device__ __forceinline__ void key_switch_wave(const uint32_t* __restrict__ aprime, uint32_t bprime,
                                                const uint32_t* __restrict__ ksk, int ksw, int n,
                                                uint32_t* __restrict__ out, int lane) {
    uint4 sum[KSQ];
    ks_accumulate<LOGN, KS_T, KS_BB, KSQ>(aprime, 0, 1 << LOGN, ksk, ksw, sum, lane);
#pragma unroll
    for (int q = 0; q < KSQ; q++) {
        const int col = 4 * (lane + 64 * q);
        const uint32_t s[4] = {sum[q].x, sum[q].y, sum[q].z, sum[q].w};
#pragma unroll
        for (int e = 0; e < 4; e++)
            if (col + e <= n) out[col + e] = ((col + e == n) ? bprime : 0u) - s[e];
    }
}

// one wave per SIMD (<= 4 waves per workgroup, N = 1024): separate real/imaginary exchange buffers (see exchange<>)
__host__ __device__ constexpr bool bootstrap_dual_xbuf(int logn, int waves) { return logn == 10 && waves <= 4; }
template <int LOGN>
__host__ __device__ constexpr size_t bootstrap_wave_lds_bytes(int npad, bool dual = false) {
    return (size_t)Geo<LOGN>::XSLOTS * sizeof(double) * (dual ? 2 : 1) + (size_t)2 * Geo<LOGN>::N * 4 + (size_t)npad * 4;
}
template <int LOGN>
__host__ __device__ constexpr size_t bootstrap_lds_bytes(int waves, int npad, bool dual = false) {
    return (size_t)TwStage<LOGN>::LDS_CPLX * sizeof(cplx) + (size_t)waves * bootstrap_wave_lds_bytes<LOGN>(npad, dual);
}

// The hot-path kernel: pre-step, blind rotate (n CMUX steps), sample extract and identity key switch
// of `count` independent gates in ONE launch; wave w of block b owns gate b * WAVES + w from start to end.
template <int LOGN, int L, int BGBIT, int KS_T, int KS_BB, int KSQ, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void k_bootstrap(const BootstrapArgs a) {
    typedef Geo<LOGN> G;
    constexpr int N = G::N, R = G::R;
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    cplx* tw = reinterpret_cast<cplx*>(smem);
    TwStage<LOGN>::stage(tw, a.tw, tid, 64 * WAVES);
    __syncthreads();
    // from here on waves never synchronise with each other

    const int g = blockIdx.x * WAVES + wave;
    if (g >= a.count) return;

    constexpr bool DUAL = bootstrap_dual_xbuf(LOGN, WAVES);
    unsigned char* wbase = smem + (size_t)TwStage<LOGN>::LDS_CPLX * sizeof(cplx) + (size_t)wave * bootstrap_wave_lds_bytes<LOGN>(a.npad, DUAL);
    double* xbuf = reinterpret_cast<double*>(wbase);
    uint32_t* accbuf = reinterpret_cast<uint32_t*>(wbase + (size_t)G::XSLOTS * sizeof(double) * (DUAL ? 2 : 1));
    uint32_t* abar = accbuf + 2 * N;
    const cplx* twf = TwStage<LOGN>::fwd(tw);
    const cplx* twi = TwStage<LOGN>::inv_small(tw);
    const cplx* twi_big = TwStage<LOGN>::inv_big(tw, a.tw);

    const int n = a.n;
    // pre-step + mod switch (tfhe.rs:97, 107-108): b floor, a_i rounded, both to [0, 2N)
    const GateIo io = gate_io(a, g);
    if (!io.ok) return;
    {
        constexpr int SH = 32 - LOGN - 1;
        for (int i = lane; i <= n; i += 64) {
            const uint32_t t = gate_linear(io.op, io.p0[i], io.p1[i], i == n);
            abar[i] = (i == n) ? (t >> SH) : ((t + (1u << (SH - 1))) >> SH);
        }
    }
    wave_lds_sync();
    // acc = X^{-bbar} * testvec, testvec = (1/8, ..., 1/8 ; 0)   (tfhe.rs:85, 98-106)
    {
        const int bbar = (int)abar[n];
#pragma unroll
        for (int mm = 0; mm < 2 * R; mm++) {
            const int c = lane + 64 * mm;
            const int e = (c + bbar) & (2 * N - 1);
            accbuf[c] = (e >> LOGN) ? 0xE0000000u : 0x20000000u;
            accbuf[N + c] = 0u;
        }
    }
    wave_lds_sync();

    const size_t trgsw_cplx = (size_t)2 * L * 2 * R * 64;
#pragma unroll 1
    for (int i = 0; i < a.steps; i++) {
        const int r = __builtin_amdgcn_readfirstlane((int)abar[i]);
        cmux_step<LOGN, L, BGBIT, true, DUAL>(accbuf, r, a.bk + (size_t)i * trgsw_cplx, twf, twi, twi_big, xbuf, lane);
    }

    if (a.mode == MODE_BLIND_ROTATE) {
        uint32_t* o = a.out + (size_t)g * 2 * N;
        for (int c = lane; c < 2 * N; c += 64) o[c] = accbuf[c];
        return;
    }

    // sample extract index 0 (trlwe.rs:110-121): a'_0 = a_0, a'_k = -a_{N-k}; b' = b_0
    uint32_t av[2 * R];
#pragma unroll
    for (int mm = 0; mm < 2 * R; mm++) av[mm] = accbuf[N + lane + 64 * mm];
    const uint32_t bprime = accbuf[0];
    wave_lds_sync();
#pragma unroll
    for (int mm = 0; mm < 2 * R; mm++) {
        const int c = lane + 64 * mm;
        accbuf[N + ((N - c) & (N - 1))] = (c == 0) ? av[mm] : (0u - av[mm]);
    }
    wave_lds_sync();
    key_switch_wave<LOGN, KS_T, KS_BB, KSQ>(accbuf + N, bprime, a.ksk, a.ksw, n, io.out, lane);
}

// ------------------------------------------------------------------------------------------------
// stage-level kernels (one wave per item)
// ------------------------------------------------------------------------------------------------

struct FftArgs {
    const cplx* tw;
    const void* src;
    void* dst;
    int32_t count;
    int32_t dst_layout;   // forward: 0 = FrrSeries (Re[0..P) | Im[0..P)), 1 = device BK layout [R][64] cplx
    int32_t rows;         // dst_layout 1: 2l; source polys ordered [i][comp][row], device polys [i][row][comp]
    int32_t f64_io;       // forward: the source is double[count][N] (Spqlios_ifft); inverse: the result is double[count][N] (Spqlios_fft)
};

// source poly index (i, comp, row) -> device poly index (i, row, comp)
__host__ __device__ inline size_t bk_poly_remap(size_t g, int rows) {
    const size_t i = g / (2 * (size_t)rows), rem = g % (2 * (size_t)rows);
    const size_t comp = rem / rows, row = rem % rows;
    return (i * rows + row) * 2 + comp;
}

// Spqlios_ifft_i32 / _u32 (spqlios-wrapper.cpp:22-28): count polynomials of N int32 -> FrrSeries
template <int LOGN, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void k_fft_forward(const FftArgs a) {
    typedef Geo<LOGN> G;
    constexpr int N = G::N, P = G::P, R = G::R;
    extern __shared__ __align__(16) unsigned char smem[];
    cplx* tw = reinterpret_cast<cplx*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int idx = tid; idx < G::TW_DIR; idx += 64 * WAVES) tw[idx] = a.tw[idx];
    __syncthreads();
    double* xbuf = reinterpret_cast<double*>(tw + G::TW_DIR) + (size_t)wave * G::XSLOTS;
    for (int g = blockIdx.x * WAVES + wave; g < a.count; g += gridDim.x * WAVES) {
        double re[R], im[R];
        if (a.f64_io) {      // Spqlios_ifft (execute_reverse, fft_processor_spqlios.cpp:16-55): doubles in
            const double* src = reinterpret_cast<const double*>(a.src) + (size_t)g * N;
#pragma unroll
            for (int m = 0; m < R; m++) { re[m] = src[lane + 64 * m]; im[m] = src[lane + 64 * m + P]; }
        } else {
            const int32_t* src = reinterpret_cast<const int32_t*>(a.src) + (size_t)g * N;
#pragma unroll
            for (int m = 0; m < R; m++) { re[m] = (double)src[lane + 64 * m]; im[m] = (double)src[lane + 64 * m + P]; }
        }
        fft_forward<LOGN>(re, im, tw, xbuf, lane);
        if (a.dst_layout == 0) {
            double* dst = reinterpret_cast<double*>(a.dst) + (size_t)g * N;
#pragma unroll
            for (int m = 0; m < R; m++) { dst[G::pos3(lane, m)] = re[m]; dst[P + G::pos3(lane, m)] = im[m]; }
        } else {
            cplx* dst = reinterpret_cast<cplx*>(a.dst) + bk_poly_remap((size_t)g, a.rows) * P;
#pragma unroll
            for (int m = 0; m < R; m++) dst[m * 64 + lane] = make_double2(re[m], im[m]);
        }
    }
}

// Spqlios_fft_u32 (spqlios-wrapper.cpp:34-36): count FrrSeries -> polynomials of N torus words
template <int LOGN, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void k_fft_inverse(const FftArgs a) {
    typedef Geo<LOGN> G;
    constexpr int N = G::N, P = G::P, R = G::R;
    extern __shared__ __align__(16) unsigned char smem[];
    cplx* tw = reinterpret_cast<cplx*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int idx = tid; idx < G::TW_DIR; idx += 64 * WAVES) tw[idx] = a.tw[G::TW_DIR + idx];
    __syncthreads();
    double* xbuf = reinterpret_cast<double*>(tw + G::TW_DIR) + (size_t)wave * G::XSLOTS;
    for (int g = blockIdx.x * WAVES + wave; g < a.count; g += gridDim.x * WAVES) {
        const double* src = reinterpret_cast<const double*>(a.src) + (size_t)g * N;
        double re[R], im[R];   // 2/N is folded into the untwist twiddles
#pragma unroll
        for (int m = 0; m < R; m++) { re[m] = src[G::pos3(lane, m)]; im[m] = src[P + G::pos3(lane, m)]; }
        fft_inverse<LOGN>(re, im, tw, tw, xbuf, lane);
        if (a.f64_io) {      // Spqlios_fft (execute_direct, fft_processor_spqlios.cpp:108-153): doubles out, no truncation
            double* dst = reinterpret_cast<double*>(a.dst) + (size_t)g * N;
#pragma unroll
            for (int m = 0; m < R; m++) { dst[lane + 64 * m] = re[m]; dst[lane + 64 * m + P] = im[m]; }
            continue;
        }
        uint32_t* dst = reinterpret_cast<uint32_t*>(a.dst) + (size_t)g * N;
#pragma unroll
        for (int m = 0; m < R; m++) { dst[lane + 64 * m] = trunc_to_torus_wide(re[m]); dst[lane + 64 * m + P] = trunc_to_torus_wide(im[m]); }
    }
}

// FrrSeries layout <-> device BK layout, count polynomials (dir 0: FrrSeries -> device, 1: device -> FrrSeries)
template <int LOGN>
__global__ void k_bk_permute(const double* __restrict__ src, double* __restrict__ dst, size_t count, int dir, int rows) {
    typedef Geo<LOGN> G;
    constexpr int N = G::N, P = G::P;
    const size_t total = count * P;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t g = idx / P;
        const int k = (int)(idx % P);          // device index m * 64 + lane
        const int m = k >> 6, lane = k & 63;
        const int pos = G::pos3(lane, m);
        const size_t gd = bk_poly_remap(g, rows);   // g: FrrSeries-side index, gd: device-side index
        if (dir == 0) {
            dst[gd * N + 2 * k] = src[g * N + pos];
            dst[gd * N + 2 * k + 1] = src[g * N + P + pos];
        } else {
            dst[g * N + pos] = src[gd * N + 2 * k];
            dst[g * N + P + pos] = src[gd * N + 2 * k + 1];
        }
    }
}

// Spqlios_poly_mul (spqlios-wrapper.cpp:38-53): res = a (*) b mod X^N + 1 through the transform: both operands viewed as signed
// i32, forward transforms, complex pointwise product (aimbim = ai bi; arebim = ar bi; re = ar br - aimbim; im = ai br + arebim),
// inverse transform with truncation.  Every product and sum rounded on its own (the reference builds this loop with -Ofast and
// may contract it: off the gate path, parity there is +-1 LSB).
struct PolyMulArgs {
    const cplx* tw;
    const uint32_t* a;    // [count][N]
    const uint32_t* b;    // [count][N]
    uint32_t* res;        // [count][N]
    int32_t count;
};
template <int LOGN, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void k_poly_mul(const PolyMulArgs a) {
    typedef Geo<LOGN> G;
    constexpr int N = G::N, P = G::P, R = G::R;
    extern __shared__ __align__(16) unsigned char smem[];
    cplx* tw = reinterpret_cast<cplx*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int idx = tid; idx < G::TW_TOTAL; idx += 64 * WAVES) tw[idx] = a.tw[idx];
    __syncthreads();
    double* xbuf = reinterpret_cast<double*>(tw + G::TW_TOTAL) + (size_t)wave * G::XSLOTS;
    for (int g = blockIdx.x * WAVES + wave; g < a.count; g += gridDim.x * WAVES) {
        const int32_t* pa = reinterpret_cast<const int32_t*>(a.a) + (size_t)g * N;
        const int32_t* pb = reinterpret_cast<const int32_t*>(a.b) + (size_t)g * N;
        double ar[R], ai[R], br[R], bi[R];
#pragma unroll
        for (int m = 0; m < R; m++) { ar[m] = (double)pa[lane + 64 * m]; ai[m] = (double)pa[lane + 64 * m + P]; }
        fft_forward<LOGN>(ar, ai, tw, xbuf, lane);
#pragma unroll
        for (int m = 0; m < R; m++) { br[m] = (double)pb[lane + 64 * m]; bi[m] = (double)pb[lane + 64 * m + P]; }
        fft_forward<LOGN>(br, bi, tw, xbuf, lane);
#pragma unroll
        for (int m = 0; m < R; m++) {
            const double aimbim = ai[m] * bi[m], arebim = ar[m] * bi[m];
            const double t0 = ar[m] * br[m], t1 = ai[m] * br[m];
            ar[m] = t0 - aimbim;
            ai[m] = t1 + arebim;
        }
        fft_inverse<LOGN>(ar, ai, tw + G::TW_DIR, tw + G::TW_DIR, xbuf, lane);
        uint32_t* dst = a.res + (size_t)g * N;
#pragma unroll
        for (int m = 0; m < R; m++) { dst[lane + 64 * m] = trunc_to_torus_wide(ar[m]); dst[lane + 64 * m + P] = trunc_to_torus_wide(ai[m]); }
    }
}

struct ExtProdArgs {
    const cplx* tw;
    const cplx* bk;
    const int32_t* bk_index;   // [count]
    const uint32_t* trlwe;     // [count][2][N]
    uint32_t* out;             // [count][2][N]
    int32_t count;
};

template <int LOGN, int L, int BGBIT, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void k_external_product(const ExtProdArgs a) {
    typedef Geo<LOGN> G;
    constexpr int N = G::N, R = G::R;
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    cplx* tw = reinterpret_cast<cplx*>(smem);
    TwStage<LOGN>::stage(tw, a.tw, tid, 64 * WAVES);
    __syncthreads();
    const int g = blockIdx.x * WAVES + wave;
    if (g >= a.count) return;
    unsigned char* wbase = smem + (size_t)TwStage<LOGN>::LDS_CPLX * sizeof(cplx) + (size_t)wave * bootstrap_wave_lds_bytes<LOGN>(0);
    double* xbuf = reinterpret_cast<double*>(wbase);
    uint32_t* accbuf = reinterpret_cast<uint32_t*>(wbase + (size_t)G::XSLOTS * sizeof(double));
    for (int c = lane; c < 2 * N; c += 64) accbuf[c] = a.trlwe[(size_t)g * 2 * N + c];
    wave_lds_sync();
    const size_t trgsw_cplx = (size_t)2 * L * 2 * R * 64;
    cmux_step<LOGN, L, BGBIT, false>(accbuf, 0, a.bk + (size_t)a.bk_index[g] * trgsw_cplx, TwStage<LOGN>::fwd(tw), TwStage<LOGN>::inv_small(tw), TwStage<LOGN>::inv_big(tw, a.tw), xbuf, lane);
    for (int c = lane; c < 2 * N; c += 64) a.out[(size_t)g * 2 * N + c] = accbuf[c];
}

struct KeySwitchArgs {
    const uint32_t* ksk;
    const uint32_t* tlwe1;   // [count][N+1]
    uint32_t* out;           // [count][n+1]
    int32_t count, n, ksw;
};

template <int LOGN, int KS_T, int KS_BB, int KSQ, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void k_key_switch(const KeySwitchArgs a) {
    constexpr int N = 1 << LOGN;
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = blockIdx.x * WAVES + wave;
    if (g >= a.count) return;
    uint32_t* ap = reinterpret_cast<uint32_t*>(smem) + (size_t)wave * N;
    const uint32_t* src = a.tlwe1 + (size_t)g * (N + 1);
    for (int c = lane; c < N; c += 64) ap[c] = src[c];
    wave_lds_sync();
    key_switch_wave<LOGN, KS_T, KS_BB, KSQ>(ap, src[N], a.ksk, a.ksw, a.n, a.out + (size_t)g * (a.n + 1), lane);
}

}  // namespace rtfhe
