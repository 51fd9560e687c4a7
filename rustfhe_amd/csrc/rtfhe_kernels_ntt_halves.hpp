// rtfhe_kernels_ntt_halves.hpp -- the exact-integer NTT backend at N = 2048 (BASELINE config 5 on the backend north_star names).
//
// Same prime (P = 2^50 - 16383: 2^14 | P - 1, so the 4096-th roots of unity the negacyclic transform of 2048 points needs exist)
// and the same 1024-point wave transforms as N = 1024 (rtfhe_ntt.hpp), arranged as k_bootstrap_halves arranges the FFT mirror:
// the two waves of a gate split every transform by its top index bit.
//   forward : the first Cooley-Tukey stage (stride 1024, one twiddle zeta_1 = psi^1024) pairs coefficient j with j + 1024:
//             wave 0 keeps x0 + zeta_1 x1, wave 1 keeps x0 - zeta_1 x1 (both gather and decompose all 2048 coefficients: no
//             synchronisation inside a forward transform), then each runs the 1024-point transform of its half, whose block
//             twiddles are those of the 2048-point table: zeta_{k' + (1 + H) 2^floor(log2 k')} for the sub-transform's index k'
//   products: each wave owns its half of the points for all six rows and both components
//   inverse : the 1024-point Gentleman-Sande transform per half, then the last stage (stride 1024) across the halves: the waves
//             swap their results through their exchange buffers (two LDS-only barriers); wave 0 keeps u + v = coefficients
//             [0, 1024), wave 1 zeta_1^-1 (u - v) = coefficients [1024, 2048)
//   sums    : a gate's sum of 2l = 6 products reaches 2^49.58 > P/2 at N = 2048, a sum of 3 stays below 2^48.58: the three b-rows
//             and the three a-rows are accumulated and inverse-transformed SEPARATELY and added as torus words (four inverse
//             transforms per step instead of two)
// scripts/ntt/model2048.py runs this construction on exact integers (equal to the schoolbook negacyclic product on random and
// extreme inputs, every intermediate below 2^53).  Outputs are bit-identical to the oracle's exact_int backend.
//
// Tables: only the two halves' FORWARD tables exist (16 KiB of LDS): the inverse transform of half H reads the forward table of
// half 1 - H with the block index reversed (ntt::inverse_rev: zeta_{nb+b}^-1 = -zeta_{nb+(nb-1-b)}), and the last stage's
// twiddle is zeta_1^-1 = -zeta_1.  With 16 KiB of accumulator per gate four gates per CU fit (154 KiB).
#pragma once


#include "rtfhe_kernels_ntt.hpp"

namespace rtfhe {

struct NttHalvesTw {
    // doubles: [half][ntt::TW_DIR_PAD]: the forward table of the half's 1024-point transform; the pad entry (index ntt::TW_DIR)
    // holds zeta_1, the twiddle of the stage across the halves
    static constexpr int CROSS = ntt::TW_DIR;
    static constexpr int TABLE = ntt::TW_DIR_PAD;
    static constexpr int DIG = 2 * TABLE;          // [64]: (signed 6-bit digit) * zeta_1 mod P, centred (see ntt::TW_DIG)
    static constexpr int TOTAL = DIG + ntt::DIGITS;
};

struct NttHalvesLds {
    static constexpr size_t TW = (size_t)NttHalvesTw::TOTAL * sizeof(double);
    static constexpr size_t XB = (size_t)ntt::XSLOTS * sizeof(double);
    static_assert(ntt::XSLOTS >= ntt::N, "an exchange buffer must hold one half");
    __host__ __device__ static constexpr size_t abar_bytes(int npad) { return ((size_t)npad * 2 + 15) / 16 * 16; }
    static constexpr size_t FLAGS = 16;       // two arrival counters per gate (pair_sync)
    __host__ __device__ static constexpr size_t gate_bytes(int npad) { return (size_t)2 * 2048 * 4 + abar_bytes(npad) + 2 * XB + FLAGS; }
    __host__ __device__ static constexpr size_t bytes(int gates, int npad) { return TW + (size_t)gates * gate_bytes(npad); }
};

// key in the NTT domain, N = 2048: double[n][2l rows][2 comp][2 half][8][64 lanes][2]; N^-1 folded in; centred residues
__device__ __forceinline__ const double2* ntt_halves_bk_row(const double* bk_i, int row, int comp, int H, int lane) {
    return reinterpret_cast<const double2*>(bk_i + ((size_t)((row * 2 + comp) * 2 + H) * ntt::N)) + lane;
}

// the tables of one wave: fwd = forward table of its half, mir = forward table of the other half (read by the inverse transform)
struct NttHalvesTables {
    const double* fwd;
    const double* mir;
    const double* dig;
    __device__ __forceinline__ NttHalvesTables(const double* lds, int H)
        : fwd(lds + (size_t)H * NttHalvesTw::TABLE), mir(lds + (size_t)(1 - H) * NttHalvesTw::TABLE), dig(lds + NttHalvesTw::DIG) {}
};

__device__ __forceinline__ void ntt_halves_load_tables(double* lds, const double* glob, int tid, int nthreads) {
    for (int idx = tid; idx < NttHalvesTw::TOTAL; idx += nthreads) lds[idx] = glob[idx];
}

// One external product (CMUX = false: acc <- BK_i (x) acc) or one CMUX step (acc += BK_i (x) ((X^r - 1) acc)) by the two waves of a
// gate.  Every wave of the workgroup must call it (workgroup barriers inside).
// PAIRSYNC: the two halves synchronise with each other only (pair_sync on the gate's arrival counters: my_flag / partner_flag are their LDS
// addresses, sync_k the running count) instead of the workgroup barrier.
template <int L, int BGBIT, bool CMUX, bool PAIRSYNC = false>
__device__ __forceinline__ void ntt_halves_step(uint32_t* __restrict__ accbuf, int r, const __amdgpu_buffer_rsrc_t bk_rsrc, int bk_off,
                                                const NttHalvesTables& t, double* myx, const double* otx,      /* the partner writes otx: no restrict */
                                                int lane0, int H, unsigned my_flag = 0, unsigned partner_flag = 0, unsigned* sync_k = nullptr) {
    auto halves_sync = [&]() {
        if constexpr (PAIRSYNC) pair_sync(my_flag, partner_flag, ++*sync_k);
        else lds_barrier();
    };
    constexpr int LOGN = 11, N = 2048, HN = 1024, R = ntt::R;
    constexpr uint32_t M = decomp_mask(L, BGBIT);
    static_assert((1 << BGBIT) == ntt::DIGITS, "the digit table has one entry per digit value");
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    auto ld2 = [&](int voff, int soff) {
        const v4u v = __builtin_amdgcn_raw_buffer_load_b128(bk_rsrc, voff, soff, 0);
        return make_double2(__longlong_as_double(((unsigned long long)v.y << 32) | v.x), __longlong_as_double(((unsigned long long)v.w << 32) | v.z));
    };
    uint32_t u0[R], u1[R];
    auto gather = [&](const uint32_t* poly, int lane) {
#pragma unroll
        for (int m = 0; m < R; m++) {
            const int c = lane + 64 * m;
            const uint32_t d0 = CMUX ? (rotated_coef<LOGN>(poly, c, r) - poly[c]) : poly[c];
            const uint32_t d1 = CMUX ? (rotated_coef<LOGN>(poly, c + HN, r) - poly[c + HN]) : poly[c + HN];
            u0[m] = (d0 + M) ^ M;
            u1[m] = (d1 + M) ^ M;
        }
    };
    gather(accbuf, lane0);
#pragma unroll 1
    for (int h = 0; h < 2; h++) {
        int lane = lane0;
        asm volatile("" : "+v"(lane));      // addresses are re-derived per polynomial instead of living across the step
        double s0[R], s1[R];
#pragma unroll
        for (int m = 0; m < R; m++) { s0[m] = 0.0; s1[m] = 0.0; }
        const double zc = t.fwd[NttHalvesTw::CROSS];
#pragma unroll 1
        for (int jj = 0; jj < L; jj++) {
            double x[R];
            if (H) {        // branch on the half around the loop (a select inside it computes both sums)
#pragma unroll
                for (int m = 0; m < R; m++)
                    x[m] = (double)decomp_digit(u0[m], BGBIT, jj) - t.dig[ntt::digit_entry(u1[m], BGBIT, jj)];     // digit * zeta_1 mod P from the 64-entry table
            } else {
#pragma unroll
                for (int m = 0; m < R; m++)
                    x[m] = (double)decomp_digit(u0[m], BGBIT, jj) + t.dig[ntt::digit_entry(u1[m], BGBIT, jj)];
            }
            double2 b0[R / 2], b1[R / 2];
            ntt::forward_a(x, t.fwd, myx, lane);
            // key rows through a buffer resource (scalar row offset + one per-lane VGPR + immediates, see k_bootstrap_pair): component 0
            // requested under the last pass, component 1 under component 0's products (all 64 registers of key rows requested up
            // front cost 30 spilled registers: 45.0 vs 42.4 ms per 1024 gates)
            const int s0off = __builtin_amdgcn_readfirstlane(bk_off + (((h * L + jj) * 2 + 0) * 2 + H) * ntt::N * 8);
            const int s1off = s0off + 2 * ntt::N * 8;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < R / 2; q++) b0[q] = ld2(lane * 16 + q * 1024, s0off);
            __builtin_amdgcn_sched_barrier(0);
            ntt::forward_b(x, t.fwd, myx, lane);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < R / 2; q++) b1[q] = ld2(lane * 16 + q * 1024, s1off);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < R / 2; q++) { s0[2 * q] += ntt::modmul(x[2 * q], b0[q].x);     s0[2 * q + 1] += ntt::modmul(x[2 * q + 1], b0[q].y); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < R / 2; q++) { s1[2 * q] += ntt::modmul(x[2 * q], b1[q].x);     s1[2 * q + 1] += ntt::modmul(x[2 * q + 1], b1[q].y); }
        }
        // the a-polynomial's words leave the accumulator before the b-rows' results go into it (the partner's gather is ordered
        // before its own next barrier in the same way)
        if (h == 0) gather(accbuf + N, lane);
#pragma unroll 1
        for (int comp = 0; comp < 2; comp++) {
            double x[R];
#pragma unroll
            for (int m = 0; m < R; m++) x[m] = comp ? s1[m] : s0[m];
            ntt::inverse_rev(x, t.mir, myx, lane);
            // x[m] = sub-coefficient lane + 64 m of this half (u on wave 0, v on wave 1), |x| <= P/2
#pragma unroll
            for (int m = 0; m < R; m++) lds_st(&myx[lane + 64 * m], x[m]);      // cross-wave payload: relaxed atomics, as in the FFT halves kernel
            halves_sync();
            uint32_t* poly = accbuf + comp * N + H * HN;
            // the branch on the (wave-uniform) half stays outside the point loop (inside it: one branch and one LDS wait per point)
            if (H) {
#pragma unroll
                for (int m = 0; m < R; m++) x[m] = ntt::normalize(ntt::modmul(x[m] - lds_ld(&otx[lane + 64 * m]), zc));     // zeta_1^-1 (u - v) = zeta_1 (v - u)
            } else {
#pragma unroll
                for (int m = 0; m < R; m++) x[m] = ntt::normalize(x[m] + lds_ld(&otx[lane + 64 * m]));
            }
#pragma unroll
            for (int m = 0; m < R; m++) {
                const uint32_t w = ntt::to_torus(x[m]);
                if (CMUX || h == 1) poly[lane + 64 * m] += w;
                else poly[lane + 64 * m] = w;
            }
            halves_sync();      // the partner has read my buffer; both halves of the polynomial are written
        }
    }
}

struct NttHalvesArgs {
    BootstrapArgs b;          // tw and bk of `b` are unused here
    const double* ntt_tw;     // [NttHalvesTw::TOTAL]
    const double* ntt_bk;     // layout above
};

template <int L, int BGBIT, int KS_T, int KS_BB, int KSQ, int GATES>
__global__ __launch_bounds__(128 * GATES, 1) void k_bootstrap_ntt_halves(const NttHalvesArgs args) {
    constexpr int LOGN = 11, N = 2048, R = ntt::R, NT = 128 * GATES;
    const BootstrapArgs& a = args.b;
    extern __shared__ __align__(16) unsigned char smem[];
    double* tw = reinterpret_cast<double*>(smem);
    const int tid = threadIdx.x, lane0 = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slot = wave % GATES, H = wave / GATES;          // the two halves of a gate share a SIMD (waves w, w + GATES)
    ntt_halves_load_tables(tw, args.ntt_tw, tid, NT);
    const NttHalvesTables tables(tw, H);

    const int g_raw = blockIdx.x * GATES + slot;
    const int g = g_raw < a.count ? g_raw : a.count - 1;
    const GateIo io = gate_io(a, g);
    const bool live = g_raw < a.count && io.ok;      // idle / skipped gates still take part in every barrier

    unsigned char* gbase = smem + NttHalvesLds::TW + (size_t)slot * NttHalvesLds::gate_bytes(a.npad);
    uint32_t* accbuf = reinterpret_cast<uint32_t*>(gbase);                                  // [2][N]
    uint16_t* abar = reinterpret_cast<uint16_t*>(gbase + (size_t)2 * N * 4);
    double* xb0 = reinterpret_cast<double*>(gbase + (size_t)2 * N * 4 + NttHalvesLds::abar_bytes(a.npad));
    double* xb1 = xb0 + ntt::XSLOTS;
    double* myx = H ? xb1 : xb0;
    const double* otx = H ? xb0 : xb1;
    // arrival counters of the two halves of this gate (zeroed before the start-up barriers)
    constexpr bool PAIRSYNC = GATES >= 2;     // 1024 gates 31.30 -> 31.06 ms, 512 gates 19.70 -> 19.25; one gate per workgroup 18.22 -> 18.34: off there
    uint32_t* flags = reinterpret_cast<uint32_t*>(gbase + NttHalvesLds::gate_bytes(a.npad) - NttHalvesLds::FLAGS);
    if (lane0 == 0) flags[H] = 0u;
    const unsigned my_flag = (unsigned)(size_t)(__attribute__((address_space(3))) uint32_t*)(flags + H);
    const unsigned partner_flag = (unsigned)(size_t)(__attribute__((address_space(3))) uint32_t*)(flags + (1 - H));
    unsigned sync_k = 0;

    const int n = a.n;
    {   // pre-step + mod switch (tfhe.rs:41-71, 97, 107-108) to [0, 2N)
        constexpr int SH = 32 - LOGN - 1;
        for (int i = lane0 + 64 * H; i <= n; i += 128) {
            const uint32_t t = gate_linear(io.op, io.p0[i], io.p1[i], i == n);
            abar[i] = (uint16_t)((i == n) ? (t >> SH) : ((t + (1u << (SH - 1))) >> SH));
        }
    }
    __syncthreads();
    {   // acc = X^{-bbar} * testvec (tfhe.rs:85, 98-106); each wave initialises half of the words
        const int bbar = (int)abar[n];
        for (int c = lane0 + 64 * H; c < 2 * N; c += 128) {
            const int e = (c + bbar) & (2 * N - 1);
            accbuf[c] = c < N ? ((e >> LOGN) ? 0xE0000000u : 0x20000000u) : 0u;
        }
    }
    __syncthreads();

    // key rows through a buffer resource based at the current step's TRGSW (the whole key exceeds a 32-bit byte offset)
    constexpr size_t trgsw_doubles = (size_t)2 * L * 2 * N;
#pragma unroll 1
    for (int i = 0; i < a.steps; i++) {
        const int r = __builtin_amdgcn_readfirstlane((int)abar[i]);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(args.ntt_bk + (size_t)i * trgsw_doubles), 0,
                                                                            (int)(trgsw_doubles * 8), 0x00020000);
        ntt_halves_step<L, BGBIT, true, PAIRSYNC>(accbuf, r, rs, 0, tables, myx, otx, lane0, H, my_flag, partner_flag, &sync_k);
    }

    if (a.mode == MODE_BLIND_ROTATE) {
        if (live) {
            uint32_t* o = a.out + (size_t)g * 2 * N;
            for (int c = lane0 + 64 * H; c < 2 * N; c += 128) o[c] = accbuf[c];
        }
        return;
    }
    // sample extract index 0 (trlwe.rs:110-121): a'_0 = a_0, a'_k = -a_{N-k}; b' = b_0
    {
        uint32_t av[R];
#pragma unroll
        for (int m = 0; m < R; m++) av[m] = accbuf[N + lane0 + 64 * m + 1024 * H];
        __syncthreads();
#pragma unroll
        for (int m = 0; m < R; m++) {
            const int c = lane0 + 64 * m + 1024 * H;
            accbuf[N + ((N - c) & (N - 1))] = (c == 0) ? av[m] : (0u - av[m]);
        }
    }
    __syncthreads();
    if (a.mode == MODE_EXTRACT) {      // the key switch of the whole batch follows as its own launch (k_key_switch_mm)
        if (live) {
            const int ge = a.ext_first + g;      // batch-wide gate number: the sample buffer is laid out for the key switch (ext_slot)
            for (int c = H * (N / 2) + lane0; c < (H + 1) * (N / 2); c += 64) *ext_slot(a.ext, ge, c, N) = accbuf[N + c];
            if (H == 0 && lane0 == 0) *ext_slot(a.ext, ge, N, N) = accbuf[0];
            for (int c = H * 64 + lane0; c <= n; c += 128) io.out[c] = 0u;
        }
        return;
    }
    // identity key switch (tlwe.rs:43-73): each wave sums the rows of half of the coefficients
    uint4 sum[KSQ];
    ks_accumulate<LOGN, KS_T, KS_BB, KSQ>(accbuf + N, H * (N / 2), (H + 1) * (N / 2), a.ksk, a.ksw, sum, lane0);
    uint4* part = reinterpret_cast<uint4*>(xb1) + lane0;   // [KSQ][64] uint4
    if (H == 1) {
#pragma unroll
        for (int q = 0; q < KSQ; q++) part[q * 64] = sum[q];
    }
    __syncthreads();
    if (H == 0 && live) {
        const uint32_t bprime = accbuf[0];
#pragma unroll
        for (int q = 0; q < KSQ; q++) {
            const uint4 o = part[q * 64];
            const int col = 4 * (lane0 + 64 * q);
            const uint32_t s[4] = {sum[q].x + o.x, sum[q].y + o.y, sum[q].z + o.z, sum[q].w + o.w};
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (col + e <= n) io.out[col + e] = ((col + e == n) ? bprime : 0u) - s[e];
        }
    }
}

struct NttHalvesBkArgs {
    const double* ntt_tw;       // [NttHalvesTw::TOTAL]
    const uint32_t* bk_torus;   // [n][2][2l][N]
    double* ntt_bk;
    int32_t count;              // polynomials
    int32_t rows;               // 2l
    double ninv;                // N^-1 mod P, centred
};

// key rows -> NTT domain at N = 2048 (the counterpart of TRGSWRepF::from, hom_nand/src/trgsw.rs:68-76): one wave per polynomial,
// both halves one after the other
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void k_ntt_bk_halves(const NttHalvesBkArgs a) {
    constexpr int N = 2048, HN = 1024, R = ntt::R;
    extern __shared__ __align__(16) unsigned char smem[];
    double* tw = reinterpret_cast<double*>(smem);                   // forward tables of both halves, whole
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    ntt_halves_load_tables(tw, a.ntt_tw, tid, 64 * WAVES);
    __syncthreads();
    double* xbuf = tw + NttHalvesTw::TOTAL + (size_t)wave * ntt::XSLOTS;
    const double zc = tw[NttHalvesTw::CROSS];
    for (int g = blockIdx.x * WAVES + wave; g < a.count; g += gridDim.x * WAVES) {
        const int32_t* src = reinterpret_cast<const int32_t*>(a.bk_torus) + (size_t)g * N;
        double* dst_poly = a.ntt_bk + bk_poly_remap((size_t)g, a.rows) * N;
#pragma unroll 1
        for (int H = 0; H < 2; H++) {
            double x[R];
#pragma unroll
            for (int m = 0; m < R; m++) {
                const double x0 = (double)src[lane + 64 * m];
                const double tt = ntt::normalize(ntt::modmul((double)src[HN + lane + 64 * m], zc));
                x[m] = H ? x0 - tt : x0 + tt;
            }
            ntt::forward(x, tw + (size_t)H * NttHalvesTw::TABLE, xbuf, lane);
            double2* dst = reinterpret_cast<double2*>(dst_poly + (size_t)H * HN) + lane;
#pragma unroll
            for (int q = 0; q < R / 2; q++)
                dst[q * 64] = make_double2(ntt::normalize(ntt::modmul(x[2 * q], a.ninv)), ntt::normalize(ntt::modmul(x[2 * q + 1], a.ninv)));
        }
    }
}

struct NttHalvesExtProdArgs {
    const double* ntt_tw;
    const double* ntt_bk;
    const int32_t* bk_index;
    const uint32_t* trlwe;
    uint32_t* out;
    int32_t count;
};

// stage-level external product (test surface): one gate per workgroup of two waves
template <int L, int BGBIT>
__global__ __launch_bounds__(128, 1) void k_external_product_ntt_halves(const NttHalvesExtProdArgs a) {
    constexpr int N = 2048;
    extern __shared__ __align__(16) unsigned char smem[];
    double* tw = reinterpret_cast<double*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int H = __builtin_amdgcn_readfirstlane(tid >> 6);
    ntt_halves_load_tables(tw, a.ntt_tw, tid, 128);
    const NttHalvesTables tables(tw, H);
    uint32_t* accbuf = reinterpret_cast<uint32_t*>(smem + NttHalvesLds::TW);
    double* xb0 = reinterpret_cast<double*>(smem + NttHalvesLds::TW + (size_t)2 * N * 4);
    double* xb1 = xb0 + ntt::XSLOTS;
    const int g = blockIdx.x;                       // grid = count
    for (int c = tid; c < 2 * N; c += 128) accbuf[c] = a.trlwe[(size_t)g * 2 * N + c];
    __syncthreads();
    constexpr size_t trgsw_doubles = (size_t)2 * L * 2 * N;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(a.ntt_bk + (size_t)a.bk_index[g] * trgsw_doubles), 0,
                                                                        (int)(trgsw_doubles * 8), 0x00020000);
    ntt_halves_step<L, BGBIT, false>(accbuf, 0, rs, 0, tables, H ? xb1 : xb0, H ? xb0 : xb1, lane, H);
    __syncthreads();
    for (int c = tid; c < 2 * N; c += 128) a.out[(size_t)g * 2 * N + c] = accbuf[c];
}

}  // namespace rtfhe
