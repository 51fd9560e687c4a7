// rtfhe_kernels_halves.hpp -- the bootstrap kernel for N = 2048 with TWO WAVES PER TRANSFORM (BASELINE config 5).
//
// At N = 2048 a polynomial's transform has 1024 complex points: 16 per lane in one wavefront.  The one-wave-per-gate kernel
// (k_bootstrap<11>) then needs 128 VGPRs of accumulators + 64 of spectrum + 128 of key rows, lives in 512 registers with
// ~790 AGPR shuffles per step, and runs one wave per SIMD at the lone-wave FP64 issue rate (6.75 cycles/instruction).  A
// split by digit rows as in k_bootstrap_pair does not fit either (three 1024-point spectra = 192 VGPRs).
//
// Here the two waves of a gate split every transform by its TOP index bit instead.  The reference's forward network is
// decimation in frequency (spqlios-fft-impl.cpp:469-641): after the twist, its first stage (halfnn = P/2 = 512) pairs point
// q with point q + 512; every later stage stays inside one half.  So
//   wave A (half 0): twist both inputs, y[q] = x0 + x1            -> 512-point sub-transform -> spectrum points [0, 512)
//   wave B (half 1): twist both inputs, y[q] = (x0 - x1) * w_q    -> 512-point sub-transform -> spectrum points [512, 1024)
// with the sub-transform exactly the 512-point network of the N = 1024 kernels (8 points per lane, three in-register passes,
// two wave-private LDS exchanges; stage twiddles depend on 2 halfnn only).  Both waves gather and twist all 16 inputs a lane
// needs (48 extra FP64 instructions per row: the price of never synchronising inside a forward transform).  Each wave then
// owns ITS HALF of the points for the multiply-accumulate over all six rows and both components: the fold order
// (trgsw.rs:290-299) holds trivially, no partial sums travel.  The inverse network is decimation in time
// (spqlios-fft-impl.cpp:204-397): each wave runs the sub-network (halfnn = 1 .. 256) on its half, then the last stage
// (halfnn = 512: t = x1 * w, x0 + t, x0 - t) needs the other half: wave B sends t, wave A sends x0 through their exchange
// buffers, two LDS-only barriers per component.  Untwist, truncate, += into the LDS accumulator as everywhere else.
//
// Registers: three rows of one polynomial side by side (96 VGPRs of half-spectra) + 64 of accumulators + a two-buffer key-row
// ring (64): 248, two waves per SIMD, four gates per CU.  LDS: 16 KiB of accumulator per gate leave room for the forward
// sub-transform's stage tables (16 KiB) and the inverse's small ones (1 KiB) only: the twist / untwist tables and the inverse's
// first-stage and pass-1 tables are read from global memory (once per polynomial / once per inverse: 2 inverses per step vs
// 6 forward rows).  (One table cannot serve both directions: the reference's inverse table is the conjugate of the forward one
// EXCEPT at the quarter-turn entry of every stage, where cos comes out as -6.1e-17 forward and +6.1e-17 inverse.)
// The 2/N input scaling of fft_processor_spqlios.cpp:158 is folded into the untwist table as in the other kernels.
#pragma once

#include "rtfhe_kernels_pair.hpp"

#ifndef HALVES_PRIO_B
#define HALVES_PRIO_B 1      // wave B carries ~9 % more arithmetic (the first-stage / last-stage twiddle products)
#endif

#ifndef HALVES_B_LOW_MASK
#define HALVES_B_LOW_MASK 0
#endif
// Round 4: buffer OWNERSHIP ping-pongs between the two halves (see "ping-pong" in the kernel): every trade needs ONE synchronisation instead of two.
// -DHALVES_PINGPONG=0: round 3's write / sync / read / sync protocol (A/B builds).
#ifndef HALVES_PINGPONG
#define HALVES_PINGPONG 1
#endif
// with the ping-pong protocol: the next row's own twist products are computed between a row's arrival flag and the wait for the partner's
#ifndef HALVES_EARLY_ROW
#define HALVES_EARLY_ROW 1
#endif
#ifndef HALVES_STAIRS
#define HALVES_STAIRS 2     // 2: pass 1 at priority 3, pass 2 at 2, pass 3 at 1, multiply-accumulate at 0 (16.13 -> 16.06 ms per 1024 gates, 14.12 -> 13.98 per 768); 1: coarser; 0: off
#endif
#ifndef HALVES_OPAQUE_WAIT
#define HALVES_OPAQUE_WAIT 1
#endif
#ifndef HALVES_SPLIT_MIN
#define HALVES_SPLIT_MIN 3   // gates per workgroup from which the two halves trade the first stage's inputs instead of both computing all of them
#endif

namespace rtfhe {

// twiddle table of the halves kernel, cplx units: forward part, then inverse part
struct HalvesTw {
    static constexpr int TWIST = 0;                 // [16][64]: k < 8: point lane + 64 k; k >= 8: point 512 + lane + 64 (k - 8)   (global memory)
    static constexpr int ST1 = TWIST + 16 * 64;     // [8][64]:  first-stage twiddle (halfnn = 512) of pair q = lane + 64 m          (LDS from here ...
    static constexpr int P1 = ST1 + 8 * 64;         // [7][64]   sub-transform, as Geo<10>
    static constexpr int P2 = P1 + 7 * 64;          // [7][8]
    static constexpr int P3 = P2 + 7 * 8;           // [4]                                                                            ... to here)
    static constexpr int IUNTW = P3 + 4;            // [16][64]  untwist, times 2/N                                                  (global memory)
    static constexpr int IST1 = IUNTW + 16 * 64;    // [8][64]   last-stage twiddle (halfnn = 512)                                   (global memory)
    static constexpr int IP1 = IST1 + 8 * 64;       // [7][64]                                                                        (global memory)
    static constexpr int IP2 = IP1 + 7 * 64;        // [7][8]                                                                         (LDS from here ...
    static constexpr int IP3 = IP2 + 7 * 8;         // [4]                                                                            ... to here)
    static constexpr int TOTAL = IP3 + 4;
    static constexpr int LDS_FWD = IUNTW - ST1;     // forward ST1 .. P3
    static constexpr int LDS_INV = TOTAL - IP2;     // inverse IP2 .. IP3
    static constexpr int LDS_CPLX = LDS_FWD + LDS_INV;
};

struct HalvesLds {
    typedef Geo<10> G;   // geometry of the 512-point sub-transform
    static constexpr size_t TW = (size_t)HalvesTw::LDS_CPLX * sizeof(cplx);
    static constexpr size_t XB = (size_t)2 * G::XSLOTS * sizeof(double);            // one wave's re + im exchange buffers: hold 512 cplx
    __host__ __device__ static constexpr size_t abar_bytes(int npad) { return ((size_t)npad * 2 + 15) / 16 * 16; }
    static constexpr size_t FLAGS = 16;       // two arrival counters per gate (pair_sync)
    __host__ __device__ static constexpr size_t gate_bytes(int npad) { return (size_t)2 * 2048 * 4 + abar_bytes(npad) + 2 * XB + FLAGS; }
    __host__ __device__ static constexpr size_t bytes(int gates, int npad) { return TW + (size_t)gates * gate_bytes(npad); }
};

struct HalvesArgs {
    BootstrapArgs b;       // b.tw / b.bk unused
    const cplx* htw;       // [HalvesTw::TOTAL]
    const cplx* hbk;       // [n][2l rows][2 comp][2 half][8][64]
};

// key spectra: device layout of k_bootstrap<11> ([n][row][comp][16][64]: lane v, register m <-> point (v << 4) | m) ->
// halves layout (lane v, register m of half H <-> point (H << 9) | (v << 3) | m)
__global__ __launch_bounds__(256) void k_bk_to_halves(const cplx* __restrict__ src, cplx* __restrict__ dst, size_t polys, double scale) {
    const size_t total = polys * 1024;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t g = idx >> 10;
        const int k = (int)(idx & 1023);                 // destination: (H, m, lane)
        const int H = k >> 9, m = (k >> 6) & 7, lane = k & 63;
        const int p = (H << 9) | (lane << 3) | m;
        const cplx v = src[g * 1024 + (size_t)(p & 15) * 64 + (p >> 4)];
        dst[idx] = make_double2(v.x * scale, v.y * scale);
    }
}

template <int L, int BGBIT, int KS_T, int KS_BB, int KSQ, int GATES>
__global__ __launch_bounds__(128 * GATES, 1) void k_bootstrap_halves(const HalvesArgs ha) {
    constexpr int LOGN = 11, N = 2048, P = 1024, R = 8, NT = 128 * GATES;
    typedef Geo<10> G;   // the 512-point sub-transform
#ifdef HALVES_DUP_STAGE1      // A/B: both halves compute the whole first stage (round 2)
    constexpr bool SPLIT1 = false;
#else
    constexpr bool SPLIT1 = GATES >= HALVES_SPLIT_MIN;      // each half twists only its own inputs and the halves trade them row by row (see below)
#endif
    // half 1's last-stage twiddles of the inverse (global memory) requested one pass ahead of their use: 11.86 -> 11.72 ms per 512 gates; with the split
    // first stage the same request costs 36 spilled registers (17.6 vs 16.65 ms per 1024 gates), so only without it
#ifdef HALVES_IST_LATE
    constexpr bool IST_EARLY = false;
#else
    constexpr bool IST_EARLY = !SPLIT1;
#endif
    constexpr uint32_t M = decomp_mask(L, BGBIT);
    static_assert(L == 3, "three digit rows of a polynomial are transformed side by side");
    const BootstrapArgs& a = ha.b;
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane0 = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifndef HALVES_MAP
#define HALVES_MAP 0
#endif
    // 0: the two halves of a gate share a SIMD (waves w, w + GATES); 1 / 2 (A/B, 4 gates): as PAIR_MAP in k_bootstrap_pair
    const int slot = (HALVES_MAP == 1 && GATES == 4) ? wave / 2 : (HALVES_MAP == 2 && GATES == 4) ? 2 * ((wave & 3) >> 1) + (wave >> 2) : wave % GATES;
    const int H = (HALVES_MAP == 1 && GATES == 4) ? wave % 2 : (HALVES_MAP == 2 && GATES == 4) ? ((wave & 1) ^ (wave >> 2)) : wave / GATES;
    cplx* tw = reinterpret_cast<cplx*>(smem);
    for (int idx = tid; idx < HalvesTw::LDS_FWD; idx += NT) tw[idx] = ha.htw[HalvesTw::ST1 + idx];
    for (int idx = tid; idx < HalvesTw::LDS_INV; idx += NT) tw[HalvesTw::LDS_FWD + idx] = ha.htw[HalvesTw::IP2 + idx];
    const cplx* tw_st1 = tw;
    const cplx* tw_sub = tw + (HalvesTw::P1 - HalvesTw::ST1) - G::TW_P1;   // so that tw_sub + G::TW_P1/P2/P3 address the sub-transform tables
    const cplx* twi_small = tw + HalvesTw::LDS_FWD - G::TW_P2;            // twi_small + G::TW_P2/P3: the inverse's pass-2/3 tables
    static_assert(HalvesTw::P2 - HalvesTw::P1 == G::TW_P2 - G::TW_P1 && HalvesTw::P3 - HalvesTw::P2 == G::TW_P3 - G::TW_P2 &&
                  HalvesTw::IP3 - HalvesTw::IP2 == G::TW_P3 - G::TW_P2, "same table geometry as Geo<10>");
#ifdef HALVES_ABL_LDSTW         // timing ablation only (wrong results): the tables that live in global memory are read from LDS addresses instead
    const cplx* gtwist0 = tw; const cplx* guntw0 = tw; const cplx* gist10 = tw; const cplx* gip10 = tw;
#else
    const cplx* gtwist0 = ha.htw + HalvesTw::TWIST;           // [16][64] in global memory
    const cplx* guntw0 = ha.htw + HalvesTw::IUNTW;            // [16][64]
    const cplx* gist10 = ha.htw + HalvesTw::IST1;             // [8][64]
    const cplx* gip10 = ha.htw + HalvesTw::IP1;               // [7][64]
#endif

    const int g_raw = blockIdx.x * GATES + slot;
    const int g = g_raw < a.count ? g_raw : a.count - 1;
    const GateIo io = gate_io(a, g);
    const bool live = g_raw < a.count && io.ok;

    unsigned char* gbase = smem + HalvesLds::TW + (size_t)slot * HalvesLds::gate_bytes(a.npad);
    uint32_t* accbuf = reinterpret_cast<uint32_t*>(gbase);                                  // [2][N]
    uint16_t* abar = reinterpret_cast<uint16_t*>(gbase + (size_t)2 * N * 4);
    double* xb0 = reinterpret_cast<double*>(gbase + (size_t)2 * N * 4 + HalvesLds::abar_bytes(a.npad));
    double* xb1 = xb0 + 2 * G::XSLOTS;
    double* myx = H ? xb1 : xb0;
    double* otx = H ? xb0 : xb1;
    // arrival counters of the two halves of this gate (zeroed before the start-up barrier): the halves synchronise with each other only
    uint32_t* flags = reinterpret_cast<uint32_t*>(gbase + HalvesLds::gate_bytes(a.npad) - HalvesLds::FLAGS);
    if (lane0 == 0) flags[H] = 0u;
    [[maybe_unused]] const unsigned my_flag = (unsigned)(size_t)(__attribute__((address_space(3))) uint32_t*)(flags + H);
    [[maybe_unused]] const unsigned partner_flag = (unsigned)(size_t)(__attribute__((address_space(3))) uint32_t*)(flags + (1 - H));
    [[maybe_unused]] unsigned sync_k = 0;
#ifdef HALVES_WG_BARRIER      // A/B: the workgroup-wide barrier of round 2
#define HALVES_SYNC() lds_barrier()
#else
#define HALVES_SYNC() pair_sync(my_flag, partner_flag, ++sync_k)
#endif
#define HALVES_ARRIVE() pair_arrive(my_flag, ++sync_k)
    // The priority staircase of k_bootstrap_eo (round 4): a wave's priority falls along every stretch between two trades and is back at 3 after
    // every wait, so that whichever half is BEHIND is favoured (round 3: half B at a fixed higher priority -- it raced to every trade and idled
    // there a quarter of the step).  With the opaque wait: 1024 gates 15.66 -> 15.52 ms, 768 gates 14.94 -> 14.05 ms (profiles/r04/
    // n2048_priority_staircase_ab.log).  -DHALVES_STAIRS=0 -DHALVES_OPAQUE_WAIT=0: round 3's form.
#if HALVES_STAIRS
#define HV_PRIO(k) __builtin_amdgcn_s_setprio(k)
#else
#define HV_PRIO(k) do { } while (0)
#endif
#if HALVES_OPAQUE_WAIT && HALVES_STAIRS && defined(HALVES_FUSED_RAISE)      // A/B: the raise to 3 inside the wait's own assembly statement (as k_bootstrap_eo does
                                                                            // at 3-4 gates per workgroup); here 15.69 -> 15.88 ms per 1024 gates, 13.98 -> 13.79 per 768
#define HALVES_WAIT() pair_wait_opaque_prio3(partner_flag, sync_k)
#elif HALVES_OPAQUE_WAIT      // the spin loop inside one inline-assembly statement (no loop header in the compiler's control-flow graph)
#define HALVES_WAIT() do { pair_wait_opaque(partner_flag, sync_k); HV_PRIO(3); } while (0)
#else
#define HALVES_WAIT() do { pair_wait(partner_flag, sync_k); HV_PRIO(3); } while (0)
#endif
#ifdef HALVES_ABL_NOSYNC       // timing ablation only (racy, wrong results)
#undef HALVES_SYNC
#define HALVES_SYNC() wave_lds_sync()
#undef HALVES_ARRIVE
#undef HALVES_WAIT
#define HALVES_ARRIVE() wave_lds_sync()
#define HALVES_WAIT() wave_lds_sync()
#endif
    constexpr bool PINGPONG = HALVES_PINGPONG != 0;
    // Ping-pong ownership of the two exchange buffer pairs of a gate (round 4).  Round 3 traded a row as: write MY buffer, sync, read the PARTNER's, sync (the
    // partner has read mine: it is free again) -- sixteen synchronisations per step, each a lock-step LDS round trip in which neither wave of the SIMD issues
    // arithmetic.  Instead, after a trade each wave OWNS THE BUFFER IT HAS JUST READ: its next writes (the next row, or its sub-transform's wave-private
    // exchanges) go there.  That buffer's previous writer is done with it (its writes precede its arrival flag), its reader is this wave itself (DS instructions
    // of a wave execute in order), and nobody else touches it -- so the "buffer free" synchronisation disappears: ONE arrival / wait per trade, 8 per step.
    // The accumulator updates at the end of a step are published by the arrivals that follow them (the other component's trade, the next step's first row).
    double* wbuf = myx;    // the buffer pair this wave owns (writes next)
    double* rbuf = otx;    // the partner's (read after its arrival)

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

    // key rows in consumption order rc = 0..11: (row rc / 2, component rc & 1) of this half.  Two buffers: the first two rows of a
    // polynomial are requested once its decomposition words are dead (after the first stage), the others as a buffer retires
    const size_t trgsw_cplx = (size_t)2 * L * 2 * 2 * R * 64;
    cplx bA[R], bB[R];
    // read through a buffer resource: scalar row offset + one per-lane VGPR + immediates (see k_bootstrap_pair)
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t bk_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<cplx*>(ha.hbk), 0, 0x7fffffff, 0x00020000);
    const int lane16 = lane0 * 16;
    auto fetch = [&](cplx (&dst)[R], int step, int rc) {
#ifdef HALVES_ABL_FETCH0       // timing ablation only (wrong results): every step reads step 0's key rows (cache-hot)
        step = 0;
#endif
        const size_t row = (size_t)step * trgsw_cplx + (size_t)((rc >> 1) * 2 + (rc & 1)) * 2 * R * 64 + (size_t)H * R * 64;
        const int s_lo = __builtin_amdgcn_readfirstlane((int)(row * sizeof(cplx)));
        const int s_hi = s_lo + (R / 2) * 64 * (int)sizeof(cplx);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < R / 2; m++) {
            const v4u v = __builtin_amdgcn_raw_buffer_load_b128(bk_rsrc, lane16 + m * 1024, s_lo, 0);
            dst[m] = make_double2(__longlong_as_double(((unsigned long long)v.y << 32) | v.x), __longlong_as_double(((unsigned long long)v.w << 32) | v.z));
        }
#pragma unroll
        for (int m = 0; m < R / 2; m++) {
            const v4u v = __builtin_amdgcn_raw_buffer_load_b128(bk_rsrc, lane16 + m * 1024, s_hi, 0);
            dst[R / 2 + m] = make_double2(__longlong_as_double(((unsigned long long)v.y << 32) | v.x), __longlong_as_double(((unsigned long long)v.w << 32) | v.z));
        }
        __builtin_amdgcn_sched_barrier(0);
    };
#if !HALVES_STAIRS
    if (H) __builtin_amdgcn_s_setprio(HALVES_PRIO_B);
#endif
    // Priority schedule of wave B: after point p it runs at priority 0 if bit p of HALVES_B_LOW_MASK is set, at HALVES_PRIO_B otherwise (wave A stays at 0; at
    // equal priority the SIMD favours the older wave, A).  Points: 0 start of a polynomial, 1 after the first stage, 2 after passes 1-2 of the sub-transforms,
    // 3 after pass 3, 4 after the multiply-accumulate, 5 after an inverse sub-network, 6 after the last stage + accumulator update
    auto prio_point = [&](int point) {
        if (HALVES_B_LOW_MASK == 0) return;
        if ((HALVES_B_LOW_MASK >> point) & 1) asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 1f\n\ts_setprio 0\n1:" ::"s"(H) : "scc");
        else asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 1f\n\ts_setprio %1\n1:" ::"s"(H), "n"(HALVES_PRIO_B) : "scc");
    };
#ifdef RTFHE_WG_STAMPS     // diagnostic builds (scripts/ubench/halves_stamps.py): s_memtime ticks per phase, summed over the steps
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
#define HV_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tsum[k] += t_ - tprev; tprev = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define HV_STAMP(k) do { } while (0)
#endif

#pragma unroll 1
    for (int i = 0; i < a.steps; i++) {
        const int r = __builtin_amdgcn_readfirstlane((int)abar[i]);
        double s0re[R], s0im[R], s1re[R], s1im[R];
#pragma unroll
        for (int m = 0; m < R; m++) { s0re[m] = 0.0; s0im[m] = 0.0; s1re[m] = 0.0; s1im[m] = 0.0; }

#pragma unroll 1
        for (int h = 0; h < 2; h++) {
            const uint32_t* poly = accbuf + h * N;
            int ln = lane0;
            asm volatile("" : "+v"(ln));        // keeps the lane-derived LDS addresses from being hoisted out of the loops and spilled
            // the 16 complex inputs of this lane are points q and q + 512, q = lane + 64 m: coefficients q + 512 k, k = 0..3
            // (k = 0: Re x0, 1: Re x1, 2: Im x0, 3: Im x1)   (rotate: math.rs:85-132; decomposition: math.rs:300-326).
            // Gather, decomposition, twist (spqlios-fft-impl.cpp:512-517) and first stage (:546-569) of the three digit rows go
            // point by point: only one point's decomposition words and twiddles are live at a time.
            double yr[L][R], yi[L][R];
            const cplx* gtwist = gtwist0 + ln;
            if constexpr (SPLIT1) {
            // Round 3 (the round-2 review's proposal): each half gathers, decomposes and twists only ITS OWN inputs -- wave H owns x_H = point q + 512 H (coefficients q + 512 H and
            // q + 512 (2 + H)) -- and the two waves trade the twisted values row by row through their exchange buffers (one row = 16 doubles per
            // lane = one buffer pair): write, LDS barrier, read the partner's, LDS barrier (the partner has read mine).  Wave A then forms
            // x0 + x1, wave B (x0 - x1) w -- the same operands in the same order as before, so the same bits -- at half the gathers, digit
            // conversions and twist products per wave (round 2 computed both x0 and x1 in both waves: 384 of a wave's ~2,250 FP64-rate
            // instructions per polynomial).  Six more barriers per polynomial; the next row's own products are computed between a row's read and
            // the barrier that frees the buffer, so the partner's reads have them to land under.  All three rows at once would need 24 KiB per wave;
            // a buffer pair holds 9.2 KiB and the workgroup's LDS is full, hence row by row.  MEASURED (profiles/r03/n2048_split_first_stage_ab.log,
            // identical outputs): behind the workgroup-wide barrier the twelve extra synchronisations per step cost more than the 17 % of the
            // arithmetic they save (17.39 vs 17.08 ms per 1024 gates); with the two halves of a gate synchronising with each other only (pair_sync:
            // arrival counters in LDS) it wins where the SIMDs are full -- 4 gates per workgroup 16.87 vs 17.07 ms (60.7 k gates/s), 3 gates 16.26 vs
            // 16.30 -- and loses where they are not (2 gates 13.04 vs 12.24, 1-2 gates 12.90 vs 12.13): on for GATES >= 3.
            HV_STAMP(6);
            prio_point(0);
            cplx tH[R];        // twist factors of this half's points, from global memory: requested before the gather they land under
#pragma unroll
            for (int m = 0; m < R; m++) tH[m] = gtwist[(8 * H + m) * 64];
            uint32_t ure[R], uim[R];
#ifndef HALVES_GATHER1    // round 4: the rotated gather on byte offsets -- one add per coefficient from a per-polynomial base, mask, sign by xor / add: 9 instead of
                          // 11 integer instructions per coefficient; 16.01 -> 15.82 ms per 1024 gates, 15.11 -> 14.91 per 768 (profiles/r04/n2048_variants_ab.log).
                          // -DHALVES_GATHER1: rotated_coef() as in the other kernels
            {
                const int e0 = (ln + 512 * H - r) * 4;                   // 4 (c - r) for m = 0; the coefficients of a lane are 256 m and 4096 bytes apart
                const unsigned char* pb = reinterpret_cast<const unsigned char*>(poly);
#pragma unroll
                for (int m = 0; m < R; m++) {
                    const int c0 = ln + 64 * m + 512 * H, c1 = c0 + 1024;
                    const int t0 = e0 + 256 * m, t1 = t0 + 4096;
                    const uint32_t v0 = *reinterpret_cast<const uint32_t*>(pb + (t0 & (4 * N - 4)));
                    const uint32_t v1 = *reinterpret_cast<const uint32_t*>(pb + (t1 & (4 * N - 4)));
                    const uint32_t s0 = (uint32_t)((int32_t)((uint32_t)t0 << (31 - LOGN - 2)) >> 31);     // all ones iff bit LOGN of (c - r) is set
                    const uint32_t s1 = (uint32_t)((int32_t)((uint32_t)t1 << (31 - LOGN - 2)) >> 31);
                    ure[m] = ((((v0 ^ s0) - s0) - poly[c0]) + M) ^ M;
                    uim[m] = ((((v1 ^ s1) - s1) - poly[c1]) + M) ^ M;
                }
            }
#else
#pragma unroll
            for (int m = 0; m < R; m++) {
                const int c0 = ln + 64 * m + 512 * H, c1 = c0 + 1024;
                ure[m] = ((rotated_coef<LOGN>(poly, c0, r) - poly[c0]) + M) ^ M;
                uim[m] = ((rotated_coef<LOGN>(poly, c1, r) - poly[c1]) + M) ^ M;
            }
#endif
            HV_STAMP(0);
            if constexpr (PINGPONG) {
            // one arrival / wait per row; the buffers swap owners after every row (see "ping-pong" above).  The next row's own twist products are computed
            // between this row's arrival and the wait for the partner's (they need nothing from the partner; the row's sums then overwrite its own values,
            // so the second register set costs nothing at the point where the register file is fullest: the third row).
            double x[2][2][R];     // [row parity][re / im][point]
            auto own_row2 = [&](int jj) {
#pragma unroll
                for (int m = 0; m < R; m++) {
                    const double a0 = (double)decomp_digit(ure[m], BGBIT, jj), b0 = (double)decomp_digit(uim[m], BGBIT, jj);
                    const double rc = a0 * tH[m].x, ic = b0 * tH[m].x, rs = a0 * tH[m].y, is = b0 * tH[m].y;
                    x[jj & 1][0][m] = rc - is; x[jj & 1][1][m] = ic + rs;
                }
            };
            own_row2(0);
#pragma unroll
            for (int jj = 0; jj < L; jj++) {
                double* wb = (jj & 1) ? rbuf : wbuf;    // row 0: my buffer; row 1: the one I read in row 0; row 2: the one I read in row 1
                double* rb = (jj & 1) ? wbuf : rbuf;
                const double (&xr)[R] = x[jj & 1][0];
                const double (&xi)[R] = x[jj & 1][1];
#pragma unroll
                for (int m = 0; m < R; m++) { lds_st(&wb[ln + 64 * m], xr[m]); lds_st(&wb[G::XSLOTS + ln + 64 * m], xi[m]); }
                HALVES_ARRIVE();
                if (HALVES_EARLY_ROW && jj + 1 < L) own_row2(jj + 1);
                HALVES_WAIT();
                if (H == 0) {                           // mine = x0, partner's = x1
#pragma unroll
                    for (int m = 0; m < R; m++) {
                        yr[jj][m] = xr[m] + lds_ld(&rb[ln + 64 * m]);
                        yi[jj][m] = xi[m] + lds_ld(&rb[G::XSLOTS + ln + 64 * m]);
                    }
                } else {                                // mine = x1, partner's = x0
#pragma unroll
                    for (int m = 0; m < R; m++) {
                        const cplx w1 = tw_st1[m * 64 + ln];
                        const double dr = lds_ld(&rb[ln + 64 * m]) - xr[m], di = lds_ld(&rb[G::XSLOTS + ln + 64 * m]) - xi[m];
                        double p = dr * w1.x, q = di * w1.y;
                        yr[jj][m] = p - q;
                        p = dr * w1.y; q = di * w1.x;
                        yi[jj][m] = p + q;
                    }
                }
                if (!HALVES_EARLY_ROW && jj + 1 < L) own_row2(jj + 1);
            }
            { double* t = wbuf; wbuf = rbuf; rbuf = t; }        // three rows: I now own the buffer I read last (row 2 read rbuf)
            } else {
            double xr[R], xi[R];
            auto own_row = [&](int jj) {
#pragma unroll
                for (int m = 0; m < R; m++) {
                    const double a0 = (double)decomp_digit(ure[m], BGBIT, jj), b0 = (double)decomp_digit(uim[m], BGBIT, jj);
                    const double rc = a0 * tH[m].x, ic = b0 * tH[m].x, rs = a0 * tH[m].y, is = b0 * tH[m].y;
                    xr[m] = rc - is; xi[m] = ic + rs;
                }
            };
            own_row(0);
#pragma unroll
            for (int jj = 0; jj < L; jj++) {
#pragma unroll
                for (int m = 0; m < R; m++) { lds_st(&myx[ln + 64 * m], xr[m]); lds_st(&myx[G::XSLOTS + ln + 64 * m], xi[m]); }
                HALVES_SYNC();
                // the branch on the (wave-uniform) half stays OUTSIDE the point loop: inside it the compiler emits one branch and one LDS wait per
                // point (48 per step) and nothing is scheduled across them
                if (H == 0) {                           // mine = x0, partner's = x1
#pragma unroll
                    for (int m = 0; m < R; m++) {
                        yr[jj][m] = xr[m] + lds_ld(&otx[ln + 64 * m]);
                        yi[jj][m] = xi[m] + lds_ld(&otx[G::XSLOTS + ln + 64 * m]);
                    }
                } else {                                // mine = x1, partner's = x0
#pragma unroll
                    for (int m = 0; m < R; m++) {
                        const cplx w1 = tw_st1[m * 64 + ln];
                        const double dr = lds_ld(&otx[ln + 64 * m]) - xr[m], di = lds_ld(&otx[G::XSLOTS + ln + 64 * m]) - xi[m];
                        double p = dr * w1.x, q = di * w1.y;
                        yr[jj][m] = p - q;
                        p = dr * w1.y; q = di * w1.x;
                        yi[jj][m] = p + q;
                    }
                }
                if (jj + 1 < L) own_row(jj + 1);
                HALVES_SYNC();                          // both waves have read: the buffers are free (next row / the sub-transforms' exchanges)
            }
            }
            } else {
#pragma unroll
            for (int m = 0; m < R; m++) {
                uint32_t u[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int c = ln + 64 * m + 512 * k;
                    u[k] = ((rotated_coef<LOGN>(poly, c, r) - poly[c]) + M) ^ M;
                }
                const cplx t0 = gtwist[m * 64], t1 = gtwist[(8 + m) * 64], w1 = tw_st1[m * 64 + ln];
                double x0r[L], x0i[L], x1r[L], x1i[L];
#pragma unroll
                for (int jj = 0; jj < L; jj++) {
                    const double a0 = (double)decomp_digit(u[0], BGBIT, jj), b0 = (double)decomp_digit(u[2], BGBIT, jj);
                    const double a1 = (double)decomp_digit(u[1], BGBIT, jj), b1 = (double)decomp_digit(u[3], BGBIT, jj);
                    const double rc0 = a0 * t0.x, ic0 = b0 * t0.x, rs0 = a0 * t0.y, is0 = b0 * t0.y;
                    x0r[jj] = rc0 - is0; x0i[jj] = ic0 + rs0;
                    const double rc1 = a1 * t1.x, ic1 = b1 * t1.x, rs1 = a1 * t1.y, is1 = b1 * t1.y;
                    x1r[jj] = rc1 - is1; x1i[jj] = ic1 + rs1;
                }
                if (H == 0) {                           // one branch per point, not one per point and row
#pragma unroll
                    for (int jj = 0; jj < L; jj++) { yr[jj][m] = x0r[jj] + x1r[jj]; yi[jj][m] = x0i[jj] + x1i[jj]; }
                } else {
#pragma unroll
                    for (int jj = 0; jj < L; jj++) {
                        const double dr = x0r[jj] - x1r[jj], di = x0i[jj] - x1i[jj];
                        double p = dr * w1.x, q = di * w1.y;
                        yr[jj][m] = p - q;
                        p = dr * w1.y; q = di * w1.x;
                        yi[jj][m] = p + q;
                    }
                }
#ifndef HALVES_NO_POINT_BARRIER
                __builtin_amdgcn_sched_barrier(0);      // one point at a time: keeps the gather of later points from being hoisted
#endif
            }
            }
            HV_STAMP(1);
            prio_point(1);
            // the 512-point sub-transforms of the three rows side by side
#if HALVES_STAIRS == 2     // the finer staircase: pass 1 at 3, pass 2 at 2, pass 3 at 1, multiply-accumulate at 0
            auto step_down = [&]() { HV_PRIO(2); };
            fft_forward_multi_a<10, L, false, decltype(step_down)>(yr, yi, tw_sub, wbuf, wbuf + G::XSLOTS, ln, step_down);
#else
            HV_PRIO(2);
            fft_forward_multi_a<10, L, false>(yr, yi, tw_sub, wbuf, wbuf + G::XSLOTS, ln);
#endif
            prio_point(2);
            HV_PRIO(1);
#ifndef HALVES_FETCH_LATE
            fetch(bA, i, h * 2 * L);                    // (row 0, comp 0) of this polynomial: in flight under the last pass
#endif
            fft_forward_multi_b<10, L, BOOT_TRIV>(yr, yi, tw_sub);
#ifdef HALVES_FETCH_LATE
            fetch(bA, i, h * 2 * L);
#endif
            // hadamard + fold-add (spqlios.rs:204-222, trgsw.rs:290-299): this wave's half of the points, component 0 over the
            // polynomial's three rows, then component 1 (each accumulator still folds rows 0..5 in order); two key-row buffers,
            // each refilled as soon as its multiply-accumulate has retired
            HV_STAMP(2);
            prio_point(3);
            HV_PRIO(0);
            const int rc0 = h * 2 * L;                  // rc = 2 * row + comp
            fetch(bB, i, rc0 + 2);                                                     // (row 1, c0)
            mac_row<R>(s0re, s0im, bA, yr[0], yi[0]); fetch(bA, i, rc0 + 4);           // (row 2, c0)
            mac_row<R>(s0re, s0im, bB, yr[1], yi[1]); fetch(bB, i, rc0 + 1);           // (row 0, c1)
            mac_row<R>(s0re, s0im, bA, yr[2], yi[2]); fetch(bA, i, rc0 + 3);           // (row 1, c1)
            mac_row<R>(s1re, s1im, bB, yr[0], yi[0]); fetch(bB, i, rc0 + 5);           // (row 2, c1)
            mac_row<R>(s1re, s1im, bA, yr[1], yi[1]);
            mac_row<R>(s1re, s1im, bB, yr[2], yi[2]);
            HV_STAMP(3);
            prio_point(4);
        }

        // inverse: sub-network on this half, last stage across the halves, untwist, truncate, += acc
#pragma unroll 1
        for (int comp = 0; comp < 2; comp++) {
            double re[R], im[R];
#pragma unroll
            for (int m = 0; m < R; m++) { re[m] = comp ? s1re[m] : s0re[m]; im[m] = comp ? s1im[m] : s0im[m]; }
            int lane = lane0;
            asm volatile("" : "+v"(lane));      // as above: addresses are re-derived here instead of living (spilled) across the step
            [[maybe_unused]] cplx wl[R];        // half 1: last-stage twiddles (global memory), requested with the pass-1 table
            {
                Tw<G::NLOW - 4> w3; Tw<R - 1> w2, w1;
                w1.load(gip10 + lane, 64);                  // global memory: requested first, used last
                w3.load(twi_small + G::TW_P3, 1);
                P3<R, G::NLOW, G::LOW - 1, BOOT_TRIV>::template inv<false>(re, im, w3.w);
                w2.load(twi_small + G::TW_P2 + (lane & (G::NLOW - 1)), G::NLOW);
                exchange<10, 3, 2, true>(re, im, wbuf, lane);
                HV_PRIO(2);
                P12<R, G::LR - 1>::inv(re, im, w2.w);
                exchange<10, 2, 1, true>(re, im, wbuf, lane);
                HV_PRIO(1);
                if constexpr (IST_EARLY) {
                    if (H == 1) {   // requested under the last pass
#pragma unroll
                        for (int m = 0; m < R; m++) wl[m] = gist10[m * 64 + lane];
                    }
                }
                P12<R, G::LR - 1>::inv(re, im, w1.w);
            }
            HV_STAMP(4);
            prio_point(5);
            // now lane holds sub-points q = lane + 64 m of its half.  Last stage (halfnn = 512, spqlios-fft-impl.cpp:346-359):
            // t = x1 * w_q; half 0 keeps x0 + t, half 1 keeps x0 - t.  B sends t, A sends x0.
            if (H == 1) {
#pragma unroll
                for (int m = 0; m < R; m++) {
                    const cplx w = IST_EARLY ? wl[m] : gist10[m * 64 + lane];
                    const double t0 = re[m] * w.x, t1 = re[m] * w.y, t2 = im[m] * w.x, t3 = im[m] * w.y;
                    re[m] = t0 - t3; im[m] = t1 + t2;
                }
            }
#pragma unroll
            for (int m = 0; m < R; m++) { lds_st(&wbuf[lane + 64 * m], re[m]); lds_st(&wbuf[G::XSLOTS + lane + 64 * m], im[m]); }
            if constexpr (PINGPONG) HALVES_ARRIVE();
            Tw<R> wt;      // untwist (times 2/N) of this half's points, from global memory: in flight across the synchronisation
#pragma unroll
            for (int m = 0; m < R; m++) wt.w[m] = guntw0[(H * 8 + m) * 64 + lane];
            if constexpr (PINGPONG) HALVES_WAIT(); else HALVES_SYNC();
            {
                uint32_t* poly = accbuf + comp * N;
                // A: x0 + t, B: x0 - t.  Branch on the half outside the loop: a select would compute both (4 more FP64 instructions per point)
                if (H) {
#pragma unroll
                    for (int m = 0; m < R; m++) { re[m] = lds_ld(&rbuf[lane + 64 * m]) - re[m]; im[m] = lds_ld(&rbuf[G::XSLOTS + lane + 64 * m]) - im[m]; }
                } else {
#pragma unroll
                    for (int m = 0; m < R; m++) { re[m] = re[m] + lds_ld(&rbuf[lane + 64 * m]); im[m] = im[m] + lds_ld(&rbuf[G::XSLOTS + lane + 64 * m]); }
                }
#pragma unroll
                for (int m = 0; m < R; m++) {
                    const double vr = re[m], vi = im[m];
                    // (re, im) * (c, s): re c - im s, im c + re s   (spqlios-fft-impl.cpp:390-395)
                    const double rc = vr * wt.w[m].x, ic = vi * wt.w[m].x, rs = vr * wt.w[m].y, is = vi * wt.w[m].y;
                    const int c = lane + 64 * m + 512 * H;
                    poly[c] += trunc_to_torus(rc - is);
                    poly[c + P] += trunc_to_torus(ic + rs);
                }
            }
            if constexpr (PINGPONG) {
                // I own the buffer I have just read; my accumulator words are published by my next arrival (the other component's trade / the next
                // step's first row), which the partner waits for before it gathers them
                double* t = wbuf; wbuf = rbuf; rbuf = t;
            } else {
                HALVES_SYNC();     // the partner has read my buffer; both halves of the accumulator are written
            }
            HV_STAMP(5);
            prio_point(6);
        }
    }
    __builtin_amdgcn_s_setprio(0);
    if constexpr (PINGPONG) __syncthreads();      // the last accumulator update has no arrival behind it: both halves' words must be visible below
#ifdef RTFHE_WG_STAMPS
    if (a.dbg && blockIdx.x == 0 && lane0 == 0)
        for (int k = 0; k < 8; k++) a.dbg[wave * 8 + k] = tsum[k];
#endif
#ifdef ABL_NOKS       // timing ablation only (wrong results)
    if (a.steps >= 0) return;
#endif
#ifdef ABL_NOGATHER
#error "not wired for this kernel"
#endif

    if (a.mode == MODE_BLIND_ROTATE) {
        if (live) {
            uint32_t* o = a.out + (size_t)g * 2 * N;
            for (int c = lane0 + 64 * H; c < 2 * N; c += 128) o[c] = accbuf[c];
        }
        return;
    }

    // sample extract index 0 (trlwe.rs:110-121): a'_0 = a_0, a'_k = -a_{N-k}; b' = b_0
    {
        uint32_t av[2 * R];
#pragma unroll
        for (int mm = 0; mm < 2 * R; mm++) av[mm] = accbuf[N + lane0 + 64 * mm + 1024 * H];
        __syncthreads();
#pragma unroll
        for (int mm = 0; mm < 2 * R; mm++) {
            const int c = lane0 + 64 * mm + 1024 * H;
            accbuf[N + ((N - c) & (N - 1))] = (c == 0) ? av[mm] : (0u - av[mm]);
        }
    }
    __syncthreads();
    if (a.mode == MODE_EXTRACT) {      // the key switch of the whole batch follows as its own launch (k_key_switch_mm)
        if (live) {
            uint32_t* o = a.ext + (size_t)g * (N + 1);
            for (int c = H * (N / 2) + lane0; c < (H + 1) * (N / 2); c += 64) o[c] = accbuf[N + c];
            if (H == 0 && lane0 == 0) o[N] = accbuf[0];
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
        uint32_t* out = io.out;
#pragma unroll
        for (int q = 0; q < KSQ; q++) {
            const uint4 o = part[q * 64];
            const int col = 4 * (lane0 + 64 * q);
            const uint32_t s[4] = {sum[q].x + o.x, sum[q].y + o.y, sum[q].z + o.z, sum[q].w + o.w};
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (col + e <= n) out[col + e] = ((col + e == n) ? bprime : 0u) - s[e];
        }
    }
}

}  // namespace rtfhe
