// rtfhe_kernels_pair.hpp -- the bootstrap kernel with TWO WAVES PER GATE (N = 1024): the throughput shape.
//
// k_bootstrap gives a gate one wavefront.  With one gate per SIMD (the headline batch: 1024 gates on 1024 SIMDs) every
// SIMD then holds a single wave, and a lone wave issues one FP64 instruction per ~6.75 cycles where two waves sharing
// the SIMD reach one per ~5.4 (profiles/ubench/fp64_lds_issue_rates.log).  LDS (20 KiB of accumulator, exchange
// buffers and rotation amounts per gate) rules out more gates per CU, so here two waves share ONE gate's CMUX step.
// The arithmetic and its order are unchanged (see cmux_step for the reference citations):
//
//   side 0 (wave w, owns the b-poly)                       side 1 (wave w + GATES, owns the a-poly)
//   gather/decompose b-poly, transforms of rows 0..2       gather/decompose a-poly, transforms of rows 3..5   (spectra in VGPRs)
//   P: s0 = 0 + rows 0..2 of component 0    -> hand0
//   ------------------------------ hand-off 1 (the pair meets through its arrival flags in LDS) -----------------
//   Q: s1 = 0 + rows 0..2 of component 1    -> hand1       Q: s0 = hand0 + rows 3..5 of component 0  -> hand0
//   ------------------------------ hand-off 2 --------------------------------------------------------------------
//   s0 = hand0; inverse transform, += into the b-poly      R: s1 = hand1 + rows 3..5 of component 1; inverse, += into the a-poly
//
// * Fold order: every accumulator point sums rows 0, 1, ..., 5 from +0.0, exactly as the reference does
//   (trgsw.rs:290-299): the partial sums travel between the waves, products are never re-associated -> bit-identical.
// * Each side only ever reads and writes its OWN accumulator polynomial, so a side may run ahead into the next step's
//   gather and transforms; the two hand-offs per step order the partial sums only -- and only between the two waves of ONE gate (round 6:
//   flag_arrive / flag_wait, rtfhe_device.hpp; rounds 1-5 used the workgroup barrier at four gates per workgroup, which held the four gates in
//   lock step).  hand0 / hand1 are the (then idle) exchange buffers of side 0 / side 1: each is written only while its owner is between
//   transforms, in program order with the hand-offs.
// * Key rows go through a two-buffer register ring that runs ACROSS steps (each buffer is refilled right after its
//   multiply-accumulate retires; the last refills of a step fetch rows of the next).  Slot Q is the same code for both
//   sides, so the ring buffers are live in the same way on both paths at every control-flow merge -- the other
//   arrangements tried (per-side straight-line schedules, prefetch that only one side holds across a barrier) made the
//   register allocator spill 50-240 VGPRs, and scratch reloads share the vmcnt queue with the prefetches.
// * Priority schedule (see prio_point): a SIMD does not share itself evenly between two busy waves (one runs at ~0.9 of
//   its solo speed, the other on the leftovers; s_setprio selects which), so with fixed priorities one side races to each
//   hand-off and the SIMD then runs one wave; flipping side 0's priority twice per step makes both sides reach the
//   hand-offs together (8.16 -> 7.70 ms per 1024 gates).
// Measured (profiles/r01_pair): 7.70-7.75 ms per 1024 gates vs 8.69 ms for k_bootstrap's 4-wave shape, same outputs.
#pragma once

#include "rtfhe_kernels.hpp"

// What was tried on this kernel and measured slower (other priority schedules, a split-phase first hand-off, 8-byte hand-offs, 16-byte exchanges,
// the symmetric priority staircase, other wave placements) is recorded in profiles/HISTORY.md; the code below is what ships.

namespace rtfhe {

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a workgroup-scope fence over ALL address
// spaces, i.e. s_waitcnt vmcnt(0): it would wait for the key rows prefetched across it.  Hand-offs here go through LDS.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Synchronisation of the TWO waves of one gate only, busy-polling form (k_bootstrap_pair's until round 6, still the four-wave and NTT kernels'
// at small workgroups; k_bootstrap_pair now uses the sleepy scalar-addressed flag_arrive / flag_wait of rtfhe_device.hpp at every size).  s_barrier is workgroup-wide although the four gates of a
// workgroup share nothing after start-up; the phase stamps show BOTH sides of a pair ~1.1 k cycles "at barrier 1", which looked like the pairs
// waiting for the slowest gate.  Here each side publishes an arrival counter in LDS after its hand-off stores and polls its partner's (DS
// instructions of a wave execute in order, so a partner that sees counter >= k also sees the stores issued before it; no fence, which would wait
// for the key rows in flight).  Identical outputs -- and no faster: 6.76 vs 6.74 ms per 1024 gates, 4.34 vs 4.37 at 512, 6.14 vs 6.20 at 768
// (profiles/r03/pair_flag_sync_ab.log).  The time at the barrier is the side's own hand-off stores draining, not skew between gates.
__device__ __forceinline__ void pair_sync(unsigned my_flag_addr, unsigned partner_flag_addr, unsigned k) {
    asm volatile("ds_write_b32 %0, %1" ::"v"(my_flag_addr), "v"(k) : "memory");
    unsigned v;
    do {
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(partner_flag_addr) : "memory");
        v = (unsigned)__builtin_amdgcn_readfirstlane((int)v);
    } while ((int)(v - k) < 0);
}

// The two halves of pair_sync as separate calls (split phase): a wave publishes its arrival, does work that needs nothing from its partner, and only
// then waits -- the flag's LDS round trip lands under that work instead of idling the wave.
__device__ __forceinline__ void pair_arrive(unsigned my_flag_addr, unsigned k) {
    asm volatile("ds_write_b32 %0, %1" ::"v"(my_flag_addr), "v"(k) : "memory");
}
__device__ __forceinline__ void pair_wait(unsigned partner_flag_addr, unsigned k) {
    unsigned v;
    do {
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(partner_flag_addr) : "memory");
        v = (unsigned)__builtin_amdgcn_readfirstlane((int)v);
    } while ((int)(v - k) < 0);
}

// pair_wait as ONE opaque statement: the spin loop lives inside the inline assembly, so the compiler sees straight-line code (a wait in the
// middle of a multiply-accumulate slot, with 200 registers live across it, otherwise becomes a loop header and the allocator spilled 88 of them)
__device__ __forceinline__ void pair_wait_opaque(unsigned partner_flag_addr, unsigned k) {
    unsigned v, t;
    asm volatile(
        "1:\n\t"
        "ds_read_b32 %0, %2\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_readfirstlane_b32 %1, %0\n\t"
        "s_sub_i32 %1, %1, %3\n\t"
        "s_cmp_lt_i32 %1, 0\n\t"
        "s_cbranch_scc1 1b"
        : "=&v"(v), "=&s"(t)
        : "v"(partner_flag_addr), "s"(k)
        : "memory", "scc");
}

// ... and with the wave's priority raised to 3 behind the wait, inside the same statement (a separate s_setprio is one more scheduling boundary
// for the compiler: in k_bootstrap_eo at 256 registers eight of them per step cost 15 spilled registers)
__device__ __forceinline__ void pair_wait_opaque_prio3(unsigned partner_flag_addr, unsigned k) {
    unsigned v, t;
    asm volatile(
        "1:\n\t"
        "ds_read_b32 %0, %2\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_readfirstlane_b32 %1, %0\n\t"
        "s_sub_i32 %1, %1, %3\n\t"
        "s_cmp_lt_i32 %1, 0\n\t"
        "s_cbranch_scc1 1b\n\t"
        "s_setprio 3"
        : "=&v"(v), "=&s"(t)
        : "v"(partner_flag_addr), "s"(k)
        : "memory", "scc");
}

// the first row of a fold: the reference adds it to FrrSeries::zero() (trgsw.rs:290-299); +0.0 + x == x for every x except that
// it turns a -0.0 into +0.0, and the sign of a zero never reaches a torus word (see fwd_stage_tw in rtfhe_device.hpp)
template <int R>
__device__ __forceinline__ void mac_row_first(double (&sre)[R], double (&sim)[R], const cplx (&b)[R], const double (&re)[R], const double (&im)[R]) {
#pragma unroll
    for (int m = 0; m < R; m++) {
        const double ii = b[m].y * im[m], rr = b[m].x * re[m], ri = b[m].x * im[m], ir = b[m].y * re[m];
        sre[m] = rr - ii;
        sim[m] = ir + ri;
    }
}

template <int R>
__device__ __forceinline__ void mac_row(double (&sre)[R], double (&sim)[R], const cplx (&b)[R], const double (&re)[R], const double (&im)[R]) {
    // hadamard + fold-add, utils/src/spqlios.rs:204-222, hom_nand/src/trgsw.rs:290-299 (same operation order as cmux_step)
#pragma unroll
    for (int m = 0; m < R; m++) {
        const double ii = b[m].y * im[m], rr = b[m].x * re[m], ri = b[m].x * im[m], ir = b[m].y * re[m];
        sre[m] = sre[m] + (rr - ii);
        sim[m] = sim[m] + (ir + ri);
    }
}

struct PairLds {
    typedef Geo<10> G;
    static constexpr size_t TW = (size_t)G::TW_TOTAL * sizeof(cplx);
    static constexpr size_t XB = (size_t)2 * G::XSLOTS * sizeof(double);            // one wave's re + im exchange buffers
    static_assert(XB >= (size_t)G::P * sizeof(cplx), "an exchange buffer pair must hold one spectrum");
    static constexpr size_t FLAGS = 16;       // two arrival counters per gate (pair_sync), 16-byte aligned tail of the gate's region
    __host__ __device__ static constexpr size_t gate_bytes(int npad) { return (size_t)2 * G::N * 4 + (size_t)npad * 4 + 2 * XB + FLAGS; }
    __host__ __device__ static constexpr size_t bytes(int gates, int npad) { return TW + (size_t)gates * gate_bytes(npad); }
};

// GATES gates per workgroup, 2 * GATES waves: wave w serves gate (w % GATES) as side (w / GATES)
template <int L, int BGBIT, int KS_T, int KS_BB, int KSQ, int GATES>
__global__ __launch_bounds__(128 * GATES, 1) void k_bootstrap_pair(const BootstrapArgs a) {
    constexpr int LOGN = 10;
    typedef Geo<LOGN> G;
    constexpr int N = G::N, P = G::P, R = G::R, NT = 128 * GATES;
    constexpr uint32_t M = decomp_mask(L, BGBIT);
    static_assert(L == 3, "three rows per side are held in registers");
    // The two waves of a gate meet through their own arrival flags in LDS at every workgroup size, never through the workgroup barrier (round 6).
    // Round 3 had measured the barrier and a busy-polling pair_sync equal at four gates per workgroup (6.74 vs 6.76 ms); what the split-FFT
    // kernel then showed (rtfhe_kernels_xfft.hpp: gates held in lock step collide on what a CU shares) holds here too once the wait costs the
    // SIMD's other wave nothing -- flag addresses in scalar registers, a sleep between polls: 6.56 -> 6.41 ms per 1,024 gates, 52.06 -> 51.37
    // per 8,192 (profiles/r06/pair_flags_ab.log).
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // which gate and side a wave serves.  Waves go to the SIMDs round robin (wave w on SIMD w % 4): the two sides of a gate share a SIMD (waves w,
    // w + GATES) -- they are in complementary phases, each other's best SIMD partner (profiles/r03/wave_placement_on_simds_ab.log)
    const int slot = wave % GATES;
    const int side = wave / GATES;
    cplx* tw = reinterpret_cast<cplx*>(smem);
    for (int idx = tid; idx < G::TW_TOTAL; idx += NT) tw[idx] = a.tw[idx];
    const cplx* twf = tw;
    const cplx* twi = tw + G::TW_DIR;

    // idle pairs of the last workgroup shadow the last gate (they run every step and take part in the barriers of prologue and epilogue) and store nothing
    const int g_raw = blockIdx.x * GATES + slot;
    const int g = g_raw < a.count ? g_raw : a.count - 1;
    const GateIo io = gate_io(a, g);
    const bool live = g_raw < a.count && io.ok;      // a skipped netlist gate still runs every step

    unsigned char* gbase = smem + PairLds::TW + (size_t)slot * PairLds::gate_bytes(a.npad);
    uint32_t* accbuf = reinterpret_cast<uint32_t*>(gbase);
    uint32_t* abar = accbuf + 2 * N;
    double* xb0 = reinterpret_cast<double*>(gbase + (size_t)2 * N * 4 + (size_t)a.npad * 4);
    // arrival counters of the pair (zeroed before the start-up barrier)
    uint32_t* flags = reinterpret_cast<uint32_t*>(gbase + PairLds::gate_bytes(a.npad) - PairLds::FLAGS);
    if (lane == 0) flags[side] = 0u;
    const unsigned my_flag = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) uint32_t*)(flags + side));
    const unsigned partner_flag = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) uint32_t*)(flags + (1 - side)));
    double* xb1 = xb0 + 2 * G::XSLOTS;
    double* myx = side ? xb1 : xb0;
    cplx* hand0 = reinterpret_cast<cplx*>(xb0) + lane;    // [R][64] cplx
    cplx* hand1 = reinterpret_cast<cplx*>(xb1) + lane;
    uint32_t* poly = accbuf + side * N;

    const int n = a.n;
    {   // pre-step + mod switch (tfhe.rs:41-71, 97, 107-108)
        constexpr int SH = 32 - LOGN - 1;
        for (int i = lane + 64 * side; i <= n; i += 128) {
            const uint32_t t = gate_linear(io.op, io.p0[i], io.p1[i], i == n);
            abar[i] = (i == n) ? (t >> SH) : ((t + (1u << (SH - 1))) >> SH);
        }
    }
    __syncthreads();
    {   // acc = X^{-bbar} * testvec (tfhe.rs:85, 98-106): side 0 holds the b-poly, side 1 the (zero) a-poly
        const int bbar = (int)abar[n];
#pragma unroll
        for (int mm = 0; mm < 2 * R; mm++) {
            const int c = lane + 64 * mm;
            const int e = (c + bbar) & (2 * N - 1);
            poly[c] = side ? 0u : ((e >> LOGN) ? 0xE0000000u : 0x20000000u);
        }
    }
    wave_lds_sync();
    // this side's own coefficients (lane + 64 mm) are handed from the update at the end of a step to the gather that
    // opens the next one in registers: the gather then reads only the rotated coefficients from LDS
    uint32_t own[2 * R];
#pragma unroll
    for (int mm = 0; mm < 2 * R; mm++) own[mm] = poly[lane + 64 * mm];

#ifdef RTFHE_WG_STAMPS
    unsigned long long tsum[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
#define PAIR_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tsum[k] += t_ - tprev; tprev = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PAIR_STAMP(k) do { } while (0)
#endif
    const size_t trgsw_cplx = (size_t)2 * L * 2 * R * 64;
    // Key rows in consumption order rc = 0..5: (row rc % 3, component rc / 3) of this side.  A ring of two 8-point buffers
    // runs across steps: each is refilled right after its multiply-accumulate retires, two MACs ahead of its use, the
    // last two refills of a step fetching rows 0, 1 of the next one.
    cplx bA[R], bB[R];
    // Key rows are read through a buffer resource over the whole key: address = descriptor base + scalar row offset + lane * 16 +
    // immediate.  The row offset lives in SGPRs (SALU arithmetic), the per-lane part is one VGPR for the whole kernel: no
    // 64-bit vector address arithmetic per load (global_load with a vector address cost ~50 VALU instructions per step).
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t bk_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<cplx*>(a.bk), 0, 0x7fffffff, 0x00020000);
    const int lane16 = lane * 16;
    auto fetch = [&](cplx (&dst)[R], int step, int rc) {
        const size_t row = (size_t)step * trgsw_cplx + (size_t)((side * L + rc % L) * 2 + rc / L) * R * 64;
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
    if (a.steps > 0) {   // side 0 starts its ring with (bB, bA), side 1 with (bA, bB)
        fetch(bA, 0, side ? 0 : 1);
        fetch(bB, 0, side ? 1 : 0);
    }
    // Priority schedule.  Of two waves that both have work a SIMD runs one at (nearly) full speed and the other on the
    // leftovers (s_setprio selects which), so with fixed priorities the favoured side reaches every hand-off early and the
    // SIMD then runs a single wave.  Side 1 stays at priority 1; side 0 runs at 2 from the end of a step (RAISE_AT) to the end of its pass 2
    // (LOWER_AT) and at 0 for the rest of the step -- its inverse then runs at low priority under side 1's slot R + inverse -- which splits the
    // time between the hand-offs about evenly (profiles/r01_pair/priority_schedule_ab.log, profiles/r03/pair_priority_grid.log; the search over
    // every schedule these points allow: profiles/r04/pair_priority_search.log).  Points: 1 after pass 1, 2 after pass 2, 5 after pass 3,
    // 6 before hand-off 1, 7 after it, 8 before hand-off 2, 9 after it, 10 end of step (the window was searched again under the flag form in
    // round 6 and is still the optimum: profiles/r06/pair_flags_ab.log).
    constexpr int LOWER_AT = 2, RAISE_AT = 10;
    auto prio_point = [&](int point) {   // one opaque statement each: no compiler-visible control flow inside the transforms
        if (point == LOWER_AT) asm volatile("s_cmp_lg_u32 %0, 0\n\ts_cbranch_scc1 1f\n\ts_setprio 0\n1:" ::"s"(side) : "scc");
        if (point == RAISE_AT) asm volatile("s_cmp_lg_u32 %0, 0\n\ts_cbranch_scc1 1f\n\ts_setprio 2\n1:" ::"s"(side) : "scc");
    };
    if (side) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(2);
#pragma unroll 1
    for (int i = 0; i < a.steps; i++) {
        const int r = __builtin_amdgcn_readfirstlane((int)abar[i]);
        const int nxt = (i + 1 < a.steps) ? i + 1 : i;
        // an opaque copy of the lane id per step: without it the compiler hoists every lane-derived LDS address out of
        // the loop and then spills them (each is one VALU op to recompute)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        uint32_t u[2 * R];
#pragma unroll
        for (int mm = 0; mm < 2 * R; mm++) {
            const int c = ln + 64 * mm;
            u[mm] = ((rotated_coef<LOGN>(poly, c, r) - own[mm]) + M) ^ M;
        }
        PAIR_STAMP(0);
        double xr[L][R], xi[L][R];
#pragma unroll
        for (int jj = 0; jj < L; jj++) {
#pragma unroll
            for (int m = 0; m < R; m++) {
                xr[jj][m] = (double)decomp_digit(u[m], BGBIT, jj);
                xi[jj][m] = (double)decomp_digit(u[R + m], BGBIT, jj);
            }
        }
        // the three digit rows side by side: twiddles loaded once per pass, a row's exchange in flight under the next rows' passes
        auto pp1 = [&]() { prio_point(1); };
        fft_forward_multi_a<LOGN, L, true, decltype(pp1), true>(xr, xi, twf, myx, myx + G::XSLOTS, ln, pp1);
        prio_point(2);
        PAIR_STAMP(1);
        fft_forward_multi_b<LOGN, L, BOOT_TRIV>(xr, xi, twf);
        prio_point(5);
        PAIR_STAMP(3);

        double sre[R], sim[R];
        auto zero = [&]() {
#pragma unroll
            for (int m = 0; m < R; m++) { sre[m] = 0.0; sim[m] = 0.0; }
        };
        auto put = [&](cplx* h) {
#pragma unroll
            for (int m = 0; m < R; m++) h[m * 64] = make_double2(sre[m], sim[m]);
        };
        auto get = [&](const cplx* h) {
#pragma unroll
            for (int m = 0; m < R; m++) { const cplx v = h[m * 64]; sre[m] = v.x; sim[m] = v.y; }
        };

        // slot P (side 0): component 0 over rows 0..2 from +0.0 (the first row without its "+0.0 +", see mac_row_first: with the unit-twiddle
        // butterflies 64 fewer FP64 instructions per CMUX, the same torus words, 3.5 % -- profiles/r03/pair_unit_twiddle_first_row_ab.log)
        if (side == 0) {
            mac_row_first<R>(sre, sim, bB, xr[0], xi[0]); fetch(bB, i, 2);
            mac_row<R>(sre, sim, bA, xr[1], xi[1]); fetch(bA, i, 3);
            mac_row<R>(sre, sim, bB, xr[2], xi[2]); fetch(bB, i, 4);
            put(hand0);
        }
        prio_point(6);
        PAIR_STAMP(4);
        flag_arrive(my_flag, 2u * (unsigned)i + 1u); flag_wait(partner_flag, 2u * (unsigned)i + 1u);
        prio_point(7);
        PAIR_STAMP(5);
        // slot Q (both, same code): side 0 component 1 over rows 0..2 from +0.0 -> hand1; side 1 component 0 over rows 3..5
        // on top of side 0's partial sum -> hand0
        // (the same code for both sides, see the header: side 0's fold keeps its explicit +0.0 start here -- a side-dependent
        // first row made the allocator spill 60 VGPRs)
        if (side == 0) zero(); else get(hand0);
        mac_row<R>(sre, sim, bA, xr[0], xi[0]); fetch(bA, i, side ? 2 : 5);
        mac_row<R>(sre, sim, bB, xr[1], xi[1]); fetch(bB, side ? i : nxt, side ? 3 : 0);
        mac_row<R>(sre, sim, bA, xr[2], xi[2]); fetch(bA, side ? i : nxt, side ? 4 : 1);
        put(side ? hand0 : hand1);
        prio_point(8);
        PAIR_STAMP(6);
        flag_arrive(my_flag, 2u * (unsigned)i + 2u); flag_wait(partner_flag, 2u * (unsigned)i + 2u);
        prio_point(9);
        PAIR_STAMP(7);
        // slot R (side 1): component 1 over rows 3..5 on top of side 0's partial sum; side 0 picks up the finished s0
        if (side == 1) {
            get(hand1);
            mac_row<R>(sre, sim, bB, xr[0], xi[0]); fetch(bB, i, 5);
            mac_row<R>(sre, sim, bA, xr[1], xi[1]); fetch(bA, nxt, 0);
            mac_row<R>(sre, sim, bB, xr[2], xi[2]); fetch(bB, nxt, 1);
        } else {
            get(hand0);
        }
        PAIR_STAMP(8);

        // the 2/N input scaling of the reference (fft_processor_spqlios.cpp:158) is folded into the untwist twiddles
        fft_inverse<LOGN, 1, BOOT_TRIV>(sre, sim, twi, twi, myx, lane);
#pragma unroll
        for (int m = 0; m < R; m++) {
            const int c = lane + 64 * m;
            own[m] += trunc_to_torus(sre[m]);                   // own[] IS this side's polynomial: no read-back; kept for the next gather
            own[R + m] += trunc_to_torus(sim[m]);
            poly[c] = own[m];
            poly[c + P] = own[R + m];
        }
        wave_lds_sync();
        prio_point(10);
        PAIR_STAMP(9);
    }
    __builtin_amdgcn_s_setprio(0);
#ifdef RTFHE_WG_STAMPS
    if (a.dbg && blockIdx.x == 0 && lane == 0)
        for (int k = 0; k < 16; k++) a.dbg[wave * 16 + k] = tsum[k];
#endif

    if (a.mode == MODE_BLIND_ROTATE) {
        if (live) {
            uint32_t* o = a.out + (size_t)g * 2 * N + side * N;
            for (int c = lane; c < N; c += 64) o[c] = poly[c];
        }
        return;
    }

    // sample extract index 0 (trlwe.rs:110-121): a'_0 = a_0, a'_k = -a_{N-k}; b' = b_0.  Side 1 owns the a-poly.
    if (side == 1) {
        uint32_t av[2 * R];
#pragma unroll
        for (int mm = 0; mm < 2 * R; mm++) av[mm] = poly[lane + 64 * mm];
        wave_lds_sync();
#pragma unroll
        for (int mm = 0; mm < 2 * R; mm++) {
            const int c = lane + 64 * mm;
            poly[(N - c) & (N - 1)] = (c == 0) ? av[mm] : (0u - av[mm]);
        }
    }
    __syncthreads();
    if (a.mode == MODE_EXTRACT) {      // the key switch of the whole batch follows as its own launch (k_key_switch_mm)
        if (live) {
            const int ge = a.ext_first + g;      // batch-wide gate number: the sample buffer is laid out for the key switch (ext_slot)
            for (int c = side * (N / 2) + lane; c < (side + 1) * (N / 2); c += 64) *ext_slot(a.ext, ge, c, N) = accbuf[N + c];
            if (side == 0 && lane == 0) *ext_slot(a.ext, ge, N, N) = accbuf[0];
            for (int c = side * 64 + lane; c <= n; c += 128) io.out[c] = 0u;
        }
        return;
    }
    // identity key switch (tlwe.rs:43-73): each side sums the rows of half of the coefficients
    uint4 sum[KSQ];
    ks_accumulate<LOGN, KS_T, KS_BB, KSQ>(accbuf + N, side * (N / 2), (side + 1) * (N / 2), a.ksk, a.ksw, sum, lane);
    uint4* part = reinterpret_cast<uint4*>(xb1) + lane;   // [KSQ][64] uint4
    if (side == 1) {
#pragma unroll
        for (int q = 0; q < KSQ; q++) part[q * 64] = sum[q];
    }
    __syncthreads();
    if (side == 0 && live) {
        const uint32_t bprime = accbuf[0];
        uint32_t* out = io.out;
#pragma unroll
        for (int q = 0; q < KSQ; q++) {
            const uint4 o = part[q * 64];
            const int col = 4 * (lane + 64 * q);
            const uint32_t s[4] = {sum[q].x + o.x, sum[q].y + o.y, sum[q].z + o.z, sum[q].w + o.w};
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (col + e <= n) out[col + e] = ((col + e == n) ? bprime : 0u) - s[e];
        }
    }
}

}  // namespace rtfhe
