// rtfhe_kernels_pair_rr.hpp -- k_bootstrap_pair with FIVE or SIX gates on the four wave pairs of a CU (N = 1024): batches between whole rounds.
//
// k_bootstrap_pair fills a CU with four gates (eight waves at 256 registers: two per SIMD, the issue optimum) and a gate takes the whole launch,
// so a batch runs in rounds of 4 x CUs gates: 1,280 gates on 256 CUs cost a full round (6.4 ms) plus a round for the last 256 (2.7 ms on the
// latency kernel) although the arithmetic is 1.25 rounds.  Here a workgroup serves gc = 4 .. 6 gates on the same four pairs of waves by time
// slicing: the CMUX steps of its gates form one sequence of items t = step * gc + gate, and pair s works through the items t = s, s + 4, s + 8, ...
// Item t needs item t - gc (the same gate's previous step) -- at least one whole item earlier in ANOTHER pair's sequence when gc > 4 -- and
// everything a gate carries from step to step lives in LDS (its two polynomials, its rotation amounts), so a gate simply changes hands: each
// side publishes "step i of gate g done" in a flag of the gate after its update and waits for that flag before it gathers.  (A side only ever
// reads and writes its own polynomial: the flag is per gate AND side, written by one wave, waited for by one.)  The pairs free-run as in
// k_bootstrap_pair; the arithmetic and its order are that kernel's, operation for operation, so the outputs are bit-identical.
// Cost against k_bootstrap_pair: the side's own coefficients are read back from LDS at the gather (16 ds_read_b32 per wave and item) instead of
// staying in registers, and the rotation amounts are stored as 16-bit words so that six gates fit (163,328 of 163,840 bytes of LDS).
// A batch of 4 W < count <= 6 W gates on W CUs takes ceil(count / W) / 4 rounds instead of 2: on one box 1,025 gates 10.0 -> 8.6 ms, 1,280 gates
// 9.7 -> 8.4-8.7, 1,536 gates 10.7 -> 10.3-10.4, 2,304 gates 16.3 -> 15.0 (profiles/r06/pair_rr_sweep.log).  A SEVENTH gate fits once the big half
// of the inverse table is read from global memory instead (as at N = 2048); built, bit-identical, and no faster than a whole round plus a
// three-gates-per-CU tail (1,792 gates 12.1 against 12.2 ms, 1,600 gates 12.5 against 12.3): not kept.
#pragma once

#include "rtfhe_kernels_pair.hpp"

namespace rtfhe {

struct PairRrLds {
    typedef Geo<10> G;
    static constexpr int SLOTS = 4, GMAX = 6;
    static constexpr size_t TW = PairLds::TW, XB = PairLds::XB;
    static constexpr size_t SLOT = 2 * XB + 16;            // a pair's exchange / hand-off buffers and its two arrival counters
    static constexpr size_t DONE = 64;                     // [GMAX][2 sides] steps done, per gate and side
    __host__ __device__ static constexpr size_t gate_bytes(int npad) { return (size_t)2 * G::N * 4 + ((size_t)npad * 2 + 15) / 16 * 16; }
    __host__ __device__ static constexpr size_t bytes(int gates, int npad) { return TW + SLOTS * SLOT + DONE + (size_t)gates * gate_bytes(npad); }
};

// gridDim.x workgroups share a.count gates evenly (the first a.count % gridDim.x take one more); the host launches 4 <= gates per workgroup <= 6
template <int L, int BGBIT, int KS_T, int KS_BB, int KSQ>
__global__ __launch_bounds__(512, 1) void k_bootstrap_pair_rr(const BootstrapArgs a) {
    typedef PairRrLds S;
    constexpr int LOGN = 10, SLOTS = S::SLOTS;
    typedef Geo<LOGN> G;
    constexpr int N = G::N, P = G::P, R = G::R, NT = 128 * SLOTS;
    constexpr uint32_t M = decomp_mask(L, BGBIT);
    static_assert(L == 3, "three rows per side are held in registers");
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slot = wave % SLOTS;          // the pair (waves slot, slot + 4 share a SIMD, as in k_bootstrap_pair)
    const int side = wave / SLOTS;
    const int per = a.count / (int)gridDim.x, extra = a.count % (int)gridDim.x, wg = (int)blockIdx.x;
    const int gc = per + (wg < extra ? 1 : 0);
    const int g_first = wg * per + (wg < extra ? wg : extra);
    if (gc < SLOTS || gc > S::GMAX) return;      // not a shape this kernel serves (the host never launches one): uniform exit

    cplx* tw = reinterpret_cast<cplx*>(smem);
    for (int idx = tid; idx < G::TW_TOTAL; idx += NT) tw[idx] = a.tw[idx];
    const cplx* twf = tw;
    const cplx* twi = tw + G::TW_DIR;

    unsigned char* sbase = smem + S::TW + (size_t)slot * S::SLOT;
    double* xb0 = reinterpret_cast<double*>(sbase);
    double* xb1 = xb0 + 2 * G::XSLOTS;
    double* myx = side ? xb1 : xb0;
    cplx* hand0 = reinterpret_cast<cplx*>(xb0) + lane;    // [R][64] cplx
    cplx* hand1 = reinterpret_cast<cplx*>(xb1) + lane;
    uint32_t* flags = reinterpret_cast<uint32_t*>(sbase + 2 * S::XB);
    uint32_t* done = reinterpret_cast<uint32_t*>(smem + S::TW + SLOTS * S::SLOT);
    unsigned char* gates0 = smem + S::TW + SLOTS * S::SLOT + S::DONE;
    const size_t gate_bytes = S::gate_bytes(a.npad);
    if (lane == 0) flags[side] = 0u;
    if (tid < 2 * S::GMAX) done[tid] = 0u;
    const unsigned my_flag = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) uint32_t*)(flags + side));
    const unsigned partner_flag = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) uint32_t*)(flags + (1 - side)));
    const unsigned done_side = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) uint32_t*)(done + side));

    const int n = a.n;
    for (int gl = 0; gl < gc; gl++) {   // pre-step + mod switch (tfhe.rs:41-71, 97, 107-108) of every gate of the workgroup
        constexpr int SH = 32 - LOGN - 1;
        const GateIo io = gate_io(a, g_first + gl);
        uint16_t* ab = reinterpret_cast<uint16_t*>(gates0 + gl * gate_bytes + (size_t)2 * N * 4);
        for (int i = tid; i <= n; i += NT) {
            const uint32_t t = gate_linear(io.op, io.p0[i], io.p1[i], i == n);
            ab[i] = (uint16_t)((i == n) ? (t >> SH) : ((t + (1u << (SH - 1))) >> SH));
        }
    }
    __syncthreads();
    for (int gl = 0; gl < gc; gl++) {   // acc = X^{-bbar} * testvec (tfhe.rs:85, 98-106)
        uint32_t* acc = reinterpret_cast<uint32_t*>(gates0 + gl * gate_bytes);
        const int bbar = (int)reinterpret_cast<const uint16_t*>(acc + 2 * N)[n];
        for (int c = tid; c < N; c += NT) {
            const int e = (c + bbar) & (2 * N - 1);
            acc[c] = (e >> LOGN) ? 0xE0000000u : 0x20000000u;
            acc[N + c] = 0u;
        }
    }
    __syncthreads();

    const size_t trgsw_cplx = (size_t)2 * L * 2 * R * 64;
    // the key ring of k_bootstrap_pair, running across ITEMS: the last refills of an item fetch rows of the step of this pair's next item
    cplx bA[R], bB[R];
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
    if (a.steps > 0) {
        fetch(bA, 0, side ? 0 : 1);
        fetch(bB, 0, side ? 1 : 0);
    }
    constexpr int LOWER_AT = 2, RAISE_AT = 10;      // k_bootstrap_pair's priority schedule
    auto prio_point = [&](int point) {
        if (point == LOWER_AT) asm volatile("s_cmp_lg_u32 %0, 0\n\ts_cbranch_scc1 1f\n\ts_setprio 0\n1:" ::"s"(side) : "scc");
        if (point == RAISE_AT) asm volatile("s_cmp_lg_u32 %0, 0\n\ts_cbranch_scc1 1f\n\ts_setprio 2\n1:" ::"s"(side) : "scc");
    };
    if (side) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(2);
    unsigned seq = 0;                               // the pair's hand-offs, counted over the whole kernel
    int gl = slot, i = 0;                           // this pair's current item: step i of gate gl
#pragma unroll 1
    while (i < a.steps) {
        int gl2 = gl + SLOTS, i2 = i;               // ... and its next one
        if (gl2 >= gc) { gl2 -= gc; i2++; }
        const int nxt = (i2 < a.steps) ? i2 : i;
        uint32_t* poly = reinterpret_cast<uint32_t*>(gates0 + gl * gate_bytes) + side * N;
        const int r = __builtin_amdgcn_readfirstlane((int)reinterpret_cast<const uint16_t*>(gates0 + gl * gate_bytes + (size_t)2 * N * 4)[i]);
        const unsigned done_flag = done_side + 8u * (unsigned)gl;
        flag_wait(done_flag, (unsigned)i);          // this side's polynomial of gate gl has been through step i - 1 (on whichever pair ran it)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        uint32_t own[2 * R], u[2 * R];
#pragma unroll
        for (int mm = 0; mm < 2 * R; mm++) {
            const int c = ln + 64 * mm;
            own[mm] = poly[c];
            u[mm] = ((rotated_coef<LOGN>(poly, c, r) - own[mm]) + M) ^ M;
        }
        double xr[L][R], xi[L][R];
#pragma unroll
        for (int jj = 0; jj < L; jj++) {
#pragma unroll
            for (int m = 0; m < R; m++) {
                xr[jj][m] = (double)decomp_digit(u[m], BGBIT, jj);
                xi[jj][m] = (double)decomp_digit(u[R + m], BGBIT, jj);
            }
        }
        auto pp1 = [&]() { prio_point(1); };
        fft_forward_multi_a<LOGN, L, true, decltype(pp1), true>(xr, xi, twf, myx, myx + G::XSLOTS, ln, pp1);
        prio_point(2);
        fft_forward_multi_b<LOGN, L, BOOT_TRIV>(xr, xi, twf);
        prio_point(5);

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

        // slots P, Q, R of k_bootstrap_pair (the fold order of trgsw.rs:290-299: rows 0 .. 5 from +0.0, partial sums travel between the sides)
        if (side == 0) {
            mac_row_first<R>(sre, sim, bB, xr[0], xi[0]); fetch(bB, i, 2);
            mac_row<R>(sre, sim, bA, xr[1], xi[1]); fetch(bA, i, 3);
            mac_row<R>(sre, sim, bB, xr[2], xi[2]); fetch(bB, i, 4);
            put(hand0);
        }
        prio_point(6);
        seq++; flag_arrive(my_flag, seq); flag_wait(partner_flag, seq);
        prio_point(7);
        if (side == 0) zero(); else get(hand0);
        mac_row<R>(sre, sim, bA, xr[0], xi[0]); fetch(bA, i, side ? 2 : 5);
        mac_row<R>(sre, sim, bB, xr[1], xi[1]); fetch(bB, side ? i : nxt, side ? 3 : 0);
        mac_row<R>(sre, sim, bA, xr[2], xi[2]); fetch(bA, side ? i : nxt, side ? 4 : 1);
        put(side ? hand0 : hand1);
        prio_point(8);
        seq++; flag_arrive(my_flag, seq); flag_wait(partner_flag, seq);
        prio_point(9);
        if (side == 1) {
            get(hand1);
            mac_row<R>(sre, sim, bB, xr[0], xi[0]); fetch(bB, i, 5);
            mac_row<R>(sre, sim, bA, xr[1], xi[1]); fetch(bA, nxt, 0);
            mac_row<R>(sre, sim, bB, xr[2], xi[2]); fetch(bB, nxt, 1);
        } else {
            get(hand0);
        }

        fft_inverse<LOGN, 1, BOOT_TRIV>(sre, sim, twi, twi, myx, lane);
#pragma unroll
        for (int m = 0; m < R; m++) {
            const int c = lane + 64 * m;
            poly[c] = own[m] + trunc_to_torus(sre[m]);
            poly[c + P] = own[R + m] + trunc_to_torus(sim[m]);
        }
        wave_lds_sync();
        flag_arrive(done_flag, (unsigned)i + 1u);   // behind the stores above in this wave's LDS queue: who sees the flag sees the polynomial
        prio_point(10);
        gl = gl2; i = i2;
    }
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();       // every pair has left its item loop: all gates are through their last step

    // epilogue, gate by gate on the pair that started it (k_bootstrap_pair's, with the pair's own hand-offs in place of the workgroup barrier:
    // pairs serve different numbers of gates here)
#define RR_PAIR_SYNC() do { seq++; flag_arrive(my_flag, seq); flag_wait(partner_flag, seq); } while (0)
#pragma unroll 1
    for (int ge_l = slot; ge_l < gc; ge_l += SLOTS) {
        const int g = g_first + ge_l;
        const GateIo io = gate_io(a, g);
        const bool live = io.ok;                 // a skipped netlist gate has run every step and stores nothing
        uint32_t* accbuf = reinterpret_cast<uint32_t*>(gates0 + ge_l * gate_bytes);
        uint32_t* poly = accbuf + side * N;
        if (a.mode == MODE_BLIND_ROTATE) {
            if (live) {
                uint32_t* o = a.out + (size_t)g * 2 * N + side * N;
                for (int c = lane; c < N; c += 64) o[c] = poly[c];
            }
            continue;
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
        RR_PAIR_SYNC();
        if (a.mode == MODE_EXTRACT) {      // the key switch of the whole batch follows as its own launch (k_key_switch_mm)
            if (live) {
                const int ge = a.ext_first + g;
                for (int c = side * (N / 2) + lane; c < (side + 1) * (N / 2); c += 64) *ext_slot(a.ext, ge, c, N) = accbuf[N + c];
                if (side == 0 && lane == 0) *ext_slot(a.ext, ge, N, N) = accbuf[0];
                for (int c = side * 64 + lane; c <= n; c += 128) io.out[c] = 0u;
            }
            continue;
        }
        // identity key switch (tlwe.rs:43-73): each side sums the rows of half of the coefficients
        uint4 sum[KSQ];
        ks_accumulate<LOGN, KS_T, KS_BB, KSQ>(accbuf + N, side * (N / 2), (side + 1) * (N / 2), a.ksk, a.ksw, sum, lane);
        uint4* part = reinterpret_cast<uint4*>(xb1) + lane;   // [KSQ][64] uint4
        if (side == 1) {
#pragma unroll
            for (int q = 0; q < KSQ; q++) part[q * 64] = sum[q];
        }
        RR_PAIR_SYNC();
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
        RR_PAIR_SYNC();      // side 0 has read the partial sums: side 1 may write the next gate's
    }
#undef RR_PAIR_SYNC
}

}  // namespace rtfhe
