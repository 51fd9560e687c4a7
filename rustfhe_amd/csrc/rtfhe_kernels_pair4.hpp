// rtfhe_kernels_pair4.hpp -- N = 1024 with FOUR waves per gate: (polynomial, parity of the point index).  k_bootstrap_eo4 (rtfhe_kernels_eo4.hpp)
// one size down, for batches and tails of more than one and up to three gates per CU, where k_bootstrap_pair leaves SIMDs one wave to issue from.
// (Three gates per workgroup: 168 registers per wave, the parity tables read pass by pass from LDS; four -- 128 registers, 22 spilled -- lose
// 12 % against k_bootstrap_pair's full round.)
//
//   wave (side 0, parity H), owns the b-poly's points of parity H     wave (side 1, parity H), owns the a-poly's points of parity H
//   gather / decompose, rows 0..2 forward, trades with (0, 1 - H)     gather / decompose, rows 3..5 forward, trades with (1, 1 - H)
//   P: s0 = 0 + rows 0..2 of component 0     -> hand0
//   ------------------------------------------ hand-off 1 (the two sides of a parity) ------------------------------------------
//   Q: s1 = 0 + rows 0..2 of component 1     -> hand1                Q: s0 = hand0 + rows 3..5 of component 0   -> hand0
//   ------------------------------------------ hand-off 2 ----------------------------------------------------------------------
//   s0 = hand0; inverse transform (trade with (0, 1 - H)), += b-poly  R: s1 = hand1 + rows 3..5 of component 1; inverse, += a-poly
//
// A wave runs one parity of a 512-point transform: the 256-point sub-network of rtfhe_sub256.hpp (4 points per lane, its 15 twiddles per direction
// resident in registers, its exchanges as 8-byte planes: at two waves per SIMD they beat the 16-byte form the latency kernel uses, 4.05 -> 3.89 ms), then the size-2 stage across the parities as a HALF-WIDTH trade -- the even wave finishes both outputs of the butterflies
// k = 4 v + m, m < 2, the odd wave m >= 2: it sends two complex values per lane and receives two -- so that both sides of a parity hold the same
// spectrum points in the same registers (register j < 2: point 2k, register 2 + j: point 2k + 1, k = 4 v + 2 H + j; key layout: k_bk_to_p4) and
// the partial sums travel lane to lane.  Buffers, flags, fold order, publishing of the accumulator words: as in k_bootstrap_eo4.
// Same arithmetic DAG as the reference, every product and sum rounded on its own: bit-identical to k_bootstrap_pair.
// MODE_EXTRACT / MODE_BLIND_ROTATE only: the fused key switch (MODE_GATE) stays on k_bootstrap_pair.
#pragma once

#include "rtfhe_kernels_pair.hpp"
#include "rtfhe_sub256.hpp"

// priorities at two gates per workgroup (the two sides of a parity share a SIMD): side 1 at 1, side 0 at 2 from the start of a step and at 0 from a
// point on -- 4 = its last forward trade (default), 5 = the end of the passes that use the exchange buffer, 6 = the end of its slot P; 0 = none;
// 1 = side 0 at 2 throughout.  Measured (profiles/r04/pair4_ab.log): 512 gates 4.34 (0) / 4.35 (1) / 4.06 (4) ms.
// (at three gates per workgroup, where the two sides of a parity do not always share a SIMD, the same schedule: 5.83 vs 6.04 ms per 768 gates without)

namespace rtfhe {

struct Pair4Args {
    BootstrapArgs b;       // b.tw: the staged table with the parity tables (Q4Tw) behind it; b.bk unused
    const cplx* p4bk;      // [n][2l rows][2 comp][2 waves][4][64]: k_bk_to_p4
};

struct Pair4Lds {
    static constexpr size_t XB = (size_t)Q4::XS * sizeof(cplx);       // one exchange buffer: 320 complex slots
    static constexpr size_t FLAGS = 32;                                // per gate: 4 trade counters + 4 hand-off counters
    __host__ __device__ static constexpr size_t abar_bytes(int npad) { return ((size_t)npad * 2 + 15) / 16 * 16; }
    __host__ __device__ static constexpr size_t gate_bytes(int npad) { return (size_t)2 * 1024 * 4 + abar_bytes(npad) + 4 * XB + FLAGS; }
    // three gates per workgroup (168 registers per wave): the parity tables are staged into LDS in front of the gates
    __host__ __device__ static constexpr size_t tw_bytes(int gates) { return gates >= 3 ? (size_t)Q4Tw::TOTAL * sizeof(cplx) : 0; }
    __host__ __device__ static constexpr size_t bytes(int gates, int npad) { return tw_bytes(gates) + (size_t)gates * gate_bytes(npad); }
};

// key spectra: device layout of the N = 1024 kernels ([n][row][comp][8][64]: lane v, register q <-> point (v << 3) | q) -> the layout the waves of
// k_bootstrap_pair4 hold their spectra in: wave H, register s, lane v <-> q = 4 H + 2 (s & 1) + (s >> 1)
__global__ __launch_bounds__(256) void k_bk_to_p4(const cplx* __restrict__ src, cplx* __restrict__ dst, size_t polys) {
    const size_t total = polys * 512;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t g = idx >> 9;
        const int k = (int)(idx & 511);                  // destination: (H, register, lane)
        const int H = k >> 8, s = (k >> 6) & 3, lane = k & 63;
        const int q = 4 * H + 2 * (s & 1) + (s >> 1);
        dst[idx] = src[g * 512 + (size_t)q * 64 + lane];
    }
}

template <int L, int BGBIT, int GATES>
__global__ __launch_bounds__(256 * GATES, 1) void k_bootstrap_pair4(const Pair4Args pa) {
    constexpr int LOGN = 10, N = 1024, P = 512, R = 4;
    typedef Geo<LOGN> G;
    constexpr uint32_t M = decomp_mask(L, BGBIT);
    static_assert(L == 3, "slots P / Q / R hold three digit rows each");
    const BootstrapArgs& a = pa.b;
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane0 = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slot = wave % GATES;
    const int q = wave / GATES;             // 0..3 = 2 side + parity
    const int side = q >> 1, H = q & 1;

    const int g_raw = blockIdx.x * GATES + slot;
    const int g = g_raw < a.count ? g_raw : a.count - 1;
    const GateIo io = gate_io(a, g);
    const bool live = g_raw < a.count && io.ok;

    constexpr bool TWLDS = GATES >= 3;
    if constexpr (TWLDS) {
        cplx* qt = reinterpret_cast<cplx*>(smem);
        for (int idx = tid; idx < Q4Tw::TOTAL; idx += 256 * GATES) qt[idx] = a.tw[G::TW_TOTAL + idx];
    }
    unsigned char* gbase = smem + Pair4Lds::tw_bytes(GATES) + (size_t)slot * Pair4Lds::gate_bytes(a.npad);
    uint32_t* accbuf = reinterpret_cast<uint32_t*>(gbase);                                  // [2][N]
    uint16_t* abar = reinterpret_cast<uint16_t*>(gbase + (size_t)2 * N * 4);
    cplx* xbase = reinterpret_cast<cplx*>(gbase + (size_t)2 * N * 4 + Pair4Lds::abar_bytes(a.npad));
    auto xb = [&](int s, int idx) { return xbase + (size_t)(s * 2 + idx) * Q4::XS; };      // the two buffers of side s
    int widx = H;                             // which of my side's buffers I own (write next); flips after every trade
    uint32_t* flags = reinterpret_cast<uint32_t*>(gbase + Pair4Lds::gate_bytes(a.npad) - Pair4Lds::FLAGS);
    if (lane0 == 0) { flags[q] = 0u; flags[4 + q] = 0u; }
    auto lds_addr = [](uint32_t* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) uint32_t*)p; };
    const unsigned my_flag = lds_addr(flags + q), partner_flag = lds_addr(flags + (q ^ 1));             // trades: the other parity of my side
    const unsigned my_hflag = lds_addr(flags + 4 + q), other_hflag = lds_addr(flags + 4 + (q ^ 2));     // hand-offs: the other side of my parity
    unsigned sync_k = 0, hand_k = 0;
#define P4_ARRIVE() pair_arrive(my_flag, ++sync_k)
#define P4_WAIT() pair_wait_opaque(partner_flag, sync_k)
#define P4_HANDOFF() pair_sync(my_hflag, other_hflag, ++hand_k)

    const int n = a.n;
    {   // pre-step + mod switch (tfhe.rs:41-71, 97, 107-108) to [0, 2N)
        constexpr int SH = 32 - LOGN - 1;
        for (int i = lane0 + 64 * q; i <= n; i += 256) {
            const uint32_t t = gate_linear(io.op, io.p0[i], io.p1[i], i == n);
            abar[i] = (uint16_t)((i == n) ? (t >> SH) : ((t + (1u << (SH - 1))) >> SH));
        }
    }
    __syncthreads();
    {   // acc = X^{-bbar} * testvec (tfhe.rs:85, 98-106); each wave initialises a quarter of the words
        const int bbar = (int)abar[n];
        for (int c = lane0 + 64 * q; c < 2 * N; c += 256) {
            const int e = (c + bbar) & (2 * N - 1);
            accbuf[c] = c < N ? ((e >> LOGN) ? 0xE0000000u : 0x20000000u) : 0u;
        }
    }
    __syncthreads();

    // this parity's twiddles, both directions: resident over the whole blind rotation (the parity tables ride behind the staged table)
    typedef typename std::conditional<TWLDS, Q4Lds, Q4Regs>::type QT;
    QT qf, qi;
    if constexpr (TWLDS) {
        qf = Q4Lds{reinterpret_cast<const cplx*>(smem) + Q4Tw::off(0, H), lane0};
        qi = Q4Lds{reinterpret_cast<const cplx*>(smem) + Q4Tw::off(1, H), lane0};
    } else {
        qf.load(a.tw + G::TW_TOTAL + Q4Tw::off(0, H), lane0);
        qi.load(a.tw + G::TW_TOTAL + Q4Tw::off(1, H), lane0);
    }

    // key rows rc = 2 row + comp of the step, this wave's half of the points (k_bk_to_p4); two buffers, refilled as a multiply-accumulate retires
    const size_t trgsw_cplx = (size_t)2 * L * 2 * 2 * R * 64;
    cplx bA[R], bB[R];
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t bk_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<cplx*>(pa.p4bk), 0, 0x7fffffff, 0x00020000);
    const int lane16 = lane0 * 16;
    auto fetch = [&](cplx (&dst)[R], int step, int rc) {
        const size_t row = (size_t)step * trgsw_cplx + (size_t)rc * 2 * R * 64 + (size_t)H * R * 64;
        const int s0 = __builtin_amdgcn_readfirstlane((int)(row * sizeof(cplx)));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < R; m++) {
            const v4u v = __builtin_amdgcn_raw_buffer_load_b128(bk_rsrc, lane16 + m * 1024, s0, 0);
            dst[m] = make_double2(__longlong_as_double(((unsigned long long)v.y << 32) | v.x), __longlong_as_double(((unsigned long long)v.w << 32) | v.z));
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    // the size-2 stage across the parities, half-width trades (see k_bootstrap_eo): two complex values per lane each way, 16-byte accesses
    auto cross_write = [&](auto odd, const double (&re)[R], const double (&im)[R], cplx* wb, int ln) {
        constexpr int SEND = decltype(odd)::value ? 0 : R / 2;
#pragma unroll
        for (int j = 0; j < R / 2; j++) lds_st128(&wb[ln + 64 * j], re[SEND + j], im[SEND + j]);
    };
    auto cross_read = [&](auto odd, double (&re)[R], double (&im)[R], const cplx* rb, int ln) {
#pragma unroll
        for (int j = 0; j < R / 2; j++) {
            const cplx p = lds_ld128(&rb[ln + 64 * j]);
            if constexpr (!decltype(odd)::value) {      // mine = out_E, partner's = out_O
                const double ar = re[j], ai = im[j];
                re[j] = ar + p.x; im[j] = ai + p.y; re[R / 2 + j] = ar + (-p.x); im[R / 2 + j] = ai + (-p.y);
            } else {                                    // partner's = out_E, mine = out_O
                const double br = re[R / 2 + j], bi = im[R / 2 + j];
                re[j] = p.x + br; im[j] = p.y + bi; re[R / 2 + j] = p.x + (-br); im[R / 2 + j] = p.y + (-bi);
            }
        }
    };
    auto inv_cross_write = [&](auto odd, double (&re)[R], double (&im)[R], cplx* wb, int ln) {
        constexpr int SEND = decltype(odd)::value ? 0 : R / 2;      // sums stay in registers j, differences in 2 + j; the partner's overwrite what was sent
#pragma unroll
        for (int j = 0; j < R / 2; j++) {
            const double ar = re[j], br = re[R / 2 + j], ai = im[j], bi = im[R / 2 + j];
            re[j] = ar + br; im[j] = ai + bi; re[R / 2 + j] = ar + (-br); im[R / 2 + j] = ai + (-bi);
        }
#pragma unroll
        for (int j = 0; j < R / 2; j++) lds_st128(&wb[ln + 64 * j], re[SEND + j], im[SEND + j]);
    };
    auto inv_cross_read = [&](auto odd, double (&re)[R], double (&im)[R], const cplx* rb, int ln) {
        constexpr int RECV = decltype(odd)::value ? 0 : R / 2;
#pragma unroll
        for (int j = 0; j < R / 2; j++) { const cplx p = lds_ld128(&rb[ln + 64 * j]); re[RECV + j] = p.x; im[RECV + j] = p.y; }
    };
    // a partial sum (4 complex values per lane) to / from a hand-off buffer
    auto put = [&](cplx* hb, const double (&re)[R], const double (&im)[R], int ln) {
#pragma unroll
        for (int m = 0; m < R; m++) lds_st128(&hb[ln + 64 * m], re[m], im[m]);
    };
    auto get = [&](const cplx* hb, double (&re)[R], double (&im)[R], int ln) {
#pragma unroll
        for (int m = 0; m < R; m++) { const cplx p = lds_ld128(&hb[ln + 64 * m]); re[m] = p.x; im[m] = p.y; }
    };

    // The step loop exists four times -- (side, parity) compile-time constants -- and is chosen once: straight-line code per wave.
    auto steps = [&](auto sidec, auto parity) {
    constexpr bool ODD = decltype(parity)::value;
    constexpr int SIDE = decltype(sidec)::value;
    uint32_t* poly = accbuf + SIDE * N;
    const int rc0 = SIDE * 2 * L;                 // rc = 2 * row + comp of this side's first row
#pragma unroll 1
    for (int i = 0; i < a.steps; i++) {
        const int r = __builtin_amdgcn_readfirstlane((int)abar[i]);
        if constexpr (GATES >= 2) __builtin_amdgcn_s_setprio(SIDE == 0 ? 2 : 1);
        cplx* wbuf = xb(SIDE, widx);              // the buffer I own (write next)
        cplx* rbuf = xb(SIDE, widx ^ 1);          // my parity partner's (read after its arrival)
        int ln = lane0;
        asm volatile("" : "+v"(ln));        // keeps the lane-derived LDS addresses from being hoisted out of the loop and spilled
        // this lane's 4 complex inputs are points i = 2 (ln + 64 m) + H: coefficients i (real part) and i + 512 (imaginary part)
        // (rotate: math.rs:85-132; decomposition: math.rs:300-326)
        uint32_t ure[R], uim[R];
        {
            const int e0 = (2 * ln + H - r) * 4;
            const unsigned char* pb = reinterpret_cast<const unsigned char*>(poly);
#pragma unroll
            for (int m = 0; m < R; m++) {
                const int c0 = 2 * (ln + 64 * m) + H, c1 = c0 + P;
                const int t0 = e0 + 512 * m, t1 = t0 + 4 * P;
                const uint32_t v0 = *reinterpret_cast<const uint32_t*>(pb + (t0 & (4 * N - 4)));
                const uint32_t v1 = *reinterpret_cast<const uint32_t*>(pb + (t1 & (4 * N - 4)));
                const uint32_t sg0 = (uint32_t)((int32_t)((uint32_t)t0 << (31 - LOGN - 2)) >> 31);     // all ones iff bit LOGN of (i - r) is set
                const uint32_t sg1 = (uint32_t)((int32_t)((uint32_t)t1 << (31 - LOGN - 2)) >> 31);
                ure[m] = ((((v0 ^ sg0) - sg0) - poly[c0]) + M) ^ M;
                uim[m] = ((((v1 ^ sg1) - sg1) - poly[c1]) + M) ^ M;
            }
        }
        double yr[L][R], yi[L][R];
#pragma unroll
        for (int jj = 0; jj < L; jj++)
#pragma unroll
            for (int m = 0; m < R; m++) {
                yr[jj][m] = (double)decomp_digit(ure[m], BGBIT, jj);
                yi[jj][m] = (double)decomp_digit(uim[m], BGBIT, jj);
            }
        fetch(bA, i, rc0);                          // (first row, component 0): in flight under the transforms
        // twist and the parity's sub-network of the three rows (spqlios-fft-impl.cpp:496-603): first the passes that need the exchange buffer, for all
        // rows; then row by row the last pass (registers only) with the row's trade behind it -- the NEXT row's last pass runs between the arrival
        // flag and the wait
        sub256_forward_a_multi<L, true>(yr, yi, qf, wbuf, ln);
        sub256_forward_b<ODD, BOOT_TRIV>(yr[0], yi[0], qf);
        cross_write(parity, yr[0], yi[0], wbuf, ln); P4_ARRIVE();
        sub256_forward_b<ODD, BOOT_TRIV>(yr[1], yi[1], qf);
        P4_WAIT(); cross_read(parity, yr[0], yi[0], rbuf, ln);
        cross_write(parity, yr[1], yi[1], rbuf, ln); P4_ARRIVE();
        sub256_forward_b<ODD, BOOT_TRIV>(yr[2], yi[2], qf);
        P4_WAIT(); cross_read(parity, yr[1], yi[1], wbuf, ln);
        cross_write(parity, yr[2], yi[2], wbuf, ln); P4_ARRIVE();
        if constexpr (GATES >= 2 && SIDE == 0) __builtin_amdgcn_s_setprio(0);
        fetch(bB, i, rc0 + 2);                      // (second row, component 0)
        P4_WAIT(); cross_read(parity, yr[2], yi[2], rbuf, ln);
        widx ^= 1;                                  // three trades: I now own the buffer I read last
        cplx* mine = xb(SIDE, widx);                // idle until my inverse: the hand-off buffer on my side
        cplx* theirs = xb(1 - SIDE, widx);          // ... and the one the other side of my parity owns (it made the same three trades)

        // hadamard + fold-add (spqlios.rs:204-222, trgsw.rs:290-299) over this side's three rows, one component at a time
        double sre[R], sim[R];
        if constexpr (SIDE == 0) {
#pragma unroll
            for (int m = 0; m < R; m++) { sre[m] = 0.0; sim[m] = 0.0; }
            mac_row<R>(sre, sim, bA, yr[0], yi[0]); fetch(bA, i, rc0 + 4);            // P: rows 0..2 of component 0
            mac_row<R>(sre, sim, bB, yr[1], yi[1]); fetch(bB, i, rc0 + 1);
            mac_row<R>(sre, sim, bA, yr[2], yi[2]); fetch(bA, i, rc0 + 3);
            put(mine, sre, sim, ln);                                                  // hand0
            P4_HANDOFF();
#pragma unroll
            for (int m = 0; m < R; m++) { sre[m] = 0.0; sim[m] = 0.0; }
            mac_row<R>(sre, sim, bB, yr[0], yi[0]); fetch(bB, i, rc0 + 5);            // Q: rows 0..2 of component 1
            mac_row<R>(sre, sim, bA, yr[1], yi[1]);
            mac_row<R>(sre, sim, bB, yr[2], yi[2]);
            put(theirs, sre, sim, ln);                                                // hand1 (side 1 has finished its transforms: hand-off 1)
            P4_HANDOFF();
            get(mine, sre, sim, ln);                                                  // component 0, rows 0..5
        } else {
            P4_HANDOFF();
            get(theirs, sre, sim, ln);                                                // hand0: component 0, rows 0..2
            mac_row<R>(sre, sim, bA, yr[0], yi[0]); fetch(bA, i, rc0 + 4);            // Q: + rows 3..5 of component 0
            mac_row<R>(sre, sim, bB, yr[1], yi[1]); fetch(bB, i, rc0 + 1);
            mac_row<R>(sre, sim, bA, yr[2], yi[2]); fetch(bA, i, rc0 + 3);
            put(theirs, sre, sim, ln);                                                // back into hand0
            P4_HANDOFF();
            get(mine, sre, sim, ln);                                                  // hand1: component 1, rows 0..2
            mac_row<R>(sre, sim, bB, yr[0], yi[0]); fetch(bB, i, rc0 + 5);            // R: + rows 3..5 of component 1
            mac_row<R>(sre, sim, bA, yr[1], yi[1]);
            mac_row<R>(sre, sim, bB, yr[2], yi[2]);
        }

        // inverse of component SIDE: the size-2 stage across the parities comes FIRST, then this parity's sub-network (untwist included), truncate, += acc
        {
            wbuf = mine; rbuf = xb(SIDE, widx ^ 1);
            int lane = lane0;
            asm volatile("" : "+v"(lane));
            inv_cross_write(parity, sre, sim, wbuf, lane); P4_ARRIVE();
            P4_WAIT(); inv_cross_read(parity, sre, sim, rbuf, lane);
            widx ^= 1;
            sub256_inverse<ODD, BOOT_TRIV, true>(sre, sim, qi, xb(SIDE, widx), lane);
#pragma unroll
            for (int m = 0; m < R; m++) {
                const int c = 2 * (lane + 64 * m) + H;
                poly[c] += trunc_to_torus(sre[m]);
                poly[c + P] += trunc_to_torus(sim[m]);
            }
            // my accumulator words reach the other parity of my side (its next gather reads them) with this arrival
            P4_ARRIVE(); P4_WAIT();
        }
    }
    };
    if (side) { if (H) steps(std::integral_constant<int, 1>{}, std::true_type{}); else steps(std::integral_constant<int, 1>{}, std::false_type{}); }
    else      { if (H) steps(std::integral_constant<int, 0>{}, std::true_type{}); else steps(std::integral_constant<int, 0>{}, std::false_type{}); }
    __syncthreads();

    if (a.mode == MODE_BLIND_ROTATE) {
        if (live) {
            uint32_t* o = a.out + (size_t)g * 2 * N;
            for (int c = lane0 + 64 * q; c < 2 * N; c += 256) o[c] = accbuf[c];
        }
        return;
    }
    // sample extract index 0 (trlwe.rs:110-121): a'_0 = a_0, a'_k = -a_{N-k}; b' = b_0
    {
        uint32_t av[R];
#pragma unroll
        for (int mm = 0; mm < R; mm++) av[mm] = accbuf[N + lane0 + 64 * mm + 256 * q];
        __syncthreads();
#pragma unroll
        for (int mm = 0; mm < R; mm++) {
            const int c = lane0 + 64 * mm + 256 * q;
            accbuf[N + ((N - c) & (N - 1))] = (c == 0) ? av[mm] : (0u - av[mm]);
        }
    }
    __syncthreads();
    // MODE_EXTRACT: the key switch of the whole batch follows as its own launch (k_key_switch_mm)
    if (live) {
        const int ge = a.ext_first + g;      // batch-wide gate number: the sample buffer is laid out for the key switch (ext_slot)
        for (int c = q * (N / 4) + lane0; c < (q + 1) * (N / 4); c += 64) *ext_slot(a.ext, ge, c, N) = accbuf[N + c];
        if (q == 0 && lane0 == 0) *ext_slot(a.ext, ge, N, N) = accbuf[0];
        for (int c = q * 64 + lane0; c <= n; c += 256) io.out[c] = 0u;
    }
}

}  // namespace rtfhe
