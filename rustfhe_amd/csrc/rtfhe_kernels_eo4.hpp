// rtfhe_kernels_eo4.hpp -- N = 2048 with FOUR waves per gate: (polynomial, parity of the point index).  The shape for batches of up to two gates
// per CU, where k_bootstrap_eo leaves a SIMD one wave (or none) to issue from.
//
// k_bootstrap_eo (rtfhe_kernels_eo.hpp) splits every 1024-point transform over two waves by the parity of the point index; each of the two
// waves of a gate then runs all six forward rows and both inverse transforms of a CMUX step: 4.7 k instructions per wave and step, on ONE wave per
// SIMD when a CU holds one or two gates.  Here the step is also split the way k_bootstrap_pair (rtfhe_kernels_pair.hpp) splits it at N = 1024:
//
//   wave (side 0, parity H), owns the b-poly's points of parity H     wave (side 1, parity H), owns the a-poly's points of parity H
//   gather / decompose, rows 0..2 forward, trades with (0, 1 - H)     gather / decompose, rows 3..5 forward, trades with (1, 1 - H)
//   P: s0 = 0 + rows 0..2 of component 0     -> hand0
//   ------------------------------------------ hand-off 1 (the two sides of a parity) ------------------------------------------
//   Q: s1 = 0 + rows 0..2 of component 1     -> hand1                Q: s0 = hand0 + rows 3..5 of component 0   -> hand0
//   ------------------------------------------ hand-off 2 ----------------------------------------------------------------------
//   s0 = hand0; inverse transform (trade with (0, 1 - H)), += b-poly  R: s1 = hand1 + rows 3..5 of component 1; inverse, += a-poly
//
// * Every accumulator point sums rows 0, 1, ..., 5 from +0.0 in order (trgsw.rs:290-299): partial sums travel, products are never re-associated.
//   A wave holds BOTH outputs of half of the crossing butterflies (k_bootstrap_eo's half-width trade): side 0 and side 1 of a parity hold the same
//   spectrum points in the same registers, which is what the hand-offs need; the key is read in k_bk_to_eo's layout.
// * hand0 / hand1 are the exchange buffer a side's wave owns at that moment (ping-pong ownership between the parities, as in k_bootstrap_eo): idle
//   between a wave's last forward trade and its inverse.  Both sides of a parity make the same number of trades, so each knows which of the
//   other side's two buffers that is.
// * A side only ever reads and writes its OWN polynomial; its two parities publish their accumulator words to each other with one more
//   arrival / wait pair at the end of a step (k_bootstrap_eo gets that from the other component's trade).
// Same arithmetic DAG as the reference, every product and sum rounded on its own: bit-identical to k_bootstrap_eo and the one-wave kernel.
// MODE_EXTRACT / MODE_BLIND_ROTATE only: the fused key switch (MODE_GATE: RTFHE_KS_MM_MIN=0, foreign stream captures) stays on k_bootstrap_eo.
#pragma once

#include "rtfhe_kernels_eo.hpp"

// priorities at two gates per workgroup, where the two sides of a parity share a SIMD: side 1 at 1, side 0 at 2 from the start of a step and at 0 from
// a point on -- 1 = the end of its slot P, 2 = of its slot Q, 3 = its second exchange (k_bootstrap_pair's schedule), 4 = its last forward trade; 0 = no
// priorities.  Measured (profiles/r04/n2048_four_waves_priority_ab.log): 512 gates 8.12 (0) / 7.70-7.86 (1) / 7.83 (2) / 7.91 (3) / 7.84 (4) ms, 300
// gates 7.91 / 7.25-7.29 / 7.29 / 7.42 / 7.11 ms.
// (The wave-private exchanges as one 16-byte LDS access per complex value at one gate per workgroup, every wave alone on its SIMD: 5.82 -> 5.96 ms.)

namespace rtfhe {

struct Eo4Lds {
    typedef Geo<10> G;
    static constexpr size_t TW = EoLds::TW;
    static constexpr size_t XB = EoLds::XB;
    static constexpr size_t FLAGS = 32;       // per gate: 4 trade counters + 4 hand-off counters
    __host__ __device__ static constexpr size_t gate_bytes(int npad) { return (size_t)2 * 2048 * 4 + EoLds::abar_bytes(npad) + 4 * XB + FLAGS; }
    __host__ __device__ static constexpr size_t bytes(int gates, int npad) { return TW + (size_t)gates * gate_bytes(npad); }
};

template <int L, int BGBIT, int GATES>
__global__ __launch_bounds__(256 * GATES, 1) void k_bootstrap_eo4(const EoArgs ea) {
    constexpr int LOGN = 11, N = 2048, P = 1024, R = 8, NT = 256 * GATES;
    typedef Geo<10> G;   // geometry of a parity's 512-point sub-network
    constexpr uint32_t M = decomp_mask(L, BGBIT);
    static_assert(L == 3, "three digit rows of a polynomial are transformed side by side");
    const BootstrapArgs& a = ea.b;
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane0 = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slot = wave % GATES;
    const int q = wave / GATES;             // 0..3 = 2 side + parity
    const int side = q >> 1, H = q & 1;
    cplx* tw = reinterpret_cast<cplx*>(smem);
    for (int idx = tid; idx < EoTw::LDS_CPLX; idx += NT) tw[idx] = ea.etw[EoTw::P1 + idx];
    const cplx* tw_fwd12 = tw + (size_t)H * 7 * 64 - G::TW_P1;
    const cplx* tw_p2 = tw + (EoTw::P2 - EoTw::P1) + (size_t)H * 7 * 8;
    const cplx* tw_p3 = tw + (EoTw::P3 - EoTw::P1) + (size_t)H * 8;
    const cplx* twi_p2 = tw + (EoTw::IP2 - EoTw::P1) + (size_t)H * 7 * 8;
    const cplx* twi_p3 = tw + (EoTw::IP3 - EoTw::P1) + (size_t)H * 8;
    const cplx* gtwist0 = ea.etw + EoTw::TWIST + (size_t)H * 8 * 64;     // global memory
    const cplx* guntw0 = ea.etw + EoTw::IUNTW + (size_t)H * 8 * 64;
    const cplx* gip10 = ea.etw + EoTw::IP1 + (size_t)H * 7 * 64;

    const int g_raw = blockIdx.x * GATES + slot;
    const int g = g_raw < a.count ? g_raw : a.count - 1;
    const GateIo io = gate_io(a, g);
    const bool live = g_raw < a.count && io.ok;

    unsigned char* gbase = smem + Eo4Lds::TW + (size_t)slot * Eo4Lds::gate_bytes(a.npad);
    uint32_t* accbuf = reinterpret_cast<uint32_t*>(gbase);                                  // [2][N]
    uint16_t* abar = reinterpret_cast<uint16_t*>(gbase + (size_t)2 * N * 4);
    double* xbase = reinterpret_cast<double*>(gbase + (size_t)2 * N * 4 + EoLds::abar_bytes(a.npad));
    auto xb = [&](int s, int idx) { return xbase + (size_t)(s * 2 + idx) * 2 * G::XSLOTS; };      // the two buffers of side s
    int widx = H;                             // which of my side's buffers I own (write next); flips after every trade
    uint32_t* flags = reinterpret_cast<uint32_t*>(gbase + Eo4Lds::gate_bytes(a.npad) - Eo4Lds::FLAGS);
    if (lane0 == 0) { flags[q] = 0u; flags[4 + q] = 0u; }
    auto lds_addr = [](uint32_t* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) uint32_t*)p; };
    const unsigned my_flag = lds_addr(flags + q), partner_flag = lds_addr(flags + (q ^ 1));             // trades: the other parity of my side
    const unsigned my_hflag = lds_addr(flags + 4 + q), other_hflag = lds_addr(flags + 4 + (q ^ 2));     // hand-offs: the other side of my parity
    unsigned sync_k = 0, hand_k = 0;
#define EO4_ARRIVE() pair_arrive(my_flag, ++sync_k)
#define EO4_WAIT() pair_wait_opaque(partner_flag, sync_k)
#define EO4_HANDOFF() pair_sync(my_hflag, other_hflag, ++hand_k)

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

    // key rows rc = 2 row + comp of the step, this wave's half of the points (k_bk_to_eo); two buffers, refilled as a multiply-accumulate retires
    const size_t trgsw_cplx = (size_t)2 * L * 2 * 2 * R * 64;
    cplx bA[R], bB[R];
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t bk_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<cplx*>(ea.ebk), 0, 0x7fffffff, 0x00020000);
    const int lane16 = lane0 * 16;
    auto fetch = [&](cplx (&dst)[R], int step, int rc) {
        const size_t row = (size_t)step * trgsw_cplx + (size_t)rc * 2 * R * 64 + (size_t)H * R * 64;
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

    // the size-2 stage across the parities, half-width trades: see k_bootstrap_eo
    auto cross_write = [&](auto odd, const double (&re)[R], const double (&im)[R], double* wb, int ln) {
        constexpr int SEND = decltype(odd)::value ? 0 : R / 2;
#pragma unroll
        for (int j = 0; j < R / 2; j++) { lds_st(&wb[ln + 64 * j], re[SEND + j]); lds_st(&wb[G::XSLOTS + ln + 64 * j], im[SEND + j]); }
    };
    auto cross_read = [&](auto odd, double (&re)[R], double (&im)[R], const double* rb, int ln) {
#pragma unroll
        for (int j = 0; j < R / 2; j++) {
            const double pr = lds_ld(&rb[ln + 64 * j]), pi = lds_ld(&rb[G::XSLOTS + ln + 64 * j]);
            if constexpr (!decltype(odd)::value) {      // mine = out_E, partner's = out_O
                const double ar = re[j], ai = im[j];
                re[j] = ar + pr; im[j] = ai + pi; re[R / 2 + j] = ar + (-pr); im[R / 2 + j] = ai + (-pi);
            } else {                                    // partner's = out_E, mine = out_O
                const double br = re[R / 2 + j], bi = im[R / 2 + j];
                re[j] = pr + br; im[j] = pi + bi; re[R / 2 + j] = pr + (-br); im[R / 2 + j] = pi + (-bi);
            }
        }
    };
    auto inv_cross_write = [&](auto odd, double (&re)[R], double (&im)[R], double* wb, int ln) {
        constexpr int SEND = decltype(odd)::value ? 0 : R / 2;      // sums stay in registers j, differences in 4 + j; the partner's overwrite what was sent
#pragma unroll
        for (int j = 0; j < R / 2; j++) {
            const double ar = re[j], br = re[R / 2 + j], ai = im[j], bi = im[R / 2 + j];
            re[j] = ar + br; im[j] = ai + bi; re[R / 2 + j] = ar + (-br); im[R / 2 + j] = ai + (-bi);
        }
#pragma unroll
        for (int j = 0; j < R / 2; j++) { lds_st(&wb[ln + 64 * j], re[SEND + j]); lds_st(&wb[G::XSLOTS + ln + 64 * j], im[SEND + j]); }
    };
    auto inv_cross_read = [&](auto odd, double (&re)[R], double (&im)[R], const double* rb, int ln) {
        constexpr int RECV = decltype(odd)::value ? 0 : R / 2;
#pragma unroll
        for (int j = 0; j < R / 2; j++) { re[RECV + j] = lds_ld(&rb[ln + 64 * j]); im[RECV + j] = lds_ld(&rb[G::XSLOTS + ln + 64 * j]); }
    };
    // a partial sum (8 complex values per lane) to / from a hand-off buffer
    auto put = [&](double* hb, const double (&re)[R], const double (&im)[R], int ln) {
#pragma unroll
        for (int m = 0; m < R; m++) { lds_st(&hb[ln + 64 * m], re[m]); lds_st(&hb[G::XSLOTS + ln + 64 * m], im[m]); }
    };
    auto get = [&](const double* hb, double (&re)[R], double (&im)[R], int ln) {
#pragma unroll
        for (int m = 0; m < R; m++) { re[m] = lds_ld(&hb[ln + 64 * m]); im[m] = lds_ld(&hb[G::XSLOTS + ln + 64 * m]); }
    };

    // The step loop exists four times -- (side, parity) compile-time constants -- and is chosen once: straight-line code per wave (see k_bootstrap_eo).
    auto steps = [&](auto sidec, auto parity) {
    constexpr bool ODD = decltype(parity)::value;
    constexpr int SIDE = decltype(sidec)::value;
    uint32_t* poly = accbuf + SIDE * N;
    const int rc0 = SIDE * 2 * L;                 // rc = 2 * row + comp of this side's first row
#pragma unroll 1
    for (int i = 0; i < a.steps; i++) {
        const int r = __builtin_amdgcn_readfirstlane((int)abar[i]);
        if constexpr (GATES == 2) __builtin_amdgcn_s_setprio(SIDE == 0 ? 2 : 1);
        double* wbuf = xb(SIDE, widx);            // the buffer I own (write next)
        double* rbuf = xb(SIDE, widx ^ 1);        // my parity partner's (read after its arrival)
        int ln = lane0;
        asm volatile("" : "+v"(ln));        // keeps the lane-derived LDS addresses from being hoisted out of the loop and spilled
        // this lane's 8 complex inputs are points i = 2 (ln + 64 m) + H: coefficients i (real part) and i + 1024 (imaginary part)
        // (rotate: math.rs:85-132; decomposition: math.rs:300-326; twist: spqlios-fft-impl.cpp:496-518)
        cplx tH[R];
#pragma unroll
        for (int m = 0; m < R; m++) tH[m] = gtwist0[m * 64 + ln];
        uint32_t ure[R], uim[R];
        {
            const int e0 = (2 * ln + H - r) * 4;
            const unsigned char* pb = reinterpret_cast<const unsigned char*>(poly);
#pragma unroll
            for (int m = 0; m < R; m++) {
                const int c0 = 2 * (ln + 64 * m) + H, c1 = c0 + 1024;
                const int t0 = e0 + 512 * m, t1 = t0 + 4096;
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
                const double a0 = (double)decomp_digit(ure[m], BGBIT, jj), b0 = (double)decomp_digit(uim[m], BGBIT, jj);
                const double rc = a0 * tH[m].x, ic = b0 * tH[m].x, rs = a0 * tH[m].y, is = b0 * tH[m].y;
                yr[jj][m] = rc - is; yi[jj][m] = ic + rs;
            }
        {   // passes 1 and 2 of the three rows side by side, both wave-private exchanges
            Tw<R - 1> w1;
            w1.load(tw_fwd12 + G::TW_P1 + ln, 64);
#pragma unroll
            for (int jj = 0; jj < L; jj++) {
                P12<R, G::LR - 1>::fwd(yr[jj], yi[jj], w1.w);
                exchange<10, 1, 2, true>(yr[jj], yi[jj], wbuf, ln);
            }
            Tw<R - 1> w2;
            w2.load(tw_p2 + (ln & (G::NLOW - 1)), G::NLOW);
#pragma unroll
            for (int jj = 0; jj < L; jj++) {
                P12<R, G::LR - 1>::fwd(yr[jj], yi[jj], w2.w);
                exchange<10, 2, 3, true>(yr[jj], yi[jj], wbuf, ln);
            }
        }
        fetch(bA, i, rc0);                          // (first row, component 0): in flight under pass 3 and the trades
        {   // pass 3 row by row; a row goes to the parity partner right behind it, the NEXT row's pass 3 runs between the arrival flag and the wait
            Tw<6> w3;
            w3.load(tw_p3, 1);
            eo_fwd_pass3<R, ODD>(yr[0], yi[0], w3.w);
            cross_write(parity, yr[0], yi[0], wbuf, ln); EO4_ARRIVE();
            eo_fwd_pass3<R, ODD>(yr[1], yi[1], w3.w);
            EO4_WAIT(); cross_read(parity, yr[0], yi[0], rbuf, ln);
            cross_write(parity, yr[1], yi[1], rbuf, ln); EO4_ARRIVE();
            eo_fwd_pass3<R, ODD>(yr[2], yi[2], w3.w);
            EO4_WAIT(); cross_read(parity, yr[1], yi[1], wbuf, ln);
            cross_write(parity, yr[2], yi[2], wbuf, ln); EO4_ARRIVE();
        }
        if constexpr (GATES == 2 && SIDE == 0) __builtin_amdgcn_s_setprio(0);
        fetch(bB, i, rc0 + 2);                      // (second row, component 0)
        EO4_WAIT(); cross_read(parity, yr[2], yi[2], rbuf, ln);
        widx ^= 1;                                  // three trades: I now own the buffer I read last
        double* mine = xb(SIDE, widx);              // idle until my inverse: the hand-off buffer on my side
        double* theirs = xb(1 - SIDE, widx);        // ... and the one the other side of my parity owns (it made the same three trades)

        // hadamard + fold-add (spqlios.rs:204-222, trgsw.rs:290-299) over this side's three rows, one component at a time
        double sre[R], sim[R];
        if constexpr (SIDE == 0) {
#pragma unroll
            for (int m = 0; m < R; m++) { sre[m] = 0.0; sim[m] = 0.0; }
            mac_row<R>(sre, sim, bA, yr[0], yi[0]); fetch(bA, i, rc0 + 4);            // P: rows 0..2 of component 0
            mac_row<R>(sre, sim, bB, yr[1], yi[1]); fetch(bB, i, rc0 + 1);
            mac_row<R>(sre, sim, bA, yr[2], yi[2]); fetch(bA, i, rc0 + 3);
            put(mine, sre, sim, ln);                                                  // hand0
            EO4_HANDOFF();
#pragma unroll
            for (int m = 0; m < R; m++) { sre[m] = 0.0; sim[m] = 0.0; }
            mac_row<R>(sre, sim, bB, yr[0], yi[0]); fetch(bB, i, rc0 + 5);            // Q: rows 0..2 of component 1
            mac_row<R>(sre, sim, bA, yr[1], yi[1]);
            mac_row<R>(sre, sim, bB, yr[2], yi[2]);
            put(theirs, sre, sim, ln);                                                // hand1 (side 1 has finished its transforms: hand-off 1)
            EO4_HANDOFF();
            get(mine, sre, sim, ln);                                                  // component 0, rows 0..5
        } else {
            EO4_HANDOFF();
            get(theirs, sre, sim, ln);                                                // hand0: component 0, rows 0..2
            mac_row<R>(sre, sim, bA, yr[0], yi[0]); fetch(bA, i, rc0 + 4);            // Q: + rows 3..5 of component 0
            mac_row<R>(sre, sim, bB, yr[1], yi[1]); fetch(bB, i, rc0 + 1);
            mac_row<R>(sre, sim, bA, yr[2], yi[2]); fetch(bA, i, rc0 + 3);
            put(theirs, sre, sim, ln);                                                // back into hand0
            EO4_HANDOFF();
            get(mine, sre, sim, ln);                                                  // hand1: component 1, rows 0..2
            mac_row<R>(sre, sim, bB, yr[0], yi[0]); fetch(bB, i, rc0 + 5);            // R: + rows 3..5 of component 1
            mac_row<R>(sre, sim, bA, yr[1], yi[1]);
            mac_row<R>(sre, sim, bB, yr[2], yi[2]);
        }

        // inverse of component SIDE: the size-2 stage across the parities comes FIRST, then this parity's sub-network, untwist, truncate, += acc
        {
            wbuf = mine; rbuf = xb(SIDE, widx ^ 1);
            int lane = lane0;
            asm volatile("" : "+v"(lane));
            inv_cross_write(parity, sre, sim, wbuf, lane); EO4_ARRIVE();
            Tw<6> w3; Tw<R - 1> w2, w1; Tw<R> wt;
            w1.load(gip10 + lane, 64);                  // global memory: requested first, used last
#pragma unroll
            for (int m = 0; m < R; m++) wt.w[m] = guntw0[m * 64 + lane];
            w3.load(twi_p3, 1);
            w2.load(twi_p2 + (lane & (G::NLOW - 1)), G::NLOW);
            EO4_WAIT(); inv_cross_read(parity, sre, sim, rbuf, lane);
            widx ^= 1;
            wbuf = xb(SIDE, widx);
            eo_inv_pass3<R, ODD>(sre, sim, w3.w);
            exchange<10, 3, 2, true>(sre, sim, wbuf, lane);
            P12<R, G::LR - 1>::inv(sre, sim, w2.w);
            exchange<10, 2, 1, true>(sre, sim, wbuf, lane);
            P12<R, G::LR - 1>::inv(sre, sim, w1.w);
#pragma unroll
            for (int m = 0; m < R; m++) {
                const double vr = sre[m], vi = sim[m];
                // (re, im) * (c, s): re c - im s, im c + re s   (spqlios-fft-impl.cpp:390-395); the 2/N of fft_processor_spqlios.cpp:158 is in the table
                const double rc = vr * wt.w[m].x, ic = vi * wt.w[m].x, rs = vr * wt.w[m].y, is = vi * wt.w[m].y;
                const int c = 2 * (lane + 64 * m) + H;
                __hip_atomic_fetch_add(&poly[c], trunc_to_torus(rc - is), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                __hip_atomic_fetch_add(&poly[c + P], trunc_to_torus(ic + rs), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
            // my accumulator words reach the other parity of my side (its next gather reads them) with this arrival
            EO4_ARRIVE(); EO4_WAIT();
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
        for (int mm = 0; mm < R; mm++) av[mm] = accbuf[N + lane0 + 64 * mm + 512 * q];
        __syncthreads();
#pragma unroll
        for (int mm = 0; mm < R; mm++) {
            const int c = lane0 + 64 * mm + 512 * q;
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
