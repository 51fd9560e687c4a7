// rtfhe_kernels_xfft2.hpp -- the split-FFT exact backend (rtfhe_xfft.hpp) at N = 2048: the bootstrap kernel with FOUR waves per gate and the key
// transform.  (Model, proof of the bound 2^-6.6 < 1/2 and the agreement of this decomposition with the whole transform: scripts/xfft/model.py.)
//
// A polynomial of N = 2048 coefficients folds into n = 1024 complex points.  Exact arithmetic has no butterfly network to mirror, so the
// 1024-point transform is cut where it is cheapest for two waves to share it:
//   forward (natural order in, bit-reversed out, ring C[X]/(X^1024 - i)): stage 1 pairs z_j with z_{j+512} under ONE block twiddle
//     c = exp(i pi/4); the sums continue in the ring X^512 - c, the differences in X^512 + c -- two independent 512-point transforms of exactly the
//     N = 1024 backend's shape, with root angles pi/4 and pi/4 + pi.  Wave h (0 / 1) forms its 512 stage-1 results straight from the digits (both
//     waves gather and decompose the whole polynomial: integers, no exchange) and owns positions [512 h, 512 h + 512) of the spectrum;
//   multiply-accumulate: pointwise, so every wave works on its own half of every spectrum; the key is stored per (side, half);
//   inverse (radix-2 DIT on the bit-reversed spectrum): nine stages inside each half, the tenth across the halves fused with the untwist and the
//     rounding:  y_q = T_q U_q + B_q V_q,  y_{q+512} = (T_q U_q - B_q V_q) e^{-i pi/4}  (T / B: the halves' results, U_q = psi^-q / n,
//     V_q = omega^-q U_q).  The two waves of a polynomial split this last stage by q: wave 0 takes q = lane + 64 m for m < 4, wave 1 for m >= 4, so
//     each sends half of its results to its sibling and receives half (4 + 4 complex values per lane for the hi and lo sums: one 8 KiB trade).
// Wave (side, h): side 0 owns the b-polynomial, side 1 the a-polynomial (as k_bootstrap_pair / k_bootstrap_xpair); its PARTNER is (1 - side, h)
// -- the partial sums of the partner's output polynomial travel between them exactly as in k_bootstrap_xpair, half by half -- and its SIBLING
// is (side, 1 - h).  Every wave publishes ONE arrival counter in LDS, four arrivals per step:
//   4 i + 1  hand-off 1 (hi partials written)          waited for by the partner
//   4 i + 2  hand-off 2 (lo partials written)          waited for by the partner
//   4 i + 3  its share of the last inverse stage written to its own exchange buffer       waited for by the sibling
//   4 i + 4  its update of the polynomial done (and its reads of the sibling's buffer)    waited for by the sibling before the next gather
// Nothing in the step loop waits for another gate (rtfhe_kernels_xfft.hpp: lock step loses).
#pragma once

#include "rtfhe_kernels_xfft.hpp"

namespace rtfhe {
namespace xfft {

// device twiddle table at N = 2048, cplx units (host builder: xfft2_device_table, rtfhe_dispatch_xfft.hip)
struct XTw2 {
    static constexpr int FH = 8 + 7 * 8 + 7 * 64;      // one half's forward tables: F1 [7] (+ 1 pad), F2 [7][8], F3 [7][64] (as XTw)
    static constexpr int F = 0;                        // [2 halves][FH]
    static constexpr int I2 = 2 * FH;                  // [7][8]   inverse pass 2 (the standard DIT twiddles: both halves)
    static constexpr int I3 = I2 + 7 * 8;              // [7][64]
    static constexpr int UV = I3 + 7 * 64;             // [2 halves][2: U, V][4][64]: entry (k, lane) of half h <-> q = lane + 64 (4 h + k)
    static constexpr int TOTAL = UV + 2 * 2 * 4 * 64;
};

}  // namespace xfft

struct XQuadLds {
    typedef Geo<10> G;                                  // every wave runs 512-point transforms
    static constexpr int N = 2048;
    static constexpr size_t TW = (size_t)xfft::XTw2::TOTAL * sizeof(cplx);
    static constexpr size_t XB = (size_t)2 * G::XSLOTS * sizeof(double);            // one wave's re + im exchange buffers
    static constexpr size_t FLAGS = 16;                                             // four arrival counters
    __host__ __device__ static constexpr size_t gate_bytes(int npad) { return (size_t)2 * N * 4 + (size_t)npad * 4 + 4 * XB + FLAGS; }
    __host__ __device__ static constexpr size_t bytes(int gates, int npad) { return TW + (size_t)gates * gate_bytes(npad); }
};

template <int L, int BGBIT, int KS_T, int KS_BB, int KSQ, int GATES>
__global__ __launch_bounds__(256 * GATES, 1) void k_bootstrap_xquad(const XBootstrapArgs args) {
    constexpr int LOGN = 11;
    typedef Geo<10> G;
    typedef xfft::XTw2 T2;
    constexpr int N = 2048, P = 512, R = 8, NT = 256 * GATES;
    constexpr uint32_t M = decomp_mask(L, BGBIT);
    static_assert(L == 3 && R == xfft::R && G::P == P, "three rows per side are held in registers");
    const BootstrapArgs& a = args.b;
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = wave / GATES;                    // side * 2 + h: a wave and its partner (role ^ 2) share a SIMD at two gates per workgroup
    const int slot = wave % GATES;
    const int side = role >> 1, h = role & 1;
    cplx* tw = reinterpret_cast<cplx*>(smem);
    for (int idx = tid; idx < T2::TOTAL; idx += NT) tw[idx] = args.xtw[idx];
    const cplx* twf = tw + T2::F + h * T2::FH;        // this half's forward tables (laid out as XTw's F1 / F2 / F3)
    cplx w1[7];       // forward pass 1: wave-uniform twiddles (scalar loads)
#pragma unroll
    for (int e = 0; e < 7; e++) w1[e] = args.xtw[T2::F + h * T2::FH + xfft::XTw::F1 + e];
    const double c_s = h ? -xfft::SQRT_HALF : xfft::SQRT_HALF;     // stage 1: +- c, c = (1 + i) / sqrt 2

    const int g_raw = blockIdx.x * GATES + slot;
    const int g = g_raw < a.count ? g_raw : a.count - 1;
    const GateIo io = gate_io(a, g);
    const bool live = g_raw < a.count && io.ok;      // idle / skipped gates still run every step and take part in the barriers of the prologue and epilogue

    unsigned char* gbase = smem + XQuadLds::TW + (size_t)slot * XQuadLds::gate_bytes(a.npad);
    uint32_t* accbuf = reinterpret_cast<uint32_t*>(gbase);
    uint32_t* abar = accbuf + 2 * N;
    unsigned char* xbase = gbase + (size_t)2 * N * 4 + (size_t)a.npad * 4;
    double* myx = reinterpret_cast<double*>(xbase + (size_t)role * XQuadLds::XB);
    uint32_t* flags = reinterpret_cast<uint32_t*>(gbase + XQuadLds::gate_bytes(a.npad) - XQuadLds::FLAGS);
    if (lane == 0) flags[role] = 0u;
    // LDS addresses of the arrival counters as scalars (rtfhe_xfft.hpp: flag_arrive)
    const unsigned my_flag = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) uint32_t*)(flags + role));
    const unsigned partner_flag = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) uint32_t*)(flags + (role ^ 2)));
    const unsigned sibling_flag = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) uint32_t*)(flags + (role ^ 1)));
    // the exchange buffers as complex arrays [..][64 lanes]: own, the partner's, the sibling's.  Their lane addresses are formed where they are used,
    // from a freshly laundered lane number: as loop-invariant vector registers they would live across the whole step, which peaks at exactly 256
    // (three such registers cost k_bootstrap_xpair 181 spills)
    cplx* const buf_mine = reinterpret_cast<cplx*>(myx);
    cplx* const buf_peer = reinterpret_cast<cplx*>(xbase + (size_t)(role ^ 2) * XQuadLds::XB);
    const cplx* const buf_sib = reinterpret_cast<const cplx*>(xbase + (size_t)(role ^ 1) * XQuadLds::XB);
    auto lane_now = [&] { int l = lane; asm volatile("" : "+v"(l)); return l; };
    uint32_t* poly = accbuf + side * N;

    const int n = a.n;
    {   // pre-step + mod switch (tfhe.rs:41-71, 97, 107-108)
        constexpr int SH = 32 - LOGN - 1;
        for (int i = lane + 64 * role; i <= n; i += 256) {
            const uint32_t t = gate_linear(io.op, io.p0[i], io.p1[i], i == n);
            abar[i] = (i == n) ? (t >> SH) : ((t + (1u << (SH - 1))) >> SH);
        }
    }
    __syncthreads();
    {   // acc = X^{-bbar} * testvec (tfhe.rs:85, 98-106): each wave fills the half of its polynomial it will go on updating first
        const int bbar = (int)abar[n];
#pragma unroll
        for (int mm = 0; mm < 2 * R; mm++) {
            const int c = lane + 64 * (mm + 16 * h);
            const int e = (c + bbar) & (2 * N - 1);
            poly[c] = side ? 0u : ((e >> LOGN) ? 0xE0000000u : 0x20000000u);
        }
    }
    __syncthreads();

    // key rows: as k_bootstrap_xpair (half rows of 4 points per lane in a ring of three register buffers), 12 rows per wave and step,
    // device layout [n][role 4][12 = phase 4 x row 3][8][64 lanes]
    cplx kb[3][R / 2];
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t bk_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<cplx*>(args.xbk), 0, 0x7fffffff, 0x00020000);
    const int lane16 = lane * 16;
    constexpr int ROW_BYTES = R * 64 * (int)sizeof(cplx), HALF_ROWS = 24;
    auto fetch = [&](cplx (&dst)[R / 2], int step, int hr) {
        const int s_off = __builtin_amdgcn_readfirstlane(((step * 4 + role) * 12) * ROW_BYTES + hr * (ROW_BYTES / 2));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < R / 2; m++) {
            const v4u v = __builtin_amdgcn_raw_buffer_load_b128(bk_rsrc, lane16 + m * 1024, s_off, 0);
            dst[m] = make_double2(__longlong_as_double(((unsigned long long)v.y << 32) | v.x), __longlong_as_double(((unsigned long long)v.w << 32) | v.z));
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    if (a.steps > 0) { fetch(kb[0], 0, 0); fetch(kb[1], 0, 1); fetch(kb[2], 0, 2); }

    // priorities as in k_bootstrap_xpair: the partner that shares this wave's SIMD trades the lead with it from point to point
    auto prio = [&](int k) {
        if (k & 1) asm volatile("s_cmp_lg_u32 %0, 0\n\ts_cbranch_scc1 1f\n\ts_setprio 0\n1:" ::"s"(side) : "scc");
        else asm volatile("s_cmp_lg_u32 %0, 0\n\ts_cbranch_scc1 1f\n\ts_setprio 2\n1:" ::"s"(side) : "scc");
    };
    if (side) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(2);
#ifdef RTFHE_WG_STAMPS
    unsigned long long tsum[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
    const unsigned long long t_loop0 = tprev, rt_loop0 = __builtin_amdgcn_s_memrealtime();
#endif
#define XQ_ARRIVE(k) xfft::flag_arrive(my_flag, 4u * (unsigned)i + (k))
#define XQ_WAIT(flag, k) xfft::flag_wait(flag, 4u * (unsigned)i + (k))
#pragma unroll 1
    for (int i = 0; i < a.steps; i++) {
        const int r = __builtin_amdgcn_readfirstlane((int)abar[i]);
        const int nxt = (i + 1 < a.steps) ? i + 1 : i;
        int ln = lane;
        asm volatile("" : "+v"(ln));      // (see k_bootstrap_pair: keeps lane-derived LDS addresses from being hoisted and spilled)
        // gather X^r acc - acc of the WHOLE polynomial (32 coefficients per lane: z_j and z_{j+512}, real and imaginary halves), decompose, and
        // form this half's stage-1 results row by row:  t = z_j +- c z_{j+512},  c z = s (zr - zi) + i s (zr + zi): the sums and differences of
        // the digits are exact small integers, one FMA each then lands them on z_j
        double xr[L][R], xi[L][R];
#pragma unroll
        for (int mg = 0; mg < 2; mg++) {          // four points per lane at a time: 16 coefficients live instead of 32
            uint32_t u[4][R / 2];                  // [e][k]: coefficient j + 512 e, j = lane + 64 (4 mg + k)
#pragma unroll
            for (int e = 0; e < 4; e++)
#pragma unroll
                for (int k = 0; k < R / 2; k++) {
                    const int c = ln + 64 * (8 * e + 4 * mg + k);
                    u[e][k] = ((rotated_coef<LOGN>(poly, c, r) - poly[c]) + M) ^ M;
                }
#pragma unroll
            for (int jj = 0; jj < L; jj++) {
#pragma unroll
                for (int k = 0; k < R / 2; k++) {
                    const int zr = decomp_digit(u[0][k], BGBIT, jj), zr2 = decomp_digit(u[1][k], BGBIT, jj);      // coefficients j, j + 512
                    const int zi = decomp_digit(u[2][k], BGBIT, jj), zi2 = decomp_digit(u[3][k], BGBIT, jj);      // j + 1024, j + 1536
                    xr[jj][4 * mg + k] = __builtin_fma(c_s, (double)(zr2 - zi2), (double)zr);
                    xi[jj][4 * mg + k] = __builtin_fma(c_s, (double)(zr2 + zi2), (double)zi);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        PAIR_STAMP(0);
        prio(1);
        xfft::forward_multi_t<L>(xr, xi, twf + xfft::XTw::F2, twf + xfft::XTw::F3, w1, myx, myx + G::XSLOTS, ln, [&](int k) { prio(1 + k); });
        PAIR_STAMP(1);
        prio(4);

        double sre[2][R], sim[2][R];       // [0]: hi sums, [1]: lo sums of the own output polynomial (this half of the spectrum); the partner's partials pass through [1]
        auto put = [&](cplx* hnd, const double (&pr)[R], const double (&pi)[R]) {
#pragma unroll
            for (int m = 0; m < R; m++) hnd[m * 64] = make_double2(pr[m], pi[m]);
        };
        auto get = [&](const cplx* hnd, double (&pr)[R], double (&pi)[R]) {
#pragma unroll
            for (int m = 0; m < R; m++) { const cplx v = hnd[m * 64]; pr[m] = v.x; pi[m] = v.y; }
        };
        // the four multiply-accumulate phases over this wave's 24 half rows: exactly k_bootstrap_xpair's
#pragma unroll
        for (int hr = 0; hr < HALF_ROWS; hr++) {
            const int phase = hr / 6, row = (hr % 6) / 2, hh = hr & 1;
            if (hr == 6) {
                put(buf_mine + lane_now(), sre[1], sim[1]);
                PAIR_STAMP(2);
                XQ_ARRIVE(1u); XQ_WAIT(partner_flag, 1u);
                PAIR_STAMP(3);
                prio(5);
                get(buf_peer + lane_now(), sre[0], sim[0]);
            }
            if (hr == 12) prio(6);
            if (hr == 18) {
                put(buf_peer + lane_now(), sre[1], sim[1]);
                PAIR_STAMP(4);
                XQ_ARRIVE(2u); XQ_WAIT(partner_flag, 2u);
                PAIR_STAMP(5);
                prio(7);
                get(buf_mine + lane_now(), sre[1], sim[1]);
            }
            const bool first = (phase == 0 || phase == 2) && row == 0;
            if (phase == 1) xfft::mac_half(sre[0], sim[0], kb[hr % 3], xr[row], xi[row], hh, first);
            else xfft::mac_half(sre[1], sim[1], kb[hr % 3], xr[row], xi[row], hh, first);
            fetch(kb[hr % 3], hr + 3 < HALF_ROWS ? i : nxt, (hr + 3) % HALF_ROWS);
        }

        PAIR_STAMP(6);
        prio(8);
        xfft::inverse_core<2>(sre, sim, tw + T2::I2, tw + T2::I3, myx, myx + G::XSLOTS, lane, [&](int k) { prio(8 + k); });
        PAIR_STAMP(7);
        prio(12);          // the trade and the last stage at side 0's high priority: 18.56 against 18.91 ms per 1,024 gates (profiles/r06/xquad_priorities.log)
        // the trade with the sibling: this wave keeps the points m in [4 h, 4 h + 4) of both sums and sends the other four
        {
            cplx* mine = buf_mine + lane_now();
#pragma unroll
            for (int s = 0; s < 2; s++)
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const double vr = h ? sre[s][k] : sre[s][4 + k], vi = h ? sim[s][k] : sim[s][4 + k];
                    mine[(s * 4 + k) * 64] = make_double2(vr, vi);
                }
        }
        PAIR_STAMP(8);
        XQ_ARRIVE(3u); XQ_WAIT(sibling_flag, 3u);
        PAIR_STAMP(9);
        {
            const int l2 = lane_now();
            const cplx* trade_theirs = buf_sib + l2;       // [2 sums][4][64] cplx
            cplx uu[4], vv[4];
#pragma unroll
            for (int k = 0; k < 4; k++) { uu[k] = tw[T2::UV + ((h * 2 + 0) * 4 + k) * 64 + l2]; vv[k] = tw[T2::UV + ((h * 2 + 1) * 4 + k) * 64 + l2]; }
            uint32_t add[4][4];       // [k][coefficient q, q + 512, q + 1024, q + 1536]
#pragma unroll
            for (int s = 0; s < 2; s++)
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const cplx o = trade_theirs[(s * 4 + k) * 64];
                    // T: the result of half 0, B: of half 1, at q = lane + 64 (4 h + k)
                    const double tr = h ? o.x : sre[s][k], ti = h ? o.y : sim[s][k];
                    const double br = h ? sre[s][4 + k] : o.x, bi = h ? sim[s][4 + k] : o.y;
                    const double ur = uu[k].x, ui = uu[k].y, wr = vv[k].x, wi = vv[k].y;
                    // y_q = T U + B V  (+ MAGIC: rounding by the addition, as untwist_round)
                    const double y0r = __builtin_fma(tr, ur, __builtin_fma(-ti, ui, __builtin_fma(br, wr, __builtin_fma(-bi, wi, xfft::MAGIC))));
                    const double y0i = __builtin_fma(tr, ui, __builtin_fma(ti, ur, __builtin_fma(br, wi, __builtin_fma(bi, wr, xfft::MAGIC))));
                    // D = T U - B V;  y_{q+512} = D e^{-i pi/4} = s (D.re + D.im) + i s (D.im - D.re)
                    const double dr = __builtin_fma(tr, ur, __builtin_fma(-ti, ui, __builtin_fma(-br, wr, bi * wi)));
                    const double di = __builtin_fma(tr, ui, __builtin_fma(ti, ur, __builtin_fma(-br, wi, -(bi * wr))));
                    const double y1r = __builtin_fma(xfft::SQRT_HALF, dr, __builtin_fma(xfft::SQRT_HALF, di, xfft::MAGIC));
                    const double y1i = __builtin_fma(xfft::SQRT_HALF, di, __builtin_fma(-xfft::SQRT_HALF, dr, xfft::MAGIC));
                    if (s == 0) {
                        add[k][0] = xfft::rounded_hi16(y0r); add[k][1] = xfft::rounded_hi16(y1r); add[k][2] = xfft::rounded_hi16(y0i); add[k][3] = xfft::rounded_hi16(y1i);
                    } else {
                        add[k][0] += xfft::rounded_u32(y0r); add[k][1] += xfft::rounded_u32(y1r); add[k][2] += xfft::rounded_u32(y0i); add[k][3] += xfft::rounded_u32(y1i);
                    }
                }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int q = l2 + 64 * (4 * h + k);
#pragma unroll
                for (int e = 0; e < 4; e++)
                    __hip_atomic_fetch_add(&poly[q + 512 * e], add[k][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
        }
        PAIR_STAMP(10);
        XQ_ARRIVE(4u); XQ_WAIT(sibling_flag, 4u);
        PAIR_STAMP(11);
        prio(13);
    }
    __builtin_amdgcn_s_setprio(0);
#ifdef RTFHE_WG_STAMPS
    if (a.dbg && blockIdx.x == 0 && lane == 0)
        for (int k = 0; k < 16; k++) a.dbg[wave * 16 + k] = tsum[k];
    if (a.dbg && blockIdx.x < 1024 && tid == 0) {
        a.dbg[128 + 4 * blockIdx.x] = t_loop0;
        a.dbg[128 + 4 * blockIdx.x + 1] = __builtin_amdgcn_s_memtime();
        a.dbg[128 + 4 * blockIdx.x + 2] = __builtin_amdgcn_s_memrealtime() - rt_loop0;      // 100 MHz
        a.dbg[128 + 4 * blockIdx.x + 3] = (unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));     // HW_REG_XCC_ID
    }
#endif
    __syncthreads();

    if (a.mode == MODE_BLIND_ROTATE) {
        if (live) {
            uint32_t* o = a.out + (size_t)g * 2 * N + side * N;
            for (int c = lane + 64 * 16 * h; c < N / 2 * (h + 1); c += 64) o[c] = poly[c];
        }
        return;
    }
    // sample extract index 0 (trlwe.rs:110-121) on the a-polynomial: a'[0] = a[0], a'[c] = -a[N - c].  Reversed in place, every source would have to
    // be read before any destination is written; the two waves of side 1 write the reversed polynomial into an idle exchange buffer instead
    // (N words fit in one), half each.
    uint32_t* rev = reinterpret_cast<uint32_t*>(xbase + (size_t)2 * XQuadLds::XB);      // role 2's buffer: free after the step loop
    static_assert((size_t)N * 4 <= XQuadLds::XB, "the reversed a-polynomial fits one exchange buffer");
    if (side == 1) {
#pragma unroll
        for (int mm = 0; mm < 2 * R; mm++) {
            const int c = lane + 64 * (mm + 16 * h);
            const uint32_t v = poly[c];
            rev[(N - c) & (N - 1)] = (c == 0) ? v : (0u - v);
        }
    }
    __syncthreads();
    const uint32_t* aprime = rev;
    if (a.mode == MODE_EXTRACT) {      // the key switch of the whole batch follows as its own launch (k_key_switch_mm)
        if (live) {
            const int ge = a.ext_first + g;
            for (int c = role * (N / 4) + lane; c < (role + 1) * (N / 4); c += 64) *ext_slot(a.ext, ge, c, N) = aprime[c];
            if (role == 0 && lane == 0) *ext_slot(a.ext, ge, N, N) = accbuf[0];
            for (int c = role * 64 + lane; c <= n; c += 256) io.out[c] = 0u;
        }
        return;
    }
    // identity key switch (tlwe.rs:43-73): each wave sums the rows of a quarter of the coefficients; partial sums meet in role 3's buffer
    uint4 sum[KSQ];
    ks_accumulate<LOGN, KS_T, KS_BB, KSQ>(aprime, role * (N / 4), (role + 1) * (N / 4), a.ksk, a.ksw, sum, lane);
    uint4* part = reinterpret_cast<uint4*>(xbase + (size_t)3 * XQuadLds::XB) + lane;   // [3 roles][KSQ][64] uint4
    static_assert((size_t)3 * KSQ * 64 * sizeof(uint4) <= XQuadLds::XB, "three partial sums in one exchange buffer");
    if (role != 0) {
#pragma unroll
        for (int q = 0; q < KSQ; q++) part[((role - 1) * KSQ + q) * 64] = sum[q];
    }
    __syncthreads();
    if (role == 0 && live) {
        const uint32_t bprime = accbuf[0];
#pragma unroll
        for (int q = 0; q < KSQ; q++) {
            uint32_t s[4] = {sum[q].x, sum[q].y, sum[q].z, sum[q].w};
#pragma unroll
            for (int w = 0; w < 3; w++) {
                const uint4 o = part[(w * KSQ + q) * 64];
                s[0] += o.x; s[1] += o.y; s[2] += o.z; s[3] += o.w;
            }
            const int col = 4 * (lane + 64 * q);
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (col + e <= n) io.out[col + e] = ((col + e == n) ? bprime : 0u) - s[e];
        }
    }
}

// ---- key rows -> split spectra, N = 2048: source polynomial g = (i, comp, row) -> for each key half (hi, lo) and each spectrum half h one row of
// 512 complex values at  ((i * 4 + side * 2 + h) * 12 + phase * 3 + local)  with side = row / l, local = row % l, phase = 2 * keyhalf + (comp == side)
__host__ __device__ inline size_t xbk2_row_index(size_t g, int rows, int keyhalf, int h) {
    const size_t i = g / (2 * (size_t)rows), rem = g % (2 * (size_t)rows);
    const int comp = (int)(rem / rows), row = (int)(rem % rows), l = rows / 2;
    const int side = row / l, local = row % l;
    const int phase = 2 * keyhalf + (comp == side ? 1 : 0);
    return ((i * 4 + (size_t)(side * 2 + h)) * (size_t)(4 * l)) + (size_t)phase * l + local;
}

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void k_xbk_build2(const XBkArgs a) {
    typedef Geo<10> G;
    typedef xfft::XTw2 T2;
    constexpr int N = 2048, P = 512, R = 8;
    extern __shared__ __align__(16) unsigned char smem[];
    cplx* tw = reinterpret_cast<cplx*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int idx = tid; idx < T2::TOTAL; idx += 64 * WAVES) tw[idx] = a.xtw[idx];
    __syncthreads();
    double* xbuf = reinterpret_cast<double*>(smem + (size_t)T2::TOTAL * sizeof(cplx)) + (size_t)wave * 2 * G::XSLOTS;
    // work item = (source polynomial, spectrum half)
    for (int item = blockIdx.x * WAVES + wave; item < 2 * a.count; item += gridDim.x * WAVES) {
        const int g = item >> 1, h = item & 1;
        const cplx* twf = tw + T2::F + h * T2::FH;
        cplx w1[7];
#pragma unroll
        for (int e = 0; e < 7; e++) w1[e] = twf[xfft::XTw::F1 + e];
        const double c_s = h ? -xfft::SQRT_HALF : xfft::SQRT_HALF;
        const int32_t* src = reinterpret_cast<const int32_t*>(a.bk_torus) + (size_t)g * N;
        double re[2][R], im[2][R];       // [0]: hi, [1]: lo
#pragma unroll
        for (int m = 0; m < R; m++) {
            int32_t hi[4], lo[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {                       // coefficients j, j + 512, j + 1024, j + 1536 (j = lane + 64 m)
                const int32_t k = src[lane + 64 * m + P * e];
                lo[e] = (int32_t)(int16_t)k;                    // lo in [-2^15, 2^15)
                hi[e] = (int32_t)(((int64_t)k - lo[e]) >> 16);  // hi in [-2^15, 2^15]
            }
            // t = z_j +- c z_{j+512}: (zr, zi) = coefficients (j, j + 1024), (zr2, zi2) = (j + 512, j + 1536)
            re[0][m] = __builtin_fma(c_s, (double)(hi[1] - hi[3]), (double)hi[0]); im[0][m] = __builtin_fma(c_s, (double)(hi[1] + hi[3]), (double)hi[2]);
            re[1][m] = __builtin_fma(c_s, (double)(lo[1] - lo[3]), (double)lo[0]); im[1][m] = __builtin_fma(c_s, (double)(lo[1] + lo[3]), (double)lo[2]);
        }
        xfft::forward_multi_t<2>(re, im, twf + xfft::XTw::F2, twf + xfft::XTw::F3, w1, xbuf, xbuf + G::XSLOTS, lane);
#pragma unroll
        for (int keyhalf = 0; keyhalf < 2; keyhalf++) {
            cplx* dst = a.xbk + xbk2_row_index((size_t)g, a.rows, keyhalf, h) * P + lane;
#pragma unroll
            for (int m = 0; m < R; m++) dst[m * 64] = make_double2(re[keyhalf][m], im[keyhalf][m]);
        }
    }
}

// ---- stage-level external product on this backend at N = 2048 (rtfhe_external_product_batch): one workgroup of TWO waves per sample, wave h owning
// half h of every spectrum.  Off the timed path: every digit row is transformed on its own and all four sums (2 output polynomials x hi / lo) are
// accumulated side by side; the last inverse stage is traded through LDS behind the workgroup's barrier.  The same device functions and the same
// arithmetic as k_bootstrap_xquad: what a mismatch in a whole gate is localised with.
template <int L, int BGBIT>
__global__ __launch_bounds__(128, 1) void k_external_product_xfft2(const XExtProdArgs a) {
    typedef Geo<10> G;
    typedef xfft::XTw2 T2;
    constexpr int N = 2048, P = 512, R = 8;
    constexpr uint32_t M = decomp_mask(L, BGBIT);
    extern __shared__ __align__(16) unsigned char smem[];
    cplx* tw = reinterpret_cast<cplx*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int h = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int idx = tid; idx < T2::TOTAL; idx += 128) tw[idx] = a.xtw[idx];
    __syncthreads();
    const cplx* twf = tw + T2::F + h * T2::FH;
    cplx w1[7];
#pragma unroll
    for (int e = 0; e < 7; e++) w1[e] = twf[xfft::XTw::F1 + e];
    const double c_s = h ? -xfft::SQRT_HALF : xfft::SQRT_HALF;
    double* xbuf = reinterpret_cast<double*>(smem + (size_t)T2::TOTAL * sizeof(cplx)) + (size_t)h * 2 * G::XSLOTS;
    cplx* trade = reinterpret_cast<cplx*>(smem + (size_t)T2::TOTAL * sizeof(cplx) + (size_t)2 * 2 * G::XSLOTS * sizeof(double));     // [2 waves][2 sums][4][64]
    const int g = blockIdx.x;        // grid = count
    const uint32_t* in = a.trlwe + (size_t)g * 2 * N;
    const cplx* bk_i = a.xbk + (size_t)a.bk_index[g] * (4 * 12 * P);
    double s[2][2][2][R];        // [comp][key half][re / im][R]: this half of the four spectra
#pragma unroll
    for (int comp = 0; comp < 2; comp++)
#pragma unroll
        for (int half = 0; half < 2; half++)
#pragma unroll
            for (int m = 0; m < R; m++) { s[comp][half][0][m] = 0.0; s[comp][half][1][m] = 0.0; }
#pragma unroll 1
    for (int side = 0; side < 2; side++) {
#pragma unroll 1
        for (int jj = 0; jj < L; jj++) {
            double xr[1][R], xi[1][R];
#pragma unroll
            for (int m = 0; m < R; m++) {
                int d[4];
#pragma unroll
                for (int e = 0; e < 4; e++) d[e] = decomp_digit((in[side * N + lane + 64 * m + P * e] + M) ^ M, BGBIT, jj);      // coefficients j, j + 512, j + 1024, j + 1536
                xr[0][m] = __builtin_fma(c_s, (double)(d[1] - d[3]), (double)d[0]);
                xi[0][m] = __builtin_fma(c_s, (double)(d[1] + d[3]), (double)d[2]);
            }
            xfft::forward_multi_t<1>(xr, xi, twf + xfft::XTw::F2, twf + xfft::XTw::F3, w1, xbuf, xbuf + G::XSLOTS, lane);
#pragma unroll
            for (int phase = 0; phase < 4; phase++) {
                const int half = phase >> 1, comp = (phase & 1) ? side : 1 - side;
                const cplx* row = bk_i + (size_t)(((side * 2 + h) * 12) + phase * L + jj) * P + lane;
                cplx b[R];
#pragma unroll
                for (int m = 0; m < R; m++) b[m] = row[m * 64];
                if (comp == 0) xfft::mac<false>(s[0][half][0], s[0][half][1], b, xr[0], xi[0]);
                else xfft::mac<false>(s[1][half][0], s[1][half][1], b, xr[0], xi[0]);
            }
        }
    }
    uint32_t* o = a.out + (size_t)g * 2 * N;
#pragma unroll 1
    for (int comp = 0; comp < 2; comp++) {
        double yr[2][R], yi[2][R];       // [0]: hi, [1]: lo
#pragma unroll
        for (int half = 0; half < 2; half++)
#pragma unroll
            for (int m = 0; m < R; m++) { yr[half][m] = comp ? s[1][half][0][m] : s[0][half][0][m]; yi[half][m] = comp ? s[1][half][1][m] : s[0][half][1][m]; }
        xfft::inverse_core<2>(yr, yi, tw + T2::I2, tw + T2::I3, xbuf, xbuf + G::XSLOTS, lane);
        __syncthreads();             // (the previous component's trade has been read by both waves)
#pragma unroll
        for (int sm = 0; sm < 2; sm++)
#pragma unroll
            for (int k = 0; k < 4; k++)
                trade[((h * 2 + sm) * 4 + k) * 64 + lane] = make_double2(h ? yr[sm][k] : yr[sm][4 + k], h ? yi[sm][k] : yi[sm][4 + k]);
        __syncthreads();
        uint32_t add[4][4];
#pragma unroll
        for (int sm = 0; sm < 2; sm++)
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const cplx oth = trade[(((1 - h) * 2 + sm) * 4 + k) * 64 + lane];
                const cplx uu = tw[T2::UV + ((h * 2 + 0) * 4 + k) * 64 + lane], vv = tw[T2::UV + ((h * 2 + 1) * 4 + k) * 64 + lane];
                const double tr = h ? oth.x : yr[sm][k], ti = h ? oth.y : yi[sm][k];
                const double br = h ? yr[sm][4 + k] : oth.x, bi = h ? yi[sm][4 + k] : oth.y;
                const double y0r = __builtin_fma(tr, uu.x, __builtin_fma(-ti, uu.y, __builtin_fma(br, vv.x, __builtin_fma(-bi, vv.y, xfft::MAGIC))));
                const double y0i = __builtin_fma(tr, uu.y, __builtin_fma(ti, uu.x, __builtin_fma(br, vv.y, __builtin_fma(bi, vv.x, xfft::MAGIC))));
                const double dr = __builtin_fma(tr, uu.x, __builtin_fma(-ti, uu.y, __builtin_fma(-br, vv.x, bi * vv.y)));
                const double di = __builtin_fma(tr, uu.y, __builtin_fma(ti, uu.x, __builtin_fma(-br, vv.y, -(bi * vv.x))));
                const double y1r = __builtin_fma(xfft::SQRT_HALF, dr, __builtin_fma(xfft::SQRT_HALF, di, xfft::MAGIC));
                const double y1i = __builtin_fma(xfft::SQRT_HALF, di, __builtin_fma(-xfft::SQRT_HALF, dr, xfft::MAGIC));
                if (sm == 0) {
                    add[k][0] = xfft::rounded_hi16(y0r); add[k][1] = xfft::rounded_hi16(y1r); add[k][2] = xfft::rounded_hi16(y0i); add[k][3] = xfft::rounded_hi16(y1i);
                } else {
                    add[k][0] += xfft::rounded_u32(y0r); add[k][1] += xfft::rounded_u32(y1r); add[k][2] += xfft::rounded_u32(y0i); add[k][3] += xfft::rounded_u32(y1i);
                }
            }
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int e = 0; e < 4; e++) o[comp * N + lane + 64 * (4 * h + k) + P * e] = add[k][e];
    }
}

}  // namespace rtfhe
