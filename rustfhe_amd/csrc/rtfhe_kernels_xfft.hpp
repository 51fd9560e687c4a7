// rtfhe_kernels_xfft.hpp -- kernels of the split-FFT exact backend (rtfhe_xfft.hpp), N = 1024: the bootstrap kernel with two waves per gate,
// the key transform and the stage-level external product.
//
// k_bootstrap_xpair keeps k_bootstrap_pair's skeleton (rtfhe_kernels_pair.hpp: side 0 owns the b-polynomial, side 1 the a-polynomial; gather and
// decomposition from the own polynomial only; accumulator in LDS; key rows through a buffer resource into a register ring; the same prologue and
// epilogue) and replaces the CMUX step's arithmetic:
//
//   both sides, the same code (exact sums have no order, so the split is symmetric):
//   gather / decompose the own polynomial, three forward transforms side by side                                  (spectra in VGPRs)
//   M1: hi-half partial of the PARTNER's output polynomial over the own three rows            -> own exchange buffer
//   ---------------------------------------------- hand-off 1 (the pair's arrival flags) ---------------------------
//   M2: S_hi = the partner's partial (read from ITS exchange buffer) + own three rows of the OWN output polynomial, hi half
//   M3: lo-half partial of the partner's polynomial                                           -> the PARTNER's exchange buffer (just read: free)
//   ---------------------------------------------- hand-off 2 ------------------------------------------------------
//   M4: S_lo = the partner's partial (read from the OWN exchange buffer) + own three rows, lo half
//   two inverse transforms side by side (hi, lo), untwist fused with the rounding; own polynomial += (hi << 16) + lo  (mod 2^32)
//
// Why the hand-off takes two phases: a side owes its partner two spectra (16 KiB); the idle exchange buffers hold one each (4 gates per CU
// leave no more LDS).  Every buffer has one writer and one reader per phase and DS instructions of a wave execute in order:
//   own buffer    : written by me (forward exchanges, M1) before hand-off 1; read by the partner (M2) and then written by the partner (M3)
//                   between the hand-offs; read by me (M4) and reused by my inverse transforms after hand-off 2.
// A hand-off is the two waves of ONE gate meeting through flags in LDS; nothing in the step loop ever waits for another gate (see XF_SYNC).
#pragma once

#include "rtfhe_kernels.hpp"
#include "rtfhe_kernels_pair.hpp"
#include "rtfhe_xfft.hpp"

namespace rtfhe {

struct XBootstrapArgs {
    BootstrapArgs b;          // tw and bk of `b` are unused here
    const cplx* xtw;          // [xfft::XTw::TOTAL]
    const cplx* xbk;          // key spectra, device layout [n][side 2][12 = phase 4 x row 3][8][64 lanes]; phase 0: (hi, partner's polynomial),
                              // 1: (hi, own), 2: (lo, partner's), 3: (lo, own) -- a side's rows in the order it consumes them
};

struct XPairLds {
    typedef Geo<10> G;
    static constexpr size_t TW = (size_t)xfft::XTw::TOTAL * sizeof(cplx);
    static constexpr size_t XB = (size_t)2 * G::XSLOTS * sizeof(double);            // one wave's re + im exchange buffers
    static_assert(XB >= (size_t)G::P * sizeof(cplx), "an exchange buffer pair must hold one spectrum");
    static constexpr size_t FLAGS = 16;
    __host__ __device__ static constexpr size_t gate_bytes(int npad) { return (size_t)2 * G::N * 4 + (size_t)npad * 4 + 2 * XB + FLAGS; }
    __host__ __device__ static constexpr size_t bytes(int gates, int npad) { return TW + (size_t)gates * gate_bytes(npad); }
};

template <int L, int BGBIT, int KS_T, int KS_BB, int KSQ, int GATES>
__global__ __launch_bounds__(128 * GATES, 1) void k_bootstrap_xpair(const XBootstrapArgs args) {
    constexpr int LOGN = 10;
    typedef Geo<LOGN> G;
    constexpr int N = G::N, P = G::P, R = G::R, NT = 128 * GATES;
    constexpr uint32_t M = decomp_mask(L, BGBIT);
    static_assert(L == 3 && R == xfft::R, "three rows per side are held in registers");
    const BootstrapArgs& a = args.b;
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int side = wave / GATES;
    const int slot = wave % GATES;
    cplx* tw = reinterpret_cast<cplx*>(smem);
    for (int idx = tid; idx < xfft::XTw::TOTAL; idx += NT) tw[idx] = args.xtw[idx];
    cplx w1[7];       // forward pass 1: wave-uniform twiddles (scalar loads)
#pragma unroll
    for (int e = 0; e < 7; e++) w1[e] = args.xtw[xfft::XTw::F1 + e];

    const int g_raw = blockIdx.x * GATES + slot;
    const int g = g_raw < a.count ? g_raw : a.count - 1;
    const GateIo io = gate_io(a, g);
    const bool live = g_raw < a.count && io.ok;      // idle / skipped pairs still run every step and take part in the barriers of the prologue and epilogue

    unsigned char* gbase = smem + XPairLds::TW + (size_t)slot * XPairLds::gate_bytes(a.npad);
    uint32_t* accbuf = reinterpret_cast<uint32_t*>(gbase);
    uint32_t* abar = accbuf + 2 * N;
    double* xb0 = reinterpret_cast<double*>(gbase + (size_t)2 * N * 4 + (size_t)a.npad * 4);
    double* xb1 = xb0 + 2 * G::XSLOTS;
    double* myx = side ? xb1 : xb0;
    uint32_t* flags = reinterpret_cast<uint32_t*>(gbase + XPairLds::gate_bytes(a.npad) - XPairLds::FLAGS);
    if (lane == 0) flags[side] = 0u;
    // LDS addresses of the flags as scalars (rtfhe_xfft.hpp: flag_arrive)
    const unsigned my_flag = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) uint32_t*)(flags + side));
    const unsigned partner_flag = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) uint32_t*)(flags + (1 - side)));
    cplx* hand_mine = reinterpret_cast<cplx*>(myx) + lane;                       // [R][64] cplx
    cplx* hand_peer = reinterpret_cast<cplx*>(side ? xb0 : xb1) + lane;
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
    {   // acc = X^{-bbar} * testvec (tfhe.rs:85, 98-106)
        const int bbar = (int)abar[n];
#pragma unroll
        for (int mm = 0; mm < 2 * R; mm++) {
            const int c = lane + 64 * mm;
            const int e = (c + bbar) & (2 * N - 1);
            poly[c] = side ? 0u : ((e >> LOGN) ? 0xE0000000u : 0x20000000u);
        }
    }
    wave_lds_sync();

    // key rows in consumption order rc = 0..11 of this side, as HALF rows hr = 2 rc + h (h: points 0..3 / 4..7 of the lane): a ring of THREE 4-point
    // buffers across steps (half row hr lives in buffer hr % 3; 24 half rows per step, so the assignment is the same in every step), each refilled
    // right after its multiply-accumulate retires.  Two whole-row buffers (64 registers) made the allocator spill at the peak of the step (three
    // spectra + two accumulators + the ring), and scratch traffic shares the vector-memory path the key rows are bound by.
    cplx kb[3][R / 2];
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t bk_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<cplx*>(args.xbk), 0, 0x7fffffff, 0x00020000);
    const int lane16 = lane * 16;
    constexpr int ROW_BYTES = R * 64 * (int)sizeof(cplx), HALF_ROWS = 24;
    auto fetch = [&](cplx (&dst)[R / 2], int step, int hr) {
        const int s_off = __builtin_amdgcn_readfirstlane(((step * 2 + side) * 12) * ROW_BYTES + hr * (ROW_BYTES / 2));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < R / 2; m++) {
            const v4u v = __builtin_amdgcn_raw_buffer_load_b128(bk_rsrc, lane16 + m * 1024, s_off, 0);
            dst[m] = make_double2(__longlong_as_double(((unsigned long long)v.y << 32) | v.x), __longlong_as_double(((unsigned long long)v.w << 32) | v.z));
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    if (a.steps > 0) { fetch(kb[0], 0, 0); fetch(kb[1], 0, 1); fetch(kb[2], 0, 2); }

    // Priority schedule.  Of two busy waves a SIMD runs one at nearly full speed (the older one unless s_setprio says otherwise) and the other on
    // the leftovers: with equal priorities one side reaches every hand-off thousands of cycles early and the SIMD then runs a single wave.  Side 1
    // stays at priority 1; side 0 alternates between 2 and 0 from one point to the next, so the two sides trade the lead every segment (without
    // s_setprio: 7.55 against 7.20 ms per 1,024 gates; what is left is side 1 waiting ~5 k of the step's 24.8 k cycles at the hand-offs:
    // profiles/r06/xfft_sync_forms.log, xfft_phase_stamps_flags_build.log).
    auto prio = [&](int k) {
        if (k & 1) asm volatile("s_cmp_lg_u32 %0, 0\n\ts_cbranch_scc1 1f\n\ts_setprio 0\n1:" ::"s"(side) : "scc");
        else asm volatile("s_cmp_lg_u32 %0, 0\n\ts_cbranch_scc1 1f\n\ts_setprio 2\n1:" ::"s"(side) : "scc");
    };
    if (side) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(2);
#ifdef RTFHE_WG_STAMPS
    unsigned long long tsum[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
    const unsigned long long t_loop0 = tprev, rt_loop0 = __builtin_amdgcn_s_memrealtime();
#endif
    // The two waves of a gate meet through their arrival flags (rtfhe_xfft.hpp: flag_arrive / flag_wait), never through the workgroup barrier, at
    // every number of gates per workgroup.  Gates held in lock step collide phase by phase on what a CU shares (LDS, the vector-memory path);
    // gates that drift apart spread that load.  Measured at 1,024 gates (profiles/r06/xfft_sync_forms.log): workgroup barrier 8.21-8.31 ms, pair
    // flags 7.20-7.43 ms, no synchronisation at all (wrong results, timing only) 7.30 ms.  Lock step does buy L1 hits -- the four waves of one side
    // read the same key rows: 254 M requests to L2 per launch under the barrier against 974 M with flags -- and loses all the same: flags plus a
    // counter that keeps one side's four waves together 8.13-8.47 ms, pairs of gates kept together (527 M requests) 7.65-7.79 ms.
#define XF_SYNC(k) do { xfft::flag_arrive(my_flag, 2u * (unsigned)i + (k)); xfft::flag_wait(partner_flag, 2u * (unsigned)i + (k)); } while (0)
#pragma unroll 1
    for (int i = 0; i < a.steps; i++) {
        const int r = __builtin_amdgcn_readfirstlane((int)abar[i]);
        const int nxt = (i + 1 < a.steps) ? i + 1 : i;
        int ln = lane;
        asm volatile("" : "+v"(ln));      // (see k_bootstrap_pair: keeps lane-derived LDS addresses from being hoisted and spilled)
        uint32_t u[2 * R];
#pragma unroll
        for (int mm = 0; mm < 2 * R; mm++) {
            const int c = ln + 64 * mm;
            u[mm] = ((rotated_coef<LOGN>(poly, c, r) - poly[c]) + M) ^ M;
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
        PAIR_STAMP(0);
        prio(1);
        xfft::forward_multi<L>(xr, xi, tw, w1, myx, myx + G::XSLOTS, ln, [&](int k) { prio(1 + k); });
        PAIR_STAMP(1);
        prio(4);

        double sre[2][R], sim[2][R];       // [0]: hi half, [1]: lo half of the own output polynomial; the partner's partials pass through [1]
        auto put = [&](cplx* h, const double (&pr)[R], const double (&pi)[R]) {
#pragma unroll
            for (int m = 0; m < R; m++) h[m * 64] = make_double2(pr[m], pi[m]);
        };
        auto get = [&](const cplx* h, double (&pr)[R], double (&pi)[R]) {
#pragma unroll
            for (int m = 0; m < R; m++) { const cplx v = h[m * 64]; pr[m] = v.x; pi[m] = v.y; }
        };
        // the four multiply-accumulate phases over the 24 half rows of this side (phase = hr / 6, row = hr % 6 / 2, half = hr % 2)
        //   phase 0 (M1): hi half of the PARTNER's polynomial over the own rows -> own buffer, then the first hand-off
        //   phase 1 (M2): hi half of the OWN polynomial on top of the partner's partial
        //   phase 2 (M3): lo half of the partner's polynomial -> the PARTNER's buffer (read in phase 1: free), then the second hand-off
        //   phase 3 (M4): lo half of the own polynomial on top of the partner's partial
#pragma unroll
        for (int hr = 0; hr < HALF_ROWS; hr++) {
            const int phase = hr / 6, row = (hr % 6) / 2, h = hr & 1;
            if (hr == 6) {
                put(hand_mine, sre[1], sim[1]);
                PAIR_STAMP(2);
                XF_SYNC(1u);
                PAIR_STAMP(3);
                prio(5);
                get(hand_peer, sre[0], sim[0]);
            }
            if (hr == 12) prio(6);
            if (hr == 18) {
                put(hand_peer, sre[1], sim[1]);
                PAIR_STAMP(4);
                XF_SYNC(2u);
                PAIR_STAMP(5);
                prio(7);
                get(hand_mine, sre[1], sim[1]);
            }
            const bool first = (phase == 0 || phase == 2) && row == 0;
            if (phase == 1) xfft::mac_half(sre[0], sim[0], kb[hr % 3], xr[row], xi[row], h, first);
            else xfft::mac_half(sre[1], sim[1], kb[hr % 3], xr[row], xi[row], h, first);
            fetch(kb[hr % 3], hr + 3 < HALF_ROWS ? i : nxt, (hr + 3) % HALF_ROWS);
        }

        PAIR_STAMP(6);
        prio(8);
        xfft::inverse_multi<2>(sre, sim, tw, myx, myx + G::XSLOTS, lane, [&](int k) { prio(8 + k); });
        PAIR_STAMP(7);
        prio(11);
#pragma unroll
        for (int m = 0; m < R; m++) {
            const int c = lane + 64 * m;
            // ds_add_u32: no read-back, and no 16 registers of own coefficients live across the step (they spilled)
            __hip_atomic_fetch_add(&poly[c], xfft::rounded_hi16(sre[0][m]) + xfft::rounded_u32(sre[1][m]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __hip_atomic_fetch_add(&poly[c + P], xfft::rounded_hi16(sim[0][m]) + xfft::rounded_u32(sim[1][m]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
        wave_lds_sync();
        PAIR_STAMP(8);
        prio(12);
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

    if (a.mode == MODE_BLIND_ROTATE) {
        if (live) {
            uint32_t* o = a.out + (size_t)g * 2 * N + side * N;
            for (int c = lane; c < N; c += 64) o[c] = poly[c];
        }
        return;
    }
    // sample extract index 0 (trlwe.rs:110-121): side 1 owns the a-poly
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
            const int ge = a.ext_first + g;
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
#pragma unroll
        for (int q = 0; q < KSQ; q++) {
            const uint4 o = part[q * 64];
            const int col = 4 * (lane + 64 * q);
            const uint32_t s[4] = {sum[q].x + o.x, sum[q].y + o.y, sum[q].z + o.z, sum[q].w + o.w};
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (col + e <= n) io.out[col + e] = ((col + e == n) ? bprime : 0u) - s[e];
        }
    }
}

// ---- key rows -> split spectra (the counterpart of TRGSWRepF::from, hom_nand/src/trgsw.rs:68-76) ----
struct XBkArgs {
    const cplx* xtw;
    const uint32_t* bk_torus;   // [n][2 comp][2l rows][N]
    cplx* xbk;                  // device layout (XBootstrapArgs)
    int32_t count;              // polynomials = n * 2 * 2l
    int32_t rows;               // 2l
};

// device index (in rows of 512 cplx) of source polynomial g = (i, comp, row) and key half
__host__ __device__ inline size_t xbk_row_index(size_t g, int rows, int half) {
    const size_t i = g / (2 * (size_t)rows), rem = g % (2 * (size_t)rows);
    const int comp = (int)(rem / rows), row = (int)(rem % rows), l = rows / 2;
    const int side = row / l, local = row % l;
    const int phase = 2 * half + (comp == side ? 1 : 0);
    return (i * 2 + side) * (size_t)(4 * l) + (size_t)phase * l + local;
}

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void k_xbk_build(const XBkArgs a) {
    typedef Geo<10> G;
    constexpr int N = G::N, P = G::P, R = G::R;
    extern __shared__ __align__(16) unsigned char smem[];
    cplx* tw = reinterpret_cast<cplx*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int idx = tid; idx < xfft::XTw::TOTAL; idx += 64 * WAVES) tw[idx] = a.xtw[idx];
    cplx w1[7];
#pragma unroll
    for (int e = 0; e < 7; e++) w1[e] = a.xtw[xfft::XTw::F1 + e];
    __syncthreads();
    double* xbuf = reinterpret_cast<double*>(smem + (size_t)xfft::XTw::TOTAL * sizeof(cplx)) + (size_t)wave * 2 * G::XSLOTS;
    for (int g = blockIdx.x * WAVES + wave; g < a.count; g += gridDim.x * WAVES) {
        const int32_t* src = reinterpret_cast<const int32_t*>(a.bk_torus) + (size_t)g * N;
        double re[2][R], im[2][R];       // [0]: hi, [1]: lo
#pragma unroll
        for (int m = 0; m < R; m++) {
            const int32_t k0 = src[lane + 64 * m], k1 = src[lane + 64 * m + P];
            const int32_t l0 = (int32_t)(int16_t)k0, l1 = (int32_t)(int16_t)k1;      // lo in [-2^15, 2^15)
            re[1][m] = (double)l0; im[1][m] = (double)l1;
            re[0][m] = (double)(int32_t)(((int64_t)k0 - l0) >> 16);                  // hi in [-2^15, 2^15]
            im[0][m] = (double)(int32_t)(((int64_t)k1 - l1) >> 16);
        }
        xfft::forward_multi<2>(re, im, tw, w1, xbuf, xbuf + G::XSLOTS, lane);
#pragma unroll
        for (int half = 0; half < 2; half++) {
            cplx* dst = a.xbk + xbk_row_index((size_t)g, a.rows, half) * P + lane;
#pragma unroll
            for (int m = 0; m < R; m++) dst[m * 64] = make_double2(re[half][m], im[half][m]);
        }
    }
}

// ---- stage-level external product on this backend (rtfhe_external_product_batch): one wave per sample ----
struct XExtProdArgs {
    const cplx* xtw;
    const cplx* xbk;
    const int32_t* bk_index;
    const uint32_t* trlwe;
    uint32_t* out;
    int32_t count;
};

template <int L, int BGBIT, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void k_external_product_xfft(const XExtProdArgs a) {
    typedef Geo<10> G;
    constexpr int N = G::N, P = G::P, R = G::R;
    constexpr uint32_t M = decomp_mask(L, BGBIT);
    extern __shared__ __align__(16) unsigned char smem[];
    cplx* tw = reinterpret_cast<cplx*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int idx = tid; idx < xfft::XTw::TOTAL; idx += 64 * WAVES) tw[idx] = a.xtw[idx];
    cplx w1[7];
#pragma unroll
    for (int e = 0; e < 7; e++) w1[e] = a.xtw[xfft::XTw::F1 + e];
    __syncthreads();
    const int g = blockIdx.x * WAVES + wave;
    if (g >= a.count) return;
    double* xbuf = reinterpret_cast<double*>(smem + (size_t)xfft::XTw::TOTAL * sizeof(cplx)) + (size_t)wave * 2 * G::XSLOTS;
    const uint32_t* in = a.trlwe + (size_t)g * 2 * N;
    const cplx* bk_i = a.xbk + (size_t)a.bk_index[g] * (2 * 12 * P);
    // out[c] = sum over both input polynomials (sides) and their three digit rows; hi and lo halves kept apart
    double s[2][2][2][R];        // [comp][half][re / im][R]
#pragma unroll
    for (int comp = 0; comp < 2; comp++)
#pragma unroll
        for (int half = 0; half < 2; half++)
#pragma unroll
            for (int m = 0; m < R; m++) { s[comp][half][0][m] = 0.0; s[comp][half][1][m] = 0.0; }
#pragma unroll 1
    for (int side = 0; side < 2; side++) {
        uint32_t u[2 * R];
#pragma unroll
        for (int mm = 0; mm < 2 * R; mm++) u[mm] = (in[side * N + lane + 64 * mm] + M) ^ M;
#pragma unroll 1
        for (int jj = 0; jj < L; jj++) {
            double xr[1][R], xi[1][R];
#pragma unroll
            for (int m = 0; m < R; m++) {
                xr[0][m] = (double)decomp_digit(u[m], BGBIT, jj);
                xi[0][m] = (double)decomp_digit(u[R + m], BGBIT, jj);
            }
            xfft::forward_multi<1>(xr, xi, tw, w1, xbuf, xbuf + G::XSLOTS, lane);
#pragma unroll
            for (int phase = 0; phase < 4; phase++) {
                const int half = phase >> 1, comp = (phase & 1) ? side : 1 - side;
                const cplx* row = bk_i + (size_t)((side * 12) + phase * L + jj) * P + lane;
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
        double yr[2][R], yi[2][R];
#pragma unroll
        for (int half = 0; half < 2; half++)
#pragma unroll
            for (int m = 0; m < R; m++) { yr[half][m] = comp ? s[1][half][0][m] : s[0][half][0][m]; yi[half][m] = comp ? s[1][half][1][m] : s[0][half][1][m]; }
        xfft::inverse_multi<2>(yr, yi, tw, xbuf, xbuf + G::XSLOTS, lane);
#pragma unroll
        for (int m = 0; m < R; m++) {
            const int c = lane + 64 * m;
            o[comp * N + c] = xfft::rounded_hi16(yr[0][m]) + xfft::rounded_u32(yr[1][m]);
            o[comp * N + c + P] = xfft::rounded_hi16(yi[0][m]) + xfft::rounded_u32(yi[1][m]);
        }
    }
}

}  // namespace rtfhe
