// rtfhe_kernels_xfft_rr.hpp -- k_bootstrap_xpair with FIVE or SIX gates on the four wave pairs of a CU (split-FFT exact backend, N = 1024):
// batches between whole rounds, exactly as rtfhe_kernels_pair_rr.hpp does it for the mirror backend.
//
// The CMUX steps of a workgroup's gc = 4 .. 6 gates form one sequence of items t = step * gc + gate; pair s works through t = s, s + 4, ...; a gate
// changes hands from step to step through LDS (its two polynomials -- already updated in place by ds_add_u32 in k_bootstrap_xpair -- and its
// rotation amounts live there), each side publishing "gate g through step i" in a flag of the gate after its update and waiting for that flag
// before it gathers.  The step's arithmetic is k_bootstrap_xpair's, instruction for instruction (exact sums: the words are the NTT backend's).
#pragma once

#include "rtfhe_kernels_xfft.hpp"

namespace rtfhe {

struct XPairRrLds {
    typedef Geo<10> G;
    static constexpr int SLOTS = 4, GMAX = 6;
    static constexpr size_t TW = XPairLds::TW, XB = XPairLds::XB;
    static constexpr size_t SLOT = 2 * XB + 16;            // a pair's exchange / hand-off buffers and its two arrival counters
    static constexpr size_t DONE = 64;                     // [GMAX][2 sides] steps done, per gate and side
    __host__ __device__ static constexpr size_t gate_bytes(int npad) { return (size_t)2 * G::N * 4 + ((size_t)npad * 2 + 15) / 16 * 16; }
    __host__ __device__ static constexpr size_t bytes(int gates, int npad) { return TW + SLOTS * SLOT + DONE + (size_t)gates * gate_bytes(npad); }
};

// gridDim.x workgroups share a.count gates evenly (the first a.count % gridDim.x take one more); the host launches 4 <= gates per workgroup <= 6
template <int L, int BGBIT, int KS_T, int KS_BB, int KSQ>
__global__ __launch_bounds__(512, 1) void k_bootstrap_xpair_rr(const XBootstrapArgs args) {
    typedef XPairRrLds S;
    constexpr int LOGN = 10, SLOTS = S::SLOTS;
    typedef Geo<LOGN> G;
    constexpr int N = G::N, P = G::P, R = G::R, NT = 128 * SLOTS;
    constexpr uint32_t M = decomp_mask(L, BGBIT);
    static_assert(L == 3 && R == xfft::R, "three rows per side are held in registers");
    const BootstrapArgs& a = args.b;
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int side = wave / SLOTS;
    const int slot = wave % SLOTS;
    const int per = a.count / (int)gridDim.x, extra = a.count % (int)gridDim.x, wg = (int)blockIdx.x;
    const int gc = per + (wg < extra ? 1 : 0);
    const int g_first = wg * per + (wg < extra ? wg : extra);
    if (gc < SLOTS || gc > S::GMAX) return;      // not a shape this kernel serves (the host never launches one): uniform exit

    cplx* tw = reinterpret_cast<cplx*>(smem);
    for (int idx = tid; idx < xfft::XTw::TOTAL; idx += NT) tw[idx] = args.xtw[idx];
    cplx w1[7];       // forward pass 1: wave-uniform twiddles (scalar loads)
#pragma unroll
    for (int e = 0; e < 7; e++) w1[e] = args.xtw[xfft::XTw::F1 + e];

    unsigned char* sbase = smem + S::TW + (size_t)slot * S::SLOT;
    double* xb0 = reinterpret_cast<double*>(sbase);
    double* xb1 = xb0 + 2 * G::XSLOTS;
    double* myx = side ? xb1 : xb0;
    uint32_t* flags = reinterpret_cast<uint32_t*>(sbase + 2 * S::XB);
    uint32_t* done = reinterpret_cast<uint32_t*>(smem + S::TW + SLOTS * S::SLOT);
    unsigned char* gates0 = smem + S::TW + SLOTS * S::SLOT + S::DONE;
    const size_t gate_bytes = S::gate_bytes(a.npad);
    if (lane == 0) flags[side] = 0u;
    if (tid < 2 * S::GMAX) done[tid] = 0u;
    const unsigned my_flag = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) uint32_t*)(flags + side));
    const unsigned partner_flag = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) uint32_t*)(flags + (1 - side)));
    const unsigned done_side = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) uint32_t*)(done + side));
    cplx* hand_mine = reinterpret_cast<cplx*>(myx) + lane;                       // [R][64] cplx
    cplx* hand_peer = reinterpret_cast<cplx*>(side ? xb0 : xb1) + lane;

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

    // the key ring of k_bootstrap_xpair (three half-row buffers), running across ITEMS
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

    auto prio = [&](int k) {      // k_bootstrap_xpair's schedule
        if (k & 1) asm volatile("s_cmp_lg_u32 %0, 0\n\ts_cbranch_scc1 1f\n\ts_setprio 0\n1:" ::"s"(side) : "scc");
        else asm volatile("s_cmp_lg_u32 %0, 0\n\ts_cbranch_scc1 1f\n\ts_setprio 2\n1:" ::"s"(side) : "scc");
    };
    if (side) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(2);
    unsigned seq = 0;                               // the pair's hand-offs, counted over the whole kernel
#define XRR_SYNC() do { seq++; xfft::flag_arrive(my_flag, seq); xfft::flag_wait(partner_flag, seq); } while (0)
    int gl = slot, i = 0;                           // this pair's current item: step i of gate gl
#pragma unroll 1
    while (i < a.steps) {
        int gl2 = gl + SLOTS, i2 = i;               // ... and its next one
        if (gl2 >= gc) { gl2 -= gc; i2++; }
        const int nxt = (i2 < a.steps) ? i2 : i;
        uint32_t* poly = reinterpret_cast<uint32_t*>(gates0 + gl * gate_bytes) + side * N;
        const int r = __builtin_amdgcn_readfirstlane((int)reinterpret_cast<const uint16_t*>(gates0 + gl * gate_bytes + (size_t)2 * N * 4)[i]);
        const unsigned done_flag = done_side + 8u * (unsigned)gl;
        xfft::flag_wait(done_flag, (unsigned)i);    // this side's polynomial of gate gl has been through step i - 1 (on whichever pair ran it)
        int ln = lane;
        asm volatile("" : "+v"(ln));
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
        prio(1);
        xfft::forward_multi<L>(xr, xi, tw, w1, myx, myx + G::XSLOTS, ln, [&](int k) { prio(1 + k); });
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
        // the four multiply-accumulate phases over the 24 half rows of this side: k_bootstrap_xpair's
#pragma unroll
        for (int hr = 0; hr < HALF_ROWS; hr++) {
            const int phase = hr / 6, row = (hr % 6) / 2, h = hr & 1;
            if (hr == 6) {
                put(hand_mine, sre[1], sim[1]);
                XRR_SYNC();
                prio(5);
                get(hand_peer, sre[0], sim[0]);
            }
            if (hr == 12) prio(6);
            if (hr == 18) {
                put(hand_peer, sre[1], sim[1]);
                XRR_SYNC();
                prio(7);
                get(hand_mine, sre[1], sim[1]);
            }
            const bool first = (phase == 0 || phase == 2) && row == 0;
            if (phase == 1) xfft::mac_half(sre[0], sim[0], kb[hr % 3], xr[row], xi[row], h, first);
            else xfft::mac_half(sre[1], sim[1], kb[hr % 3], xr[row], xi[row], h, first);
            fetch(kb[hr % 3], hr + 3 < HALF_ROWS ? i : nxt, (hr + 3) % HALF_ROWS);
        }

        prio(8);
        xfft::inverse_multi<2>(sre, sim, tw, myx, myx + G::XSLOTS, lane, [&](int k) { prio(8 + k); });
        prio(11);
#pragma unroll
        for (int m = 0; m < R; m++) {
            const int c = lane + 64 * m;
            __hip_atomic_fetch_add(&poly[c], xfft::rounded_hi16(sre[0][m]) + xfft::rounded_u32(sre[1][m]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __hip_atomic_fetch_add(&poly[c + P], xfft::rounded_hi16(sim[0][m]) + xfft::rounded_u32(sim[1][m]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
        wave_lds_sync();
        xfft::flag_arrive(done_flag, (unsigned)i + 1u);   // behind the updates above in this wave's LDS queue: who sees the flag sees the polynomial
        prio(12);
        gl = gl2; i = i2;
    }
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();       // every pair has left its item loop: all gates are through their last step

    // epilogue, gate by gate on the pair that started it (k_bootstrap_xpair's, the pair's own hand-offs in place of the workgroup barrier)
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
        XRR_SYNC();
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
        XRR_SYNC();
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
        XRR_SYNC();      // side 0 has read the partial sums: side 1 may write the next gate's
    }
#undef XRR_SYNC
}

}  // namespace rtfhe
