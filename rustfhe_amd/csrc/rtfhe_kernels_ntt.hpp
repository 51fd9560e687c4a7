// rtfhe_kernels_ntt.hpp -- the bootstrap kernel on the exact-integer NTT backend (rtfhe_ntt.hpp).  Same gate pipeline
// as k_bootstrap (pre-step, blind rotate, sample extract, key switch; one wave per gate) with the FP64-FFT mirror
// replaced by exact products mod P.  Integer glue (rotate, decomposition, mod switch, key switch) is shared.
#pragma once

#include "rtfhe_kernels.hpp"
#include "rtfhe_kernels_pair.hpp"
#include "rtfhe_ntt.hpp"

namespace rtfhe {

// BK in the NTT domain, device layout: double[n][2l rows][2 comp][8][64 lanes][2]: lane v holds points 16 v + m (layout L3),
// (m = 2 q + e) at [q][v][e]; N^-1 folded in; centred residues.
__device__ __forceinline__ const double2* ntt_bk_row(const double* bk_i, int row, int comp, int lane) {
    return reinterpret_cast<const double2*>(bk_i + ((size_t)(row * 2 + comp) * ntt::N)) + lane;
}

template <int L, int BGBIT, bool CMUX>
__device__ __forceinline__ void cmux_step_ntt(uint32_t* __restrict__ accbuf, int r, const double* __restrict__ bk_i,
                                              const double* __restrict__ twf, const double* __restrict__ twi,
                                              double* __restrict__ xbuf, int lane) {
    constexpr int N = ntt::N, R = ntt::R;
    constexpr uint32_t M = decomp_mask(L, BGBIT);
    static_assert((1 << BGBIT) == ntt::DIGITS, "the digit table has one entry per digit value");
    double s0[R], s1[R];
#pragma unroll
    for (int m = 0; m < R; m++) { s0[m] = 0.0; s1[m] = 0.0; }
#pragma unroll 1
    for (int h = 0; h < 2; h++) {
        const uint32_t* poly = accbuf + h * N;
        uint32_t u[R];
#pragma unroll
        for (int m = 0; m < R; m++) {
            const int c = lane + 64 * m;
            const uint32_t own = poly[c];
            const uint32_t d = CMUX ? (rotated_coef<10>(poly, c, r) - own) : own;
            u[m] = (d + M) ^ M;
        }
#pragma unroll 1
        for (int jj = 0; jj < L; jj++) {
            double x[R];
            const double2* b0p = ntt_bk_row(bk_i, h * L + jj, 0, lane);
            const double2* b1p = ntt_bk_row(bk_i, h * L + jj, 1, lane);
            // key rows requested up front (measured: 13.3 ms vs 14.3 ms per 1024 gates when requested after pass 2)
            double2 b0[R / 2], b1[R / 2];
#pragma unroll
            for (int q = 0; q < R / 2; q++) { b0[q] = b0p[q * 64]; b1[q] = b1p[q * 64]; }
            ntt::first_two_stages_digits(x, u, BGBIT, jj, twf + ntt::TW_DIG);
            ntt::forward_a<2, true>(x, twf, xbuf, lane);
            ntt::forward_b<true>(x, twf, xbuf, lane);
            // exact arithmetic: the order of the row sum is irrelevant here (it is not for the FFT mirror)
#pragma unroll
            for (int q = 0; q < R / 2; q++) {
                s0[2 * q] += ntt::modmul(x[2 * q], b0[q].x);     s0[2 * q + 1] += ntt::modmul(x[2 * q + 1], b0[q].y);
                s1[2 * q] += ntt::modmul(x[2 * q], b1[q].x);     s1[2 * q + 1] += ntt::modmul(x[2 * q + 1], b1[q].y);
            }
        }
    }
#pragma unroll 1
    for (int comp = 0; comp < 2; comp++) {
        double x[R];
#pragma unroll
        for (int m = 0; m < R; m++) x[m] = comp ? s1[m] : s0[m];
        ntt::inverse<true>(x, twi, xbuf, lane);
        uint32_t* poly = accbuf + comp * N;
#pragma unroll
        for (int m = 0; m < R; m++) {
            const int c = lane + 64 * m;
            if (CMUX) poly[c] += ntt::to_torus(x[m]);
            else poly[c] = ntt::to_torus(x[m]);
        }
    }
    wave_lds_sync();
}

__host__ __device__ constexpr size_t ntt_wave_lds_bytes(int npad) {
    return (size_t)ntt::XSLOTS * sizeof(double) + (size_t)2 * ntt::N * 4 + (size_t)npad * 4;
}
__host__ __device__ constexpr size_t ntt_lds_bytes(int waves, int npad) {
    return (size_t)ntt::TW_TOTAL * sizeof(double) + (size_t)waves * ntt_wave_lds_bytes(npad);
}

struct NttBootstrapArgs {
    BootstrapArgs b;          // tw and bk of `b` are unused here
    const double* ntt_tw;     // [ntt::TW_TOTAL]
    const double* ntt_bk;     // device layout above
};

template <int L, int BGBIT, int KS_T, int KS_BB, int KSQ, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void k_bootstrap_ntt(const NttBootstrapArgs args) {
    constexpr int N = ntt::N, R = ntt::R, LOGN = 10;
    const BootstrapArgs& a = args.b;
    extern __shared__ __align__(16) unsigned char smem[];
    double* tw = reinterpret_cast<double*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int idx = tid; idx < ntt::TW_TOTAL; idx += 64 * WAVES) tw[idx] = args.ntt_tw[idx];
    __syncthreads();
    const int g = blockIdx.x * WAVES + wave;
    if (g >= a.count) return;
    unsigned char* wbase = smem + (size_t)ntt::TW_TOTAL * sizeof(double) + (size_t)wave * ntt_wave_lds_bytes(a.npad);
    double* xbuf = reinterpret_cast<double*>(wbase);
    uint32_t* accbuf = reinterpret_cast<uint32_t*>(wbase + (size_t)ntt::XSLOTS * sizeof(double));
    uint32_t* abar = accbuf + 2 * N;
    const double* twf = tw;
    const double* twi = tw + ntt::TW_DIR_PAD;
    const int n = a.n;
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
    {
        const int bbar = (int)abar[n];
#pragma unroll
        for (int m = 0; m < R; m++) {
            const int c = lane + 64 * m;
            const int e = (c + bbar) & (2 * N - 1);
            accbuf[c] = (e >> LOGN) ? 0xE0000000u : 0x20000000u;
            accbuf[N + c] = 0u;
        }
    }
    wave_lds_sync();
    const size_t trgsw_doubles = (size_t)2 * L * 2 * N;
#pragma unroll 1
    for (int i = 0; i < a.steps; i++) {
        const int r = __builtin_amdgcn_readfirstlane((int)abar[i]);
        cmux_step_ntt<L, BGBIT, true>(accbuf, r, args.ntt_bk + (size_t)i * trgsw_doubles, twf, twi, xbuf, lane);
    }
    if (a.mode == MODE_BLIND_ROTATE) {
        uint32_t* o = a.out + (size_t)g * 2 * N;
        for (int c = lane; c < 2 * N; c += 64) o[c] = accbuf[c];
        return;
    }
    uint32_t av[R];
#pragma unroll
    for (int m = 0; m < R; m++) av[m] = accbuf[N + lane + 64 * m];
    const uint32_t bprime = accbuf[0];
    wave_lds_sync();
#pragma unroll
    for (int m = 0; m < R; m++) {
        const int c = lane + 64 * m;
        accbuf[N + ((N - c) & (N - 1))] = (c == 0) ? av[m] : (0u - av[m]);
    }
    wave_lds_sync();
    key_switch_wave<LOGN, KS_T, KS_BB, KSQ>(accbuf + N, bprime, a.ksk, a.ksw, n, io.out, lane);
}

// Two waves per gate on the NTT backend (see rtfhe_kernels_pair.hpp for why: one gate per SIMD leaves the FP64 pipe with a
// single wave).  Exact integer arithmetic carries no fold-order constraint, so the split is symmetric: side s gathers and
// decomposes accumulator polynomial s, transforms its three digit polynomials and accumulates BOTH components over its
// three key rows; then the sides swap the component the other one owns (through their own idle exchange buffers), add,
// run one inverse transform each and update their own polynomial.  Sums are sums of the same exact integers as in
// k_bootstrap_ntt => identical words.
struct NttPairLds {
    static constexpr size_t TW = (size_t)ntt::TW_TOTAL * sizeof(double);
    static constexpr size_t XB = (size_t)ntt::XSLOTS * sizeof(double);
    static_assert(ntt::XSLOTS >= ntt::N, "an exchange buffer must hold one spectrum");
    static constexpr size_t FLAGS = 16;       // two arrival counters per gate (pair_sync, rtfhe_kernels_pair.hpp)
    __host__ __device__ static constexpr size_t gate_bytes(int npad) { return (size_t)2 * ntt::N * 4 + (size_t)npad * 4 + 2 * XB + FLAGS; }
    __host__ __device__ static constexpr size_t bytes(int gates, int npad) { return TW + (size_t)gates * gate_bytes(npad); }
};


template <int L, int BGBIT, int KS_T, int KS_BB, int KSQ, int GATES>
__global__ __launch_bounds__(128 * GATES, 1) void k_bootstrap_ntt_pair(const NttBootstrapArgs args) {
    constexpr int N = ntt::N, R = ntt::R, LOGN = 10, NT = 128 * GATES;
    constexpr uint32_t M = decomp_mask(L, BGBIT);
    const BootstrapArgs& a = args.b;
    extern __shared__ __align__(16) unsigned char smem[];
    double* tw = reinterpret_cast<double*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slot = wave % GATES, side = wave / GATES;
    for (int idx = tid; idx < ntt::TW_TOTAL; idx += NT) tw[idx] = args.ntt_tw[idx];
    const double* twf = tw;
    const double* twi = tw + ntt::TW_DIR_PAD;

    const int g_raw = blockIdx.x * GATES + slot;
    const int g = g_raw < a.count ? g_raw : a.count - 1;
    const GateIo io = gate_io(a, g);
    const bool live = g_raw < a.count && io.ok;      // idle / skipped pairs still take part in every barrier

    unsigned char* gbase = smem + NttPairLds::TW + (size_t)slot * NttPairLds::gate_bytes(a.npad);
    uint32_t* accbuf = reinterpret_cast<uint32_t*>(gbase);
    uint32_t* abar = accbuf + 2 * N;
    double* xb0 = reinterpret_cast<double*>(gbase + (size_t)2 * N * 4 + (size_t)a.npad * 4);
    double* xb1 = xb0 + ntt::XSLOTS;
    double* myx = side ? xb1 : xb0;
    double* peerx = side ? xb0 : xb1;
    // the two waves of a gate synchronise with each other only when the workgroup is not full (as in k_bootstrap_pair)
    uint32_t* flags = reinterpret_cast<uint32_t*>(gbase + NttPairLds::gate_bytes(a.npad) - NttPairLds::FLAGS);
    if (lane == 0) flags[side] = 0u;
    [[maybe_unused]] const unsigned my_flag = (unsigned)(size_t)(__attribute__((address_space(3))) uint32_t*)(flags + side);
    [[maybe_unused]] const unsigned partner_flag = (unsigned)(size_t)(__attribute__((address_space(3))) uint32_t*)(flags + (1 - side));
    constexpr bool FLAG_SYNC = GATES <= 2;      // measured: 512 gates 6.93 -> 6.69 ms; 768 gates 10.26 -> 10.49 (slower); 1024 gates equal
    uint32_t* poly = accbuf + side * N;
    const int n = a.n;
    {
        constexpr int SH = 32 - LOGN - 1;
        for (int i = lane + 64 * side; i <= n; i += 128) {
            const uint32_t t = gate_linear(io.op, io.p0[i], io.p1[i], i == n);
            abar[i] = (i == n) ? (t >> SH) : ((t + (1u << (SH - 1))) >> SH);
        }
    }
    __syncthreads();
    {
        const int bbar = (int)abar[n];
#pragma unroll
        for (int m = 0; m < R; m++) {
            const int c = lane + 64 * m;
            const int e = (c + bbar) & (2 * N - 1);
            poly[c] = side ? 0u : ((e >> LOGN) ? 0xE0000000u : 0x20000000u);
        }
    }
    wave_lds_sync();
    // priority schedule as in k_bootstrap_pair: side 1 at 1, side 0 at 2 from RAISE_AT to LOWER_AT, else 0
    constexpr int LOWER_AT = 2, RAISE_AT = 8;
    auto prio_point = [&](int point) {   // `point` may be a run-time (scalar) value: selection by scalar ALU, the branch stays inside the asm
        const int lower = __builtin_amdgcn_readfirstlane((point == LOWER_AT) & (side == 0));
        const int raise = __builtin_amdgcn_readfirstlane((point == RAISE_AT) & (side == 0));
        asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 1f\n\ts_setprio 0\n1:" ::"s"(lower) : "scc");
        asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 1f\n\ts_setprio 2\n1:" ::"s"(raise) : "scc");
    };
    if (side) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(2);

    const size_t trgsw_doubles = (size_t)2 * L * 2 * N;
    uint32_t own[R];
#pragma unroll
    for (int m = 0; m < R; m++) own[m] = poly[lane + 64 * m];
#pragma unroll 1
    for (int i = 0; i < a.steps; i++) {
        const int r = __builtin_amdgcn_readfirstlane((int)abar[i]);
        const double* bk_i = args.ntt_bk + (size_t)i * trgsw_doubles;
        uint32_t u[R];
#pragma unroll
        for (int m = 0; m < R; m++) {
            const int c = lane + 64 * m;
            u[m] = ((rotated_coef<10>(poly, c, r) - own[m]) + M) ^ M;
        }
        double s0[R], s1[R];
#pragma unroll
        for (int m = 0; m < R; m++) { s0[m] = 0.0; s1[m] = 0.0; }
#pragma unroll 1
        for (int jj = 0; jj < L; jj++) {
            double x[R];
            const double2* b0p = ntt_bk_row(bk_i, side * L + jj, 0, lane);
            const double2* b1p = ntt_bk_row(bk_i, side * L + jj, 1, lane);
            double2 b0[R / 2], b1[R / 2];
#pragma unroll
            for (int q = 0; q < R / 2; q++) { b0[q] = b0p[q * 64]; b1[q] = b1p[q * 64]; }
            ntt::first_two_stages_digits(x, u, BGBIT, jj, twf + ntt::TW_DIG);
            ntt::forward_a<2, true>(x, twf, myx, lane);
            prio_point(2 * jj);
            ntt::forward_b<true>(x, twf, myx, lane);
            prio_point(2 * jj + 1);
#pragma unroll
            for (int q = 0; q < R / 2; q++) {
                s0[2 * q] += ntt::modmul(x[2 * q], b0[q].x);     s0[2 * q + 1] += ntt::modmul(x[2 * q + 1], b0[q].y);
                s1[2 * q] += ntt::modmul(x[2 * q], b1[q].x);     s1[2 * q + 1] += ntt::modmul(x[2 * q + 1], b1[q].y);
            }
        }
        // swap: export the component the other side owns through the own (now idle) exchange buffer
        {
            double2* ex = reinterpret_cast<double2*>(myx) + lane;
#pragma unroll
            for (int q = 0; q < R / 2; q++) ex[q * 64] = side ? make_double2(s0[2 * q], s0[2 * q + 1]) : make_double2(s1[2 * q], s1[2 * q + 1]);
        }
        prio_point(6);
        if constexpr (FLAG_SYNC) pair_sync(my_flag, partner_flag, 2u * (unsigned)i + 1u); else lds_barrier();
        prio_point(7);
        double x[R];
        {
            const double2* im = reinterpret_cast<const double2*>(peerx) + lane;
#pragma unroll
            for (int q = 0; q < R / 2; q++) {
                const double2 v = im[q * 64];
                x[2 * q] = (side ? s1[2 * q] : s0[2 * q]) + v.x;
                x[2 * q + 1] = (side ? s1[2 * q + 1] : s0[2 * q + 1]) + v.y;
            }
        }
        if constexpr (FLAG_SYNC) pair_sync(my_flag, partner_flag, 2u * (unsigned)i + 2u); else lds_barrier();   // both imports done: the exchange buffers are free for the inverse transforms
        prio_point(8);
        ntt::inverse<true>(x, twi, myx, lane);
#pragma unroll
        for (int m = 0; m < R; m++) {       // the new coefficients also stay in registers for the gather that follows
            own[m] = poly[lane + 64 * m] + ntt::to_torus(x[m]);
            poly[lane + 64 * m] = own[m];
        }
        wave_lds_sync();
        prio_point(9);
    }
    __builtin_amdgcn_s_setprio(0);

    if (a.mode == MODE_BLIND_ROTATE) {
        if (live) {
            uint32_t* o = a.out + (size_t)g * 2 * N + side * N;
            for (int c = lane; c < N; c += 64) o[c] = poly[c];
        }
        return;
    }
    // sample extract (side 1 owns the a-poly) + key switch split over both sides, as in k_bootstrap_pair
    if (side == 1) {
        uint32_t av[R];
#pragma unroll
        for (int m = 0; m < R; m++) av[m] = poly[lane + 64 * m];
        wave_lds_sync();
#pragma unroll
        for (int m = 0; m < R; m++) {
            const int c = lane + 64 * m;
            poly[(N - c) & (N - 1)] = (c == 0) ? av[m] : (0u - av[m]);
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
    uint4 sum[KSQ];
    ks_accumulate<LOGN, KS_T, KS_BB, KSQ>(accbuf + N, side * (N / 2), (side + 1) * (N / 2), a.ksk, a.ksw, sum, lane);
    uint4* part = reinterpret_cast<uint4*>(xb1) + lane;
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
            const uint32_t sv[4] = {sum[q].x + o.x, sum[q].y + o.y, sum[q].z + o.z, sum[q].w + o.w};
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (col + e <= n) io.out[col + e] = ((col + e == n) ? bprime : 0u) - sv[e];
        }
    }
}

struct NttBkArgs {
    const double* ntt_tw;
    const uint32_t* bk_torus;   // [n][2][2l][N]
    double* ntt_bk;             // device layout
    int32_t count;              // polynomials
    int32_t rows;               // 2l
    double ninv;                // N^-1 mod P, centred
};

// key rows -> NTT domain (the counterpart of TRGSWRepF::from, hom_nand/src/trgsw.rs:68-76): words viewed as signed i32
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void k_ntt_bk(const NttBkArgs a) {
    constexpr int N = ntt::N, R = ntt::R;
    extern __shared__ __align__(16) unsigned char smem[];
    double* tw = reinterpret_cast<double*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int idx = tid; idx < ntt::TW_DIR_PAD; idx += 64 * WAVES) tw[idx] = a.ntt_tw[idx];
    __syncthreads();
    double* xbuf = tw + ntt::TW_DIR_PAD + (size_t)wave * ntt::XSLOTS;
    for (int g = blockIdx.x * WAVES + wave; g < a.count; g += gridDim.x * WAVES) {
        const int32_t* src = reinterpret_cast<const int32_t*>(a.bk_torus) + (size_t)g * N;
        double x[R];
#pragma unroll
        for (int m = 0; m < R; m++) x[m] = (double)src[lane + 64 * m];
        ntt::forward(x, tw, xbuf, lane);
        double2* dst = reinterpret_cast<double2*>(a.ntt_bk + bk_poly_remap((size_t)g, a.rows) * N) + lane;
#pragma unroll
        for (int q = 0; q < R / 2; q++)
            dst[q * 64] = make_double2(ntt::normalize(ntt::modmul(x[2 * q], a.ninv)), ntt::normalize(ntt::modmul(x[2 * q + 1], a.ninv)));
    }
}

struct NttExtProdArgs {
    const double* ntt_tw;
    const double* ntt_bk;
    const int32_t* bk_index;
    const uint32_t* trlwe;
    uint32_t* out;
    int32_t count;
};

template <int L, int BGBIT, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void k_external_product_ntt(const NttExtProdArgs a) {
    constexpr int N = ntt::N;
    extern __shared__ __align__(16) unsigned char smem[];
    double* tw = reinterpret_cast<double*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int idx = tid; idx < ntt::TW_TOTAL; idx += 64 * WAVES) tw[idx] = a.ntt_tw[idx];
    __syncthreads();
    const int g = blockIdx.x * WAVES + wave;
    if (g >= a.count) return;
    unsigned char* wbase = smem + (size_t)ntt::TW_TOTAL * sizeof(double) + (size_t)wave * ntt_wave_lds_bytes(0);
    double* xbuf = reinterpret_cast<double*>(wbase);
    uint32_t* accbuf = reinterpret_cast<uint32_t*>(wbase + (size_t)ntt::XSLOTS * sizeof(double));
    for (int c = lane; c < 2 * N; c += 64) accbuf[c] = a.trlwe[(size_t)g * 2 * N + c];
    wave_lds_sync();
    cmux_step_ntt<L, BGBIT, false>(accbuf, 0, a.ntt_bk + (size_t)a.bk_index[g] * ((size_t)2 * L * 2 * N), tw, tw + ntt::TW_DIR_PAD, xbuf, lane);
    for (int c = lane; c < 2 * N; c += 64) a.out[(size_t)g * 2 * N + c] = accbuf[c];
}

}  // namespace rtfhe
