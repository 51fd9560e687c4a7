// rtfhe_kernels_ntt_wg.hpp -- latency shape of the exact-integer NTT backend (N = 1024): ONE GATE PER WORKGROUP of 8 waves.
//
// k_bootstrap_ntt_pair gives a gate two waves; a circuit wave of a few gates or a single hom_nand() then runs 2 x 3,672
// FP64-rate instructions per step on two lone waves.  Exact arithmetic has no fold order, so the six rows of a step are independent
// until the sum: per CMUX step
//   F  waves 0..5 : wave j gathers / decomposes digit polynomial j, transforms it, multiplies it with BOTH components of key row j and
//                   leaves the two products (partial sums of one row) in LDS.  Six rows on four SIMDs: rows 4, 5 only START on waves
//                   4, 5 (which share SIMDs 0, 1 with waves 0, 1): gather, passes 1 and 2, the write half of the last exchange -- and
//                   FINISH on waves 6, 7 (SIMDs 2, 3): read half, pass 3, products; hand-off by a release / acquire flag in LDS
//   -- barrier --
//   I  waves 0, 1 : wave c adds the six partial sums of component c (any order: exact integers, |sum| <= 3.75 P), runs the inverse
//                   transform and += into the accumulator polynomial c
//   -- barrier --
// Same exact integers as the other NTT kernels => identical words.  Key rows of step i + 1 are requested during step i's I phase.
#pragma once

#include "rtfhe_kernels_ntt.hpp"

namespace rtfhe {

struct NttWgLds {
    static constexpr int NW = 8, ROWS = 6;
    static constexpr size_t TW = 0;
    static constexpr size_t ACC = TW + (size_t)ntt::TW_TOTAL * sizeof(double);
    // two sets of ROWS buffers of one exchange buffer's size: set 0 = the waves' exchange buffers, which take the component-0
    // products once a row's transform is done; set 1 takes the component-1 products (and the key-switch partials at the end).
    // Wave c of the I phase reads the six products of set c and then runs its inverse transform in slot c of set c.
    static constexpr size_t XBUF = ACC + (size_t)2 * ntt::N * 4;                               // double[ROWS][XSLOTS]
    static constexpr size_t PART = XBUF + (size_t)ROWS * ntt::XSLOTS * sizeof(double);          // double[ROWS][XSLOTS]
    static constexpr size_t ABAR = PART + (size_t)ROWS * ntt::XSLOTS * sizeof(double);
    static_assert(ntt::XSLOTS >= ntt::N, "a buffer must hold one row of products");
    __host__ __device__ static constexpr size_t flags(int npad) { return ABAR + (size_t)npad * 4; }      // int[2]: hand-off of rows 4, 5
    __host__ __device__ static constexpr size_t bytes(int npad) { return flags(npad) + 16; }
};

template <int L, int BGBIT, int KS_T, int KS_BB, int KSQ>
__global__ __launch_bounds__(512, 1) void k_bootstrap_ntt_wg(const NttBootstrapArgs args) {
    typedef NttWgLds S;
    constexpr int N = ntt::N, R = ntt::R, LOGN = 10, NW = S::NW, ROWS = S::ROWS;
    static_assert(ROWS == 2 * L && ROWS <= NW, "one wave per row");
    static_assert((1 << BGBIT) == ntt::DIGITS, "the digit table has one entry per digit value");
    constexpr uint32_t M = decomp_mask(L, BGBIT);
    const BootstrapArgs& a = args.b;
    extern __shared__ __align__(16) unsigned char smem[];
    double* tw = reinterpret_cast<double*>(smem + S::TW);
    uint32_t* accbuf = reinterpret_cast<uint32_t*>(smem + S::ACC);
    double* part = reinterpret_cast<double*>(smem + S::PART);
    uint32_t* abar = reinterpret_cast<uint32_t*>(smem + S::ABAR);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double* set0 = reinterpret_cast<double*>(smem + S::XBUF);
    // the row this wave works on in the F phase: waves 6, 7 take over rows 4, 5 at the last exchange
    const int frow = wave < ROWS ? wave : wave - 2;
    const bool macs = wave < ROWS - 2 || wave >= ROWS;           // waves 4, 5 hand their row over before its products
    double* xbuf = set0 + (size_t)frow * ntt::XSLOTS;
    volatile int* flags = reinterpret_cast<volatile int*>(smem + S::flags(a.npad));
    if (tid < 2) flags[tid] = 0;
    const double* twf = tw;
    const double* twi = tw + ntt::TW_DIR_PAD;
    const int g = blockIdx.x;                       // grid = count
    const int n = a.n;

    for (int idx = tid; idx < ntt::TW_TOTAL; idx += 64 * NW) tw[idx] = args.ntt_tw[idx];
    const GateIo io = gate_io(a, g);
    if (!io.ok) return;                             // the whole workgroup serves this gate: uniform exit
    {   // pre-step + mod switch (tfhe.rs:41-71, 97, 107-108)
        constexpr int SH = 32 - LOGN - 1;
        for (int i = tid; i <= n; i += 64 * NW) {
            const uint32_t t = gate_linear(io.op, io.p0[i], io.p1[i], i == n);
            abar[i] = (i == n) ? (t >> SH) : ((t + (1u << (SH - 1))) >> SH);
        }
    }
    __syncthreads();
    {   // acc = X^{-bbar} * testvec (tfhe.rs:85, 98-106)
        const int bbar = (int)abar[n];
        for (int c = tid; c < N; c += 64 * NW) {
            const int e = (c + bbar) & (2 * N - 1);
            accbuf[c] = (e >> LOGN) ? 0xE0000000u : 0x20000000u;
            accbuf[N + c] = 0u;
        }
    }
    __syncthreads();

    // key row `wave` of a step, both components: double2[8][64] each
    const size_t trgsw_doubles = (size_t)2 * L * 2 * N;
    double2 b0[R / 2], b1[R / 2];
    auto load_row = [&](int step) {
        const double* bk_i = args.ntt_bk + (size_t)step * trgsw_doubles;
        const double2* b0p = ntt_bk_row(bk_i, frow, 0, lane);
        const double2* b1p = ntt_bk_row(bk_i, frow, 1, lane);
#pragma unroll
        for (int q = 0; q < R / 2; q++) { b0[q] = b0p[q * 64]; b1[q] = b1p[q * 64]; }
    };
    if (macs && a.steps > 0) load_row(0);
#pragma unroll 1
    for (int i = 0; i < a.steps; i++) {
        const int r = __builtin_amdgcn_readfirstlane((int)abar[i]);
        double x[R];
        if (wave < ROWS) {
            // ---- F: digit polynomial `wave` (trgsw.rs:269-289) and its transform
            const int h = wave / L, jj = wave - h * L;
            const uint32_t* poly = accbuf + h * N;
            if (wave >= ROWS - 2) __builtin_amdgcn_s_setprio(3);    // the waves that hand over must not be starved by waves 0, 1
            uint32_t u[R];
#pragma unroll
            for (int m = 0; m < R; m++) {
                const int c = lane + 64 * m;
                u[m] = ((rotated_coef<LOGN>(poly, c, r) - poly[c]) + M) ^ M;
            }
            ntt::first_two_stages_digits(x, u, BGBIT, jj, twf + ntt::TW_DIG);
            ntt::forward_a<2, true>(x, twf, xbuf, lane);
            if (wave < ROWS - 2) {
                ntt::forward_b<true>(x, twf, xbuf, lane);
            } else {
                ntt::forward_b_send(x, xbuf, lane);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                flags[wave - (ROWS - 2)] = i + 1;
                __builtin_amdgcn_s_setprio(0);
            }
        } else {
            const int k = wave - ROWS;
            while (__builtin_amdgcn_readfirstlane(flags[k]) != i + 1) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            ntt::forward_b_receive<true>(x, twf, xbuf, lane);
        }
        if (macs) {
            // products of the row's spectrum with both components of its key row: the exchange buffer (idle now) takes component 0
            double2* p0 = reinterpret_cast<double2*>(xbuf) + lane;
            double2* p1 = reinterpret_cast<double2*>(part + (size_t)frow * ntt::XSLOTS) + lane;
#pragma unroll
            for (int q = 0; q < R / 2; q++) {
                p0[q * 64] = make_double2(ntt::modmul(x[2 * q], b0[q].x), ntt::modmul(x[2 * q + 1], b0[q].y));
                p1[q * 64] = make_double2(ntt::modmul(x[2 * q], b1[q].x), ntt::modmul(x[2 * q + 1], b1[q].y));
            }
        }
        __syncthreads();
        if (macs) load_row(i + 1 < a.steps ? i + 1 : i);            // lands during the I phase
        if (wave < 2) {
            // ---- I: component `wave`: the six rows' products summed (<= 5.97 P in all, rtfhe_ntt.hpp), inverse transform, += (trlwe.rs:49-60)
            double* set = wave ? part : set0;
            const double2* p = reinterpret_cast<const double2*>(set) + lane;
#pragma unroll
            for (int q = 0; q < R / 2; q++) {
                double2 s = p[q * 64];
#pragma unroll
                for (int j = 1; j < ROWS; j++) {
                    const double2 v = p[(size_t)j * (ntt::XSLOTS / 2) + q * 64];
                    s.x += v.x; s.y += v.y;
                }
                x[2 * q] = s.x; x[2 * q + 1] = s.y;
            }
            wave_lds_sync();                                    // all six rows are in registers before slot `wave` becomes the exchange buffer
            ntt::inverse<true>(x, twi, set + (size_t)wave * ntt::XSLOTS, lane);
            uint32_t* poly = accbuf + wave * N;
#pragma unroll
            for (int m = 0; m < R; m++) poly[lane + 64 * m] += ntt::to_torus(x[m]);
        }
        __syncthreads();
    }

    if (a.mode == MODE_BLIND_ROTATE) {
        uint32_t* o = a.out + (size_t)g * 2 * N;
        for (int c = tid; c < 2 * N; c += 64 * NW) o[c] = accbuf[c];
        return;
    }
    // sample extract index 0 (trlwe.rs:110-121): a' is written over a(X)
    uint32_t av[N / (64 * NW)];
#pragma unroll
    for (int k = 0; k < N / (64 * NW); k++) av[k] = accbuf[N + tid + 64 * NW * k];
    const uint32_t bprime = accbuf[0];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N / (64 * NW); k++) {
        const int c = tid + 64 * NW * k;
        accbuf[N + ((N - c) & (N - 1))] = (c == 0) ? av[k] : (0u - av[k]);
    }
    __syncthreads();
    if (a.mode == MODE_EXTRACT) {      // the key switch of the whole batch follows as its own launch (k_key_switch_mm)
        const int ge = a.ext_first + g;      // batch-wide gate number: the sample buffer is laid out for the key switch (ext_slot)
        for (int c = tid; c < N; c += 64 * NW) *ext_slot(a.ext, ge, c, N) = accbuf[N + c];
        if (tid == 0) *ext_slot(a.ext, ge, N, N) = bprime;
        for (int c = tid; c <= n; c += 64 * NW) io.out[c] = 0u;
        return;
    }
    // key switch: wave w sums the rows of coefficients [w N/8, (w+1) N/8); partial sums meet in LDS
    uint4 sum[KSQ];
    ks_accumulate<LOGN, KS_T, KS_BB, KSQ>(accbuf + N, wave * (N / NW), (wave + 1) * (N / NW), a.ksk, a.ksw, sum, lane);
    uint4* ksp = reinterpret_cast<uint4*>(part);          // [NW][KSQ][64] uint4 = 24 KiB
#pragma unroll
    for (int q = 0; q < KSQ; q++) ksp[(wave * KSQ + q) * 64 + lane] = sum[q];
    __syncthreads();
    const uint32_t* pw = reinterpret_cast<const uint32_t*>(part);
    for (int col = tid; col <= n; col += 64 * NW) {
        const int slot = col >> 2, q = slot >> 6, ln = slot & 63, e = col & 3;
        uint32_t s = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) s += pw[((w * KSQ + q) * 64 + ln) * 4 + e];
        io.out[col] = ((col == n) ? bprime : 0u) - s;
    }
}

}  // namespace rtfhe
