// rtfhe_kernels_wg.hpp -- latency-oriented variant of the bootstrap kernel: ONE GATE PER WORKGROUP.
//
// k_bootstrap (rtfhe_kernels.hpp) gives each gate one wavefront: best throughput once a launch has >= 4 gates per
// CU, but a gate then takes ~8.5 ms however few gates there are.  Dependency waves of a circuit (BASELINE config 4)
// and single hom_nand() calls are small; here the 8 waves of a 512-thread workgroup share one gate:
//
//   per CMUX step (same arithmetic, same operation order as the reference -- see cmux_step for citations):
//     F  waves 0..2l-3 : wave j gathers/decomposes digit polynomial j and runs its forward transform, spectrum -> LDS;
//        waves 2l-2..2l+1: the last two rows, each on TWO waves split by the parity of the point index (rtfhe_sub256.hpp): a wave gathers its
//                        parity's 256 points and runs their sub-network (4 points per lane); the size-2 stage across the parities is folded
//                        into the M phase's reads.  Every SIMD hosts one whole-row wave and one half-row wave from the start of the phase.
//                        (Round 1-3: the two rows were cut at their first exchange and handed from waves 2l-2, 2l-1 to waves 2l, 2l+1.)
//     -- barrier --
//     M  all 8 waves   : wave w owns points (lane << 3) | w of both accumulator spectra and runs their MAC chains
//                        over the 2l rows IN ROW ORDER (the reference's fold order), BK values prefetched at step start
//     -- barrier --
//     I  waves 0..3    : (component, parity): each inverse transform on two waves, again split by parity -- a lane reads the eight sums its
//                        four inputs need and applies the size-2 stage itself, so the waves trade nothing; truncate, += into the LDS accumulator
//     -- barrier --
//   key switch: each wave sums the rows of N/8 coefficients, partial sums reduced through LDS.
// Measured (profiles/r04/latency_parity_split_ab.log, same process, bit-identical): single gate 2.99 -> 2.88 ms (inverse split) -> 2.73 ms (rows
// split) -> 2.70 ms (half-row waves at priority 1 behind their second exchange).  The F phase is bound by instruction issue per SIMD (about 6.4
// cycles per instruction for the 800 a whole row plus a half row take): sixteen waves with EVERY row split -- three half rows per SIMD -- finish
// one after the other and end later (5.7 k cycles against 5.2 k; latency_sixteen_waves_ab.log).
//
// Only N = 1024 (R = 8 points per lane = 8 waves for the M phase; LDS 132 KiB).
#pragma once

#include "rtfhe_kernels.hpp"
#include "rtfhe_sub256.hpp"

// priority of the half-row waves during the F phase (they share a SIMD with a whole-row wave each: half the arithmetic, one more LDS round trip)
// ... and from its k-th exchange on (0 = never changed).  Measured (profiles/r04/latency_half_row_priority_ab.log): raised for the whole phase +7 % time;
// raised to 1 behind the second exchange -0.6 % (one gate) / -2 % (256 gates)

namespace rtfhe {

template <int LOGN, int L>
struct WgLds {
    typedef Geo<LOGN> G;
    static constexpr int NW = 8;
    // A spectrum row's slot is XSLOTS complex values wide (P of them the spectrum): the wave that produces the row uses the slot as its exchange
    // buffer first -- one 16-byte LDS access per complex value (in this latency-bound kernel half the LDS instructions are worth 1.6 % per phase,
    // profiles/r04/latency_16byte_exchanges_ab.log; in the throughput kernels they are not, see PAIR_X128) -- and the I phase's waves do the same.
    static constexpr int SROW = G::XSLOTS;
    static constexpr size_t TW = 0;
    static constexpr size_t ACC = TW + (size_t)G::TW_TOTAL * sizeof(cplx);
    static constexpr size_t SPEC = ACC + (size_t)2 * G::N * 4;                       // cplx[2l][SROW]  (also key-switch partials)
    static constexpr size_t SBUF = SPEC + (size_t)2 * L * SROW * sizeof(cplx);       // cplx[2][P]
    static constexpr size_t XBUF = SBUF + (size_t)2 * G::P * sizeof(cplx);           // cplx[4][Q4::XS]: exchange buffers of the half-row waves
    static constexpr size_t ABAR = XBUF + (size_t)4 * Q4::XS * sizeof(cplx);
    __host__ __device__ static constexpr size_t bytes(int npad) { return ABAR + (size_t)npad * 4; }
};

template <int LOGN, int L, int BGBIT, int KS_T, int KS_BB, int KSQ>
__global__ __launch_bounds__(512, 1) void k_bootstrap_wg(const BootstrapArgs a) {
    typedef Geo<LOGN> G;
    typedef WgLds<LOGN, L> S;
    constexpr int N = G::N, P = G::P, R = G::R, NW = S::NW, ROWS = 2 * L;
    static_assert(R == NW, "the MAC phase gives each of the 8 waves one of the R = 8 points a lane holds");
    static_assert(ROWS + 2 == NW, "rows 0..2l-1 start on waves 0..2l-1; the last two rows finish on waves 2l, 2l+1");
    constexpr uint32_t M = decomp_mask(L, BGBIT);
    extern __shared__ __align__(16) unsigned char smem[];
    cplx* tw = reinterpret_cast<cplx*>(smem + S::TW);
    uint32_t* accbuf = reinterpret_cast<uint32_t*>(smem + S::ACC);
    cplx* spec = reinterpret_cast<cplx*>(smem + S::SPEC);
    cplx* sbuf = reinterpret_cast<cplx*>(smem + S::SBUF);
    uint32_t* abar = reinterpret_cast<uint32_t*>(smem + S::ABAR);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const cplx* twf = tw;
    const int g = blockIdx.x;                       // grid = count
    const int n = a.n;

    for (int idx = tid; idx < G::TW_TOTAL; idx += 64 * NW) tw[idx] = a.tw[idx];
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

#ifdef RTFHE_WG_STAMPS
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
#define WG_STAMP(k) do { unsigned long long t_ = __builtin_amdgcn_s_memtime(); tsum[k] += t_ - tprev; tprev = t_; } while (0)
#else
#define WG_STAMP(k) do { } while (0)
#endif
    const size_t trgsw_cplx = (size_t)ROWS * 2 * R * 64;
    // BK values this wave needs in the M phase: point m = wave of every row and component (coalesced 1 KiB each).
    // Software-pipelined: the values of step i + 1 are requested at the start of step i's I phase (6 of the 8 waves idle
    // there) and have landed by the barrier that ends it.
    cplx bkv[ROWS][2];
    // through a buffer resource: scalar offset of (step, row, component, wave) + one per-lane VGPR (see k_bootstrap_pair)
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t bk_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<cplx*>(a.bk), 0, 0x7fffffff, 0x00020000);
    const int lane16 = lane * 16;
    auto load_bk = [&](int step, cplx (&dst)[ROWS][2]) {
        const int s0 = __builtin_amdgcn_readfirstlane((int)(((size_t)step * trgsw_cplx + (size_t)wave * 64) * sizeof(cplx)));
#pragma unroll
        for (int j = 0; j < ROWS; j++) {
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const v4u v = __builtin_amdgcn_raw_buffer_load_b128(bk_rsrc, lane16, s0 + (j * 2 + c) * R * 64 * (int)sizeof(cplx), 0);
                dst[j][c] = make_double2(__longlong_as_double(((unsigned long long)v.y << 32) | v.x), __longlong_as_double(((unsigned long long)v.w << 32) | v.z));
            }
        }
    };
    if (a.steps > 0) load_bk(0, bkv);
    // waves 0..3 = (component wave >> 1, parity wave & 1) of the I phase: their 15 twiddles stay in registers over the whole blind rotation
    // (the parity tables ride behind the table staged into LDS above)
    Q4Regs qinv;
    qinv.load(a.tw + G::TW_TOTAL + Q4Tw::off(wave >= 4 ? 0 : 1, wave & 1), lane);      // waves 4..7: the forward tables of their half row
#pragma unroll 1
    for (int i = 0; i < a.steps; i++) {
        const int r = __builtin_amdgcn_readfirstlane((int)abar[i]);
        WG_STAMP(0);
        // ---- F: one digit polynomial per row (trgsw.rs:269-289).  Six transforms on four SIMDs: rows 0..3 run whole on waves 0..3 (one per SIMD);
        // rows 4, 5 on waves 4..7 = (row, parity of the point index), one beside every whole-row wave (rtfhe_sub256.hpp).
        if (wave < ROWS - 2) {
            const int h = wave / L, jj = wave - h * L;
            const uint32_t* poly = accbuf + h * N;
            double re[R], im[R];
#pragma unroll
            for (int m = 0; m < R; m++) {
                const int c0 = lane + 64 * m, c1 = c0 + P;
                const uint32_t d0 = rotated_coef<LOGN>(poly, c0, r) - poly[c0];
                const uint32_t d1 = rotated_coef<LOGN>(poly, c1, r) - poly[c1];
                re[m] = (double)decomp_digit((d0 + M) ^ M, BGBIT, jj);
                im[m] = (double)decomp_digit((d1 + M) ^ M, BGBIT, jj);
            }
            // exchange buffer = this row's own (still unwritten) spectrum slot, one 16-byte access per complex value
            fft_forward<LOGN, 2, BOOT_TRIV>(re, im, twf, reinterpret_cast<double*>(spec + (size_t)wave * S::SROW), lane);
            cplx* dst = spec + (size_t)wave * S::SROW + lane;
#pragma unroll
            for (int m = 0; m < R; m++) dst[m * 64] = make_double2(re[m], im[m]);
        } else {
            // rows 2l-2, 2l-1: wave = (row, parity H); this wave's points are i = 2 (lane + 64 m) + H.  Its half spectrum out_H[j], j = 4 lane + m, goes to
            // words [256 H, 256 H + 256) of the row's slot, index m * 64 + lane; the M phase adds / subtracts the two halves as it reads them.
            const int k = wave - (ROWS - 2), row = (ROWS - 2) + (k >> 1);
            const int h = row / L, jj = row - h * L;
            const uint32_t* poly = accbuf + h * N;
            cplx* half = spec + (size_t)row * S::SROW + (k & 1) * (P / 2);
            cplx* xc = reinterpret_cast<cplx*>(smem + S::XBUF) + (size_t)k * Q4::XS;
            __builtin_amdgcn_s_setprio(0);
            auto run = [&](auto odd) {
                constexpr bool ODD = decltype(odd)::value;
                double re[4], im[4];
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    const int c0 = 2 * (lane + 64 * m) + (ODD ? 1 : 0), c1 = c0 + P;
                    const uint32_t d0 = rotated_coef<LOGN>(poly, c0, r) - poly[c0];
                    const uint32_t d1 = rotated_coef<LOGN>(poly, c1, r) - poly[c1];
                    re[m] = (double)decomp_digit((d0 + M) ^ M, BGBIT, jj);
                    im[m] = (double)decomp_digit((d1 + M) ^ M, BGBIT, jj);
                }
                sub256_forward<ODD, BOOT_TRIV>(re, im, qinv, xc, lane, [](int k) { if (k == 2) __builtin_amdgcn_s_setprio(1); });
#pragma unroll
                for (int m = 0; m < 4; m++) half[m * 64 + lane] = make_double2(re[m], im[m]);
            };
            if (k & 1) run(std::true_type{}); else run(std::false_type{});
            __builtin_amdgcn_s_setprio(0);
        }
        WG_STAMP(1);
        __syncthreads();
        WG_STAMP(2);
        // ---- M: hadamard + fold-add from zero in row order (spqlios.rs:204-222, trgsw.rs:290-299), point m = wave ----
        {
            double s0r = 0.0, s0i = 0.0, s1r = 0.0, s1i = 0.0;
            const cplx* src = spec + wave * 64 + lane;
#pragma unroll
            for (int j = 0; j < ROWS; j++) {
                cplx d;
                if (j < ROWS - 2) d = src[(size_t)j * S::SROW];
                else {      // point 8 lane + wave = 2 jq + (wave & 1), jq = 4 lane + (wave >> 1): out_0[jq] + out_1[jq] or out_0[jq] + (-out_1[jq])
                    const cplx* hs = spec + (size_t)j * S::SROW + (wave >> 1) * 64 + lane;
                    const cplx e = hs[0], o = hs[P / 2];
                    d = (wave & 1) ? make_double2(e.x + (-o.x), e.y + (-o.y)) : make_double2(e.x + o.x, e.y + o.y);
                }
                {
                    const double ii = bkv[j][0].y * d.y, rr = bkv[j][0].x * d.x, ri = bkv[j][0].x * d.y, ir = bkv[j][0].y * d.x;
                    s0r = s0r + (rr - ii);
                    s0i = s0i + (ir + ri);
                }
                {
                    const double ii = bkv[j][1].y * d.y, rr = bkv[j][1].x * d.x, ri = bkv[j][1].x * d.y, ir = bkv[j][1].y * d.x;
                    s1r = s1r + (rr - ii);
                    s1i = s1i + (ir + ri);
                }
            }
            sbuf[wave * 64 + lane] = make_double2(s0r, s0i);
            sbuf[P + wave * 64 + lane] = make_double2(s1r, s1i);
        }
        WG_STAMP(3);
        __syncthreads();
        WG_STAMP(4);
        cplx bkn[ROWS][2];
        load_bk(i + 1 < a.steps ? i + 1 : i, bkn);
        // ---- I: each component's inverse transform (math.rs:279-288) on two waves, one per parity of the point index; += (trlwe.rs:49-60) ----
        // A lane reads the eight sums s[8 lane .. 8 lane + 7] its parity's four inputs need (both parities read the same words): the size-2 stage
        // across the parities is computed here, in_0[j] = s[2j] + s[2j + 1], in_1[j] = s[2j] + (-s[2j + 1]), j = 4 lane + m -- no trade between the waves.
        if (wave < 4) {
            const int comp = wave >> 1;
            const cplx* src = sbuf + (size_t)comp * P + lane;
            cplx v[R];
#pragma unroll
            for (int m = 0; m < R; m++) v[m] = src[m * 64];
            // the spectra are dead after the M phase: slot `wave` holds this wave's exchange buffers
            cplx* xc = spec + (size_t)wave * S::SROW;
            uint32_t* poly = accbuf + comp * N;
            auto run = [&](auto odd) {
                constexpr bool ODD = decltype(odd)::value;
                double re[4], im[4];
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    re[m] = ODD ? v[2 * m].x + (-v[2 * m + 1].x) : v[2 * m].x + v[2 * m + 1].x;
                    im[m] = ODD ? v[2 * m].y + (-v[2 * m + 1].y) : v[2 * m].y + v[2 * m + 1].y;
                }
                sub256_inverse<ODD, BOOT_TRIV>(re, im, qinv, xc, lane);
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    const int c = 2 * (lane + 64 * m) + (ODD ? 1 : 0);
                    poly[c] += trunc_to_torus(re[m]);
                    poly[c + P] += trunc_to_torus(im[m]);
                }
            };
            if (wave & 1) run(std::true_type{}); else run(std::false_type{});
        }
        WG_STAMP(5);
        __syncthreads();
        WG_STAMP(6);
#pragma unroll
        for (int j = 0; j < ROWS; j++) { bkv[j][0] = bkn[j][0]; bkv[j][1] = bkn[j][1]; }
    }
#ifdef RTFHE_WG_STAMPS
    if (a.dbg && blockIdx.x == 0 && lane == 0)
        for (int k = 0; k < 8; k++) a.dbg[wave * 8 + k] = tsum[k];
#endif

    if (a.mode == MODE_BLIND_ROTATE) {
        uint32_t* o = a.out + (size_t)g * 2 * N;
        for (int c = tid; c < 2 * N; c += 64 * NW) o[c] = accbuf[c];
        return;
    }

    // sample extract index 0 (trlwe.rs:110-121) into the now free abar/spec area is not needed: a' is written over a(X)
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
    uint4* part = reinterpret_cast<uint4*>(spec);          // [NW][KSQ][64] uint4 = 24 KiB
#pragma unroll
    for (int q = 0; q < KSQ; q++) part[(wave * KSQ + q) * 64 + lane] = sum[q];
    __syncthreads();
    uint32_t* out = io.out;
    const uint32_t* pw = reinterpret_cast<const uint32_t*>(spec);
    for (int col = tid; col <= n; col += 64 * NW) {
        // column col lives in uint4 slot (col/4) = lane + 64 q, element col % 4
        const int slot = col >> 2, q = slot >> 6, ln = slot & 63, e = col & 3;
        uint32_t s = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) s += pw[((w * KSQ + q) * 64 + ln) * 4 + e];
        out[col] = ((col == n) ? bprime : 0u) - s;
    }
}

}  // namespace rtfhe
