// rtfhe_kernels_eo.hpp -- N = 2048 (BASELINE config 5): two waves per transform, split by the PARITY of the point index.
//
// At N = 2048 a polynomial's transform has 1024 complex points: 16 per lane in one wavefront, which with three digit rows side by side does not
// fit the register file (k_bootstrap<11>, one wave per gate, lives in 512 registers with ~790 AGPR shuffles per step at the lone-wave issue rate).
// So the two waves of a gate share every transform.  Wave H owns the points of parity H: i = 2 j + H, j in [0, 512).  Every radix-2 stage of the
// reference network pairs i with i + halfnn (forward, decimation in frequency, spqlios-fft-impl.cpp:526-572; inverse, decimation in time,
// :315-363), and for halfnn >= 2 both have the same parity: NINE of the ten stages stay inside a wave -- twist, halfnn = 512 ... 4 (eight
// twiddled stages), and this parity's half of the size-4 stage (:575-603 / :289-310) -- as a 512-point network in j of exactly the shape the
// N = 1024 kernels run (8 points per lane, three in-register passes of three stages, two wave-private exchanges), with the twiddle of pair
// (i, i + halfnn) = entry (i mod halfnn) = 2 (j mod halfnn/2) + H of the reference's stage table.  Only the size-2 stage (halfnn = 1: x0 + x1,
// x0 + (-x1), no twiddle, :606-634 / :248-269) crosses the waves: it is the LAST stage of the forward transform and the FIRST of the inverse.
//   * both waves execute the same instruction stream (wave 1's only differences: table pointers, and "partner + (-mine)" where wave 0 has
//     "mine + partner") -- the work is balanced by construction;
//   * the forward trade comes after a row's pass 3: the NEXT row's pass 3 (and, for the last row, the first multiply-accumulates) run between
//     a row's arrival flag and the wait for the partner's -- no wave sits at a synchronisation with nothing to issue;
//   * gather, decomposition and twist are wave-private: no first-stage trade at all;
//   * the stage across the waves sits right next to the pointwise multiply-accumulate, whose layout is free: each wave finishes BOTH outputs of the
//     butterflies of half of the indices k (it sends 4 of its 8 values per lane and receives 4: half-width trade), and the key is stored in that
//     layout (k_bk_to_eo).
// Each wave owns the spectrum points it finished for the multiply-accumulate over all six rows and both components: the fold order
// (trgsw.rs:290-299) holds trivially, no partial sums travel.  Buffer OWNERSHIP ping-pongs between the two waves: after a trade each wave owns the
// buffer it has just read -- its previous writer is done with it (its writes precede its arrival flag), its reader is this wave itself (DS
// instructions of a wave execute in order) -- so there is no "buffer free" synchronisation: ONE arrival / wait per trade, 3 per polynomial + 1 per
// inverse = 8 per step.  Same arithmetic DAG as the reference, every product and sum rounded on its own (-ffp-contract=off): bit-identical to
// k_bootstrap<11> (tests/test_gpu_configs.py::test_config5_*).
// LDS: the per-parity stage tables of passes 1-3 (16 KiB forward, 2 KiB inverse); twist / untwist / inverse pass-1 tables in global memory
// (a vector-memory read costs the SIMD's issue less than an LDS read does, and LDS is full: profiles/r05/vmem_issue.log).
// The accumulator lives in LDS as PARITY PLANES: coefficient c of a polynomial at word 1024 (c & 1) + (c >> 1) of its 8 KiB.  Wave H reads and
// updates the coefficients of parity H only (c = 2 (lane + 64 k) + H), and the rotated coefficients it gathers all have the parity of H - r
// (wave-uniform): every access of a wave goes to 64 CONSECUTIVE words of one plane -- no LDS bank conflict (the natural layout's stride of
// two words made every access two-way conflicted: 7.3 % of the LDS pipe's active cycles, profiles/r04/pmc_n2048_eo.json) -- and with the
// planes 4 KiB-aligned the address of a rotated coefficient is one add, one and-or from a per-polynomial lane constant.
// The kernel this one replaced (split by the TOP index bit, rounds 2-4) and every variant measured on the way are in profiles/HISTORY.md.
#pragma once

#include <type_traits>

#include "rtfhe_kernels_pair.hpp"

namespace rtfhe {

// twiddle table of the even / odd kernel, cplx units, everything [parity 0 | parity 1]
struct EoTw {
    static constexpr int TWIST = 0;                   // [2][8][64]  point i = 2 (lane + 64 m) + H                                   (global memory)
    static constexpr int P1 = TWIST + 2 * 8 * 64;     // [2][7][64]  pass 1: j-halfnn 256, 128, 64, entry layout of Geo<10>::TW_P1      (LDS from here ...
    static constexpr int P2 = P1 + 2 * 7 * 64;        // [2][7][8]   pass 2: j-halfnn 32, 16, 8
    static constexpr int P3 = P2 + 2 * 7 * 8;         // [2][8]      pass 3: j-halfnn 4 (4 entries), 2 (2 entries), 2 pad
    static constexpr int IP2 = P3 + 2 * 8;            // [2][7][8]   inverse pass 2
    static constexpr int IP3 = IP2 + 2 * 7 * 8;       // [2][8]      inverse pass 3                                                    ... to here)
    static constexpr int IP1 = IP3 + 2 * 8;           // [2][7][64]  inverse pass 1                                                    (global memory)
    static constexpr int IUNTW = IP1 + 2 * 7 * 64;    // [2][8][64]  untwist, times 2/N                                               (global memory)
    static constexpr int TOTAL = IUNTW + 2 * 8 * 64;
    static constexpr int LDS_CPLX = IP1 - P1;
};

struct EoLds {
    typedef Geo<10> G;   // geometry of a parity's 512-point sub-network
    // layout: [accumulators of all gates: 16 KiB each, so that every parity plane is 4 KiB-aligned][stage tables][per gate: rotation amounts, two
    // exchange buffer pairs, arrival counters]
    static constexpr size_t ACC = (size_t)2 * 2048 * 4;
    static constexpr size_t TW = (size_t)EoTw::LDS_CPLX * sizeof(cplx);
    static constexpr size_t XB = (size_t)2 * G::XSLOTS * sizeof(double);            // one wave's re + im exchange buffers: hold 512 cplx
    static constexpr size_t FLAGS = 16;       // two arrival counters per gate
    __host__ __device__ static constexpr size_t abar_bytes(int npad) { return ((size_t)npad * 2 + 15) / 16 * 16; }      // rotation amounts as u16
    __host__ __device__ static constexpr size_t rest_bytes(int npad) { return abar_bytes(npad) + 2 * XB + FLAGS; }
    __host__ __device__ static constexpr size_t gate_bytes(int npad) { return ACC + rest_bytes(npad); }
    __host__ __device__ static constexpr size_t bytes(int gates, int npad) { return TW + (size_t)gates * gate_bytes(npad); }
    // word of coefficient c (0 <= c < 2048) inside a polynomial's 2048 words
    __host__ __device__ static constexpr int plane_word(int c) { return ((c & 1) << 10) | (c >> 1); }
};

struct EoArgs {
    BootstrapArgs b;       // b.tw / b.bk unused
    const cplx* etw;       // [EoTw::TOTAL]
    const cplx* ebk;       // [n][2l rows][2 comp][2 waves][8][64]: k_bk_to_eo
};

// key spectra: device layout of k_bootstrap<11> ([n][row][comp][16][64]: lane v, register q <-> point (v << 4) | q) -> the layout the waves of
// k_bootstrap_eo hold their spectra in after the stage across them.  A parity's sub-network leaves lane v, register m with its output k = 8 v + m;
// the stage across the waves makes points 2k (sum) and 2k + 1 (difference).  Wave H finishes both for m = 4 H + j, j < 4: register j holds point
// 2k = (v << 4) | (8 H + 2 j), register 4 + j point 2k + 1.
__global__ __launch_bounds__(256) void k_bk_to_eo(const cplx* __restrict__ src, cplx* __restrict__ dst, size_t polys) {
    const size_t total = polys * 1024;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t g = idx >> 10;
        const int k = (int)(idx & 1023);                 // destination: (H, register, lane)
        const int H = k >> 9, m = (k >> 6) & 7, lane = k & 63;
        const int q = 8 * H + 2 * (m & 3) + (m >> 2);
        dst[idx] = src[g * 1024 + (size_t)q * 64 + lane];
    }
}

// pass 3 of a parity's sub-network, forward: i-halfnn 8 and 4 (twiddled, wave-uniform entries w[0..3] / w[4..5]), then this parity's half of the
// size-4 stage (spqlios-fft-impl.cpp:581-602): even points x0, x2 -> x0 + x2, x0 + (-x2); odd points x1, x3 -> x1 + x3, i (x1 - x3) = ((-j1) + j3, r1 + (-r3))
template <int R, bool ODD>
__device__ __forceinline__ void eo_fwd_pass3(double (&re)[R], double (&im)[R], const cplx* w) {
    fwd_stage_tw<R, 2, BOOT_TRIV && !ODD>(re, im, w);          // entry 0 of parity 0 is the reference's (1, 0): see fwd_stage_tw
    fwd_stage_tw<R, 1, BOOT_TRIV && !ODD>(re, im, w + 4);
#pragma unroll
    for (int m = 0; m < R; m += 2) {
        const double ra = re[m], rb = re[m + 1], ja = im[m], jb = im[m + 1];
        if (!ODD) { re[m] = ra + rb; re[m + 1] = ra + (-rb); im[m] = ja + jb; im[m + 1] = ja + (-jb); }
        else      { re[m] = ra + rb; re[m + 1] = (-ja) + jb; im[m] = ja + jb; im[m + 1] = ra + (-rb); }
    }
}
// ... inverse: this parity's half of the size-4 stage (:289-310): even x0, x2 -> x0 + x2, x0 + (-x2); odd x1, x3 -> x1 - i x3 = (r1 + j3, j1 + (-r3)),
// x1 + i x3 = (r1 + (-j3), j1 + r3); then i-halfnn 4 and 8
template <int R, bool ODD>
__device__ __forceinline__ void eo_inv_pass3(double (&re)[R], double (&im)[R], const cplx* w) {
#pragma unroll
    for (int m = 0; m < R; m += 2) {
        const double ra = re[m], rb = re[m + 1], ja = im[m], jb = im[m + 1];
        if (!ODD) { re[m] = ra + rb; re[m + 1] = ra + (-rb); im[m] = ja + jb; im[m + 1] = ja + (-jb); }
        else      { re[m] = ra + jb; re[m + 1] = ra + (-jb); im[m] = ja + (-rb); im[m + 1] = ja + rb; }
    }
    inv_stage_tw<R, 1, false, BOOT_TRIV && !ODD>(re, im, w + 4);
    inv_stage_tw<R, 2, false, BOOT_TRIV && !ODD>(re, im, w);
}

template <int L, int BGBIT, int KS_T, int KS_BB, int KSQ, int GATES>
__global__ __launch_bounds__(128 * GATES, 1) void k_bootstrap_eo(const EoArgs ea) {
    constexpr int LOGN = 11, N = 2048, R = 8, NT = 128 * GATES;
    typedef Geo<10> G;   // geometry of a parity's 512-point sub-network
    constexpr uint32_t M = decomp_mask(L, BGBIT);
    static_assert(L == 3, "three digit rows of a polynomial are transformed side by side");
    const BootstrapArgs& a = ea.b;
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane0 = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slot = wave % GATES;          // the two parities of a gate share a SIMD (waves w, w + GATES)
    const int H = wave / GATES;
    cplx* tw = reinterpret_cast<cplx*>(smem + (size_t)GATES * EoLds::ACC);
    for (int idx = tid; idx < EoTw::LDS_CPLX; idx += NT) tw[idx] = ea.etw[EoTw::P1 + idx];
    // this parity's tables, addressed with Geo<10>'s per-direction offsets where a device function expects them
    const cplx* tw_fwd12 = tw + (size_t)H * 7 * 64 - G::TW_P1;                                   // + G::TW_P1 -> P1[H]; P2 is not contiguous with it here:
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

    uint32_t* accbuf = reinterpret_cast<uint32_t*>(smem + (size_t)slot * EoLds::ACC);        // [2 polynomials][2 parity planes][1024]
    unsigned char* gbase = smem + (size_t)GATES * EoLds::ACC + EoLds::TW + (size_t)slot * EoLds::rest_bytes(a.npad);
    uint16_t* abar = reinterpret_cast<uint16_t*>(gbase);
    double* xb0 = reinterpret_cast<double*>(gbase + EoLds::abar_bytes(a.npad));
    double* xb1 = xb0 + 2 * G::XSLOTS;
    double* wbuf = H ? xb1 : xb0;     // the buffer pair this wave owns (writes next); ownership swaps after every trade
    double* rbuf = H ? xb0 : xb1;     // the partner's (read after its arrival)
    uint32_t* flags = reinterpret_cast<uint32_t*>(gbase + EoLds::rest_bytes(a.npad) - EoLds::FLAGS);
    // LDS byte address of this gate's accumulator: the and-or addressing of the gather needs the planes 4 KiB-aligned (dynamic LDS starts at 0)
    const unsigned acc_lds = (unsigned)(size_t)(__attribute__((address_space(3))) uint32_t*)accbuf;
    if (acc_lds & 4095u) { if (tid == 0 && a.fault) *a.fault = 1; return; }
    if (lane0 == 0) flags[H] = 0u;
    const unsigned my_flag = (unsigned)(size_t)(__attribute__((address_space(3))) uint32_t*)(flags + H);
    const unsigned partner_flag = (unsigned)(size_t)(__attribute__((address_space(3))) uint32_t*)(flags + (1 - H));
    unsigned sync_k = 0;
#define EO_ARRIVE() pair_arrive(my_flag, ++sync_k)
    // Priority staircase: both waves run the same code, and of two ready waves a SIMD serves the higher priority (then the older) almost
    // exclusively -- the leader of a stretch between two trades then idles at the next trade while its partner finishes alone.  Here a wave's
    // priority falls 3 -> 0 along every stretch (EO_STEP at fixed code points) and is back at 3 after every wait: whichever wave is BEHIND is in
    // an earlier, higher-priority region, so the SIMD favours it until it has caught up.
    // Where the raise sits matters to the COMPILER: every s_setprio is a scheduling boundary.  At 3-4 gates per workgroup (256 registers per
    // wave) a raise as its own statement behind each of the eight waits costs 15-30 spilled registers, whose reloads queue with the key rows;
    // inside the wait's own assembly statement (pair_wait_opaque_prio3) it costs none: 16.17 -> 15.49 ms per 1024 gates, and the parity split
    // then beats the top-bit split (15.69).  At 1-2 gates per workgroup (512 registers, nothing spills) the separate statement measured
    // faster (10.66 vs 10.89 ms per 512 gates).  Measured in profiles/r04/n2048_parity_split_ab.log.
    constexpr bool FUSED_RAISE = GATES >= 3;
    auto eo_prio = [&](auto level) { __builtin_amdgcn_s_setprio(decltype(level)::value); };
#define EO_STEP(k) eo_prio(std::integral_constant<int, k>{})
    auto eo_wait = [&]() {
        if constexpr (FUSED_RAISE) pair_wait_opaque_prio3(partner_flag, sync_k);
        else { pair_wait_opaque(partner_flag, sync_k); EO_STEP(3); }
    };
#define EO_WAIT() eo_wait()

    const int n = a.n;
    {   // pre-step + mod switch (tfhe.rs:41-71, 97, 107-108) to [0, 2N)
        constexpr int SH = 32 - LOGN - 1;
        for (int i = lane0 + 64 * H; i <= n; i += 128) {
            const uint32_t t = gate_linear(io.op, io.p0[i], io.p1[i], i == n);
            abar[i] = (uint16_t)((i == n) ? (t >> SH) : ((t + (1u << (SH - 1))) >> SH));
        }
    }
    __syncthreads();
    {   // acc = X^{-bbar} * testvec (tfhe.rs:85, 98-106); each wave initialises half of the words
        const int bbar = (int)abar[n];
        for (int c = lane0 + 64 * H; c < 2 * N; c += 128) {
            const int e = (c + bbar) & (2 * N - 1);
            accbuf[(c & N) | EoLds::plane_word(c & (N - 1))] = c < N ? ((e >> LOGN) ? 0xE0000000u : 0x20000000u) : 0u;
        }
    }
    __syncthreads();

    // key rows in consumption order rc = 0..11 = (row rc / 2, component rc & 1) of this parity: the order of the layout.  Two buffers.
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
    if (H) __builtin_amdgcn_s_setprio(0);
#ifdef RTFHE_WG_STAMPS
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
#define EO_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tsum[k] += t_ - tprev; tprev = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define EO_STAMP(k) do { } while (0)
#endif

    // The size-2 stage across the waves, forward (the LAST stage: sub-network outputs out_E[k], out_O[k] -> points 2k = out_E + out_O, 2k + 1 =
    // out_E + (-out_O)).  A lane holds k = 8 v + m, m < 8, of its parity.  The even wave finishes both points for m < 4, the odd wave for m >= 4:
    // each sends the four values the other needs and receives four -- half the LDS traffic of "every wave finishes its parity of every k", the
    // same sums.  Afterwards register j < 4 holds point 2k, register 4 + j point 2k + 1, k = 8 v + 4 H + j (the key's layout: k_bk_to_eo).
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
    // ... inverse (the FIRST stage: points 2k, 2k + 1 -> in_E[k] = sum, in_O[k] = difference): both inputs of a butterfly are in one lane (registers
    // j, 4 + j); the even wave keeps the sums, the odd wave the differences, and the four they do not keep go to the partner.
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
        constexpr int RECV = decltype(odd)::value ? 0 : R / 2;      // even: the odd wave's sums for m >= 4; odd: the even wave's differences for m < 4
#pragma unroll
        for (int j = 0; j < R / 2; j++) { re[RECV + j] = lds_ld(&rb[ln + 64 * j]); im[RECV + j] = lds_ld(&rb[G::XSLOTS + ln + 64 * j]); }
    };

    // The whole step loop exists twice, once per parity, chosen ONCE (the waves of a workgroup meet at no barrier inside it): with the parity a
    // compile-time constant each copy is straight-line code.  A wave-uniform branch on H around the few places that differ (the size-4 half
    // stages, "mine + partner" against "partner + (-mine)") made the register allocator spill 232 of the 256 registers.
    auto steps = [&](auto parity) {
    constexpr bool ODD = decltype(parity)::value;
#pragma unroll 1
    for (int i = 0; i < a.steps; i++) {
        const int r = __builtin_amdgcn_readfirstlane((int)abar[i]);
        double s0re[R], s0im[R], s1re[R], s1im[R];
#pragma unroll
        for (int m = 0; m < R; m++) { s0re[m] = 0.0; s0im[m] = 0.0; s1re[m] = 0.0; s1im[m] = 0.0; }

#pragma unroll 1
        for (int h = 0; h < 2; h++) {
            const uint32_t* poly = accbuf + h * N;
            int ln = lane0;
            asm volatile("" : "+v"(ln));        // keeps the lane-derived LDS addresses from being hoisted out of the loops and spilled
            EO_STAMP(6);
            EO_STEP(2);
            // this lane's 8 complex inputs are points i = 2 (ln + 64 m) + H: coefficients i (real part) and i + 1024 (imaginary part)
            // (rotate: math.rs:85-132; decomposition: math.rs:300-326; twist: spqlios-fft-impl.cpp:496-518)
            cplx tH[R];        // twist factors from global memory: requested before the gather they land under
#pragma unroll
            for (int m = 0; m < R; m++) tH[m] = gtwist0[m * 64 + ln];
            uint32_t ure[R], uim[R];
            {
                // rotated gather (math.rs:85-132) and decomposition offset (math.rs:300-326).  Coefficient c = 2 (ln + 64 k) + H, k < 16 (k >= 8: the
                // imaginary parts, c + 1024).  (X^r p)[c] = +- p[(c - r) mod N], and c - r = 2 (ln + 64 k + s) + q with q = (H - r) & 1, s = (H - r) >> 1:
                // plane q, word (ln + s + 64 k) & 1023, negated iff bit 10 of ln + s + 64 k is set.  Own coefficient: plane H, word ln + 64 k.
                const int hr = H - r;
                unsigned rot_plane = acc_lds + (unsigned)h * (N * 4) + (unsigned)(hr & 1) * 4096u;      // wave-uniform; the and-or takes it from a VGPR
                asm volatile("" : "+v"(rot_plane));
                const unsigned tb = (unsigned)(ln + (hr >> 1)) * 4u;
                const uint32_t* own = poly + H * 1024 + ln;
                // the 16 own coefficients first (8 two-word reads whose addresses need no arithmetic: their latency covers the address arithmetic of
                // the rotated reads), then the 16 rotated reads, then the arithmetic: 6 integer instructions per coefficient up to the decomposition
                // offset (add, and-or, sign, two subtractions, xor-add) where the natural layout took 9
                uint32_t v[2 * R], sg[2 * R], mo[2 * R];
#pragma unroll
                for (int k = 0; k < 2 * R; k++) mo[k] = own[64 * k];
                __builtin_amdgcn_sched_barrier(0);
                unsigned addr[2 * R];
#pragma unroll
                for (int k = 0; k < 2 * R; k++) {
                    const unsigned t = tb + 256u * k;
                    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(addr[k]) : "v"(t), "s"(0xffcu), "v"(rot_plane));
                    sg[k] = (uint32_t)((int32_t)(t << 19) >> 31);       // all ones iff bit 12 of the byte offset = bit 10 of the word index
                }
                __builtin_amdgcn_sched_barrier(0);     // every rotated read is issued before the first is waited for (the scheduler otherwise waits read by read)
#pragma unroll
                for (int k = 0; k < 2 * R; k++) v[k] = *reinterpret_cast<const __attribute__((address_space(3))) uint32_t*>(addr[k]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 2 * R; k++) {
                    mo[k] = M - mo[k];
                    asm("" : "+v"(mo[k]));      // (keeps "(M - own) - sign" from being re-associated into one more bit-field extract and an or)
                }
#pragma unroll
                for (int k = 0; k < 2 * R; k++) {
                    uint32_t x;                                                     // (+-v - own) + M = (v ^ sign) + ((M - own) - sign)
                    asm("v_xad_u32 %0, %1, %2, %3" : "=v"(x) : "v"(v[k]), "v"(sg[k]), "v"(mo[k] - sg[k]));
                    const uint32_t u = x ^ M;
                    if (k < R) ure[k] = u; else uim[k - R] = u;
                }
            }
            EO_STAMP(0);
            double yr[L][R], yi[L][R];
#pragma unroll
            for (int jj = 0; jj < L; jj++)
#pragma unroll
                for (int m = 0; m < R; m++) {
                    const double a0 = (double)decomp_digit(ure[m], BGBIT, jj), b0 = (double)decomp_digit(uim[m], BGBIT, jj);
                    const double rc = a0 * tH[m].x, ic = b0 * tH[m].x, rs = a0 * tH[m].y, is = b0 * tH[m].y;
                    yr[jj][m] = rc - is; yi[jj][m] = ic + rs;
                }
            // passes 1 and 2 of the three rows side by side (twiddles of this parity), both wave-private exchanges
            {
                Tw<R - 1> w1;
                w1.load(tw_fwd12 + G::TW_P1 + ln, 64);
#pragma unroll
                for (int jj = 0; jj < L; jj++) {
                    P12<R, G::LR - 1>::fwd(yr[jj], yi[jj], w1.w);
                    exchange<10, 1, 2, true>(yr[jj], yi[jj], wbuf, ln);
                }
                Tw<R - 1> w2;
                w2.load(tw_p2 + (ln & (G::NLOW - 1)), G::NLOW);
                EO_STEP(1);
#pragma unroll
                for (int jj = 0; jj < L; jj++) {
                    P12<R, G::LR - 1>::fwd(yr[jj], yi[jj], w2.w);
                    exchange<10, 2, 3, true>(yr[jj], yi[jj], wbuf, ln);
                }
            }
            EO_STAMP(1);
            const int rc0 = h * 2 * L;                  // rc = 2 * row + comp
            fetch(bA, i, rc0);                          // (row 0, c0): in flight under pass 3 and the trades
            // pass 3 row by row; a row's values go to the partner right behind it and the NEXT row's pass 3 (for the last row: the first
            // multiply-accumulates) runs between the arrival flag and the wait.  The buffers swap owners after every trade (ping-pong, see
            // the header): row 0 is written to my buffer, row 1 to the one I read row 0 from, row 2 to the one I read row 1 from.
            Tw<6> w3;
            w3.load(tw_p3, 1);
            EO_STEP(0);
            auto pass3 = [&](int jj) { eo_fwd_pass3<R, ODD>(yr[jj], yi[jj], w3.w); };
            pass3(0);
            cross_write(parity, yr[0], yi[0], wbuf, ln); EO_ARRIVE();
            pass3(1);
            EO_WAIT(); cross_read(parity, yr[0], yi[0], rbuf, ln);
            cross_write(parity, yr[1], yi[1], rbuf, ln); EO_ARRIVE();
            pass3(2);
            EO_WAIT(); cross_read(parity, yr[1], yi[1], wbuf, ln);
            cross_write(parity, yr[2], yi[2], wbuf, ln); EO_ARRIVE();
            fetch(bB, i, rc0 + 1);                      // (row 0, c1): requested once pass 3's twiddles are dead (both buffers live through pass 3 spill)
            EO_STAMP(2);
            // hadamard + fold-add (spqlios.rs:204-222, trgsw.rs:290-299): this wave's parity of the points; each accumulator folds the
            // polynomial's rows in order (and over the step: rows 0..5 in order); two key-row buffers, refilled as a multiply-accumulate retires
            mac_row<R>(s0re, s0im, bA, yr[0], yi[0]); fetch(bA, i, rc0 + 2);           // (row 1, c0)
            mac_row<R>(s1re, s1im, bB, yr[0], yi[0]); fetch(bB, i, rc0 + 3);           // (row 1, c1)
            mac_row<R>(s0re, s0im, bA, yr[1], yi[1]); fetch(bA, i, rc0 + 4);           // (row 2, c0)
            mac_row<R>(s1re, s1im, bB, yr[1], yi[1]); fetch(bB, i, rc0 + 5);           // (row 2, c1)
            EO_WAIT(); cross_read(parity, yr[2], yi[2], rbuf, ln);
            { double* t = wbuf; wbuf = rbuf; rbuf = t; }        // three trades: I now own the buffer I read last
            mac_row<R>(s0re, s0im, bA, yr[2], yi[2]);
            mac_row<R>(s1re, s1im, bB, yr[2], yi[2]);
            EO_STAMP(3);
        }

        // inverse: the size-2 stage across the waves comes FIRST (decimation in time), then this parity's sub-network, untwist, truncate, += acc
        // At 1-3 gates per workgroup the two components are two copies of the code (no selects of the 32 accumulator registers: 0.5-1.1 % faster,
        // profiles/r04/n2048_unrolled_components_ab.log); at 4 the copy measured +-0 and the loop keeps 11 KiB of instruction cache free.
        constexpr int COMP_COPIES = GATES <= 3 ? 2 : 1;
#pragma unroll COMP_COPIES
        for (int comp = 0; comp < 2; comp++) {
            if (COMP_COPIES == 2) __builtin_amdgcn_sched_barrier(0);      // the copies one after the other, not interleaved
            double re[R], im[R];
#pragma unroll
            for (int m = 0; m < R; m++) { re[m] = comp ? s1re[m] : s0re[m]; im[m] = comp ? s1im[m] : s0im[m]; }
            int lane = lane0;
            asm volatile("" : "+v"(lane));
            inv_cross_write(parity, re, im, wbuf, lane); EO_ARRIVE();
            Tw<6> w3; Tw<R - 1> w2, w1; Tw<R> wt;
            w1.load(gip10 + lane, 64);                  // global memory: requested first, used last
#pragma unroll
            for (int m = 0; m < R; m++) wt.w[m] = guntw0[m * 64 + lane];
            w3.load(twi_p3, 1);
            w2.load(twi_p2 + (lane & (G::NLOW - 1)), G::NLOW);
            EO_WAIT(); inv_cross_read(parity, re, im, rbuf, lane);
            { double* t = wbuf; wbuf = rbuf; rbuf = t; }
            EO_STAMP(4);
            eo_inv_pass3<R, ODD>(re, im, w3.w);
            exchange<10, 3, 2, true>(re, im, wbuf, lane);
            EO_STEP(2);
            P12<R, G::LR - 1>::inv(re, im, w2.w);
            exchange<10, 2, 1, true>(re, im, wbuf, lane);
            EO_STEP(1);
            P12<R, G::LR - 1>::inv(re, im, w1.w);
            {
                uint32_t* poly = accbuf + comp * N;
#pragma unroll
                for (int m = 0; m < R; m++) {
                    const double vr = re[m], vi = im[m];
                    // (re, im) * (c, s): re c - im s, im c + re s   (spqlios-fft-impl.cpp:390-395); the 2/N of fft_processor_spqlios.cpp:158 is in the table
                    const double rc = vr * wt.w[m].x, ic = vi * wt.w[m].x, rs = vr * wt.w[m].y, is = vi * wt.w[m].y;
                    // coefficients 2 (lane + 64 m) + H and + 1024: words lane + 64 m and + 512 of this parity's plane.  ds_add_u32: no read-back through
                    // the wave (14.59 -> 14.52 ms per 1024 gates; the latency kernel, whose lone waves wait on the add's completion, loses 2 % with it)
                    uint32_t* w = poly + H * 1024 + lane + 64 * m;
                    __hip_atomic_fetch_add(w, trunc_to_torus(rc - is), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    __hip_atomic_fetch_add(w + 512, trunc_to_torus(ic + rs), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                }
            }
            // (my accumulator words are published by my next arrival -- the other component's trade / the next step's first row -- which the
            // partner waits for before it gathers them)
            EO_STAMP(5);
        }
    }
    };
    if (H) steps(std::true_type{}); else steps(std::false_type{});
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();      // the last accumulator update has no arrival behind it: both parities' words must be visible below
    {   // parity planes -> natural coefficient order for what follows (output, sample extract, key switch): each thread of the gate moves 32 words
        uint32_t nat[4 * R];
#pragma unroll
        for (int k = 0; k < 4 * R; k++) {
            const int c = lane0 + 64 * H + 128 * k;
            nat[k] = accbuf[(c & N) | EoLds::plane_word(c & (N - 1))];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4 * R; k++) accbuf[lane0 + 64 * H + 128 * k] = nat[k];
    }
    __syncthreads();
#ifdef RTFHE_WG_STAMPS
    if (a.dbg && blockIdx.x == 0 && lane0 == 0)
        for (int k = 0; k < 8; k++) a.dbg[wave * 8 + k] = tsum[k];
#endif

    if (a.mode == MODE_BLIND_ROTATE) {
        if (live) {
            uint32_t* o = a.out + (size_t)g * 2 * N;
            for (int c = lane0 + 64 * H; c < 2 * N; c += 128) o[c] = accbuf[c];
        }
        return;
    }

    // sample extract index 0 (trlwe.rs:110-121): a'_0 = a_0, a'_k = -a_{N-k}; b' = b_0
    {
        uint32_t av[2 * R];
#pragma unroll
        for (int mm = 0; mm < 2 * R; mm++) av[mm] = accbuf[N + lane0 + 64 * mm + 1024 * H];
        __syncthreads();
#pragma unroll
        for (int mm = 0; mm < 2 * R; mm++) {
            const int c = lane0 + 64 * mm + 1024 * H;
            accbuf[N + ((N - c) & (N - 1))] = (c == 0) ? av[mm] : (0u - av[mm]);
        }
    }
    __syncthreads();
    if (a.mode == MODE_EXTRACT) {      // the key switch of the whole batch follows as its own launch (k_key_switch_mm)
        if (live) {
            const int ge = a.ext_first + g;      // batch-wide gate number: the sample buffer is laid out for the key switch (ext_slot)
            for (int c = H * (N / 2) + lane0; c < (H + 1) * (N / 2); c += 64) *ext_slot(a.ext, ge, c, N) = accbuf[N + c];
            if (H == 0 && lane0 == 0) *ext_slot(a.ext, ge, N, N) = accbuf[0];
            for (int c = H * 64 + lane0; c <= n; c += 128) io.out[c] = 0u;
        }
        return;
    }
    // identity key switch (tlwe.rs:43-73): each wave sums the rows of half of the coefficients
    uint4 sum[KSQ];
    ks_accumulate<LOGN, KS_T, KS_BB, KSQ>(accbuf + N, H * (N / 2), (H + 1) * (N / 2), a.ksk, a.ksw, sum, lane0);
    uint4* part = reinterpret_cast<uint4*>(xb1) + lane0;   // [KSQ][64] uint4
    if (H == 1) {
#pragma unroll
        for (int q = 0; q < KSQ; q++) part[q * 64] = sum[q];
    }
    __syncthreads();
    if (H == 0 && live) {
        const uint32_t bprime = accbuf[0];
        uint32_t* out = io.out;
#pragma unroll
        for (int q = 0; q < KSQ; q++) {
            const uint4 o = part[q * 64];
            const int col = 4 * (lane0 + 64 * q);
            const uint32_t s[4] = {sum[q].x + o.x, sum[q].y + o.y, sum[q].z + o.z, sum[q].w + o.w};
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (col + e <= n) out[col + e] = ((col + e == n) ? bprime : 0u) - s[e];
        }
    }
}

}  // namespace rtfhe
