// rtfhe_dispatch_ntt.hip -- the exact-integer NTT backend (rtfhe_ntt.hpp): host tables, the NTT-domain key, kernel shapes per batch.
#include "rtfhe_host.hpp"

#include "rtfhe_kernels_ntt.hpp"
#include "rtfhe_kernels_ntt_halves.hpp"
#include "rtfhe_kernels_ntt_wg.hpp"

using namespace rtfhe;
using namespace rtfhe_host;

namespace {

// ---- exact-integer NTT backend: host tables (rtfhe_ntt.hpp; validated by scripts/ntt/model.py) ----
typedef unsigned __int128 u128;
uint64_t mulmod_p(uint64_t a, uint64_t b) { return (uint64_t)((u128)a * b % ntt::P_U64); }
uint64_t powmod_p(uint64_t a, uint64_t e) { uint64_t r = 1; while (e) { if (e & 1) r = mulmod_p(r, a); a = mulmod_p(a, a); e >>= 1; } return r; }
double centred_p(uint64_t x) { return x > ntt::P_U64 / 2 ? -(double)(ntt::P_U64 - x) : (double)x; }
int bitrev(int x, int bits) { int r = 0; for (int i = 0; i < bits; i++) r |= ((x >> i) & 1) << (bits - 1 - i); return r; }

// z[k], k = 1..1023: the block twiddles of a 1024-point wave transform, k = (blocks of the stage) + block;
// device order: pass 1 [15], pass 2 [15][16], pass 3 [12][64]
void ntt_fill_table(double* d, const std::vector<uint64_t>& z) {
    for (int e = 0; e < 15; e++) d[ntt::TW_P1 + e] = centred_p(z[e + 1]);
    for (int mb = 3; mb >= 0; mb--) {
        const int nb = 8 >> mb;
        for (int idx = 0; idx < nb; idx++)
            for (int B = 0; B < 16; B++)
                d[ntt::TW_P2 + (nb - 1 + idx) * 16 + B] = centred_p(z[(128 >> mb) + (B << (3 - mb)) + idx]);
    }
    for (int v = 0; v < 64; v++) {
        for (int e = 0; e < 4; e++) d[ntt::TW_P3 + e * 64 + v] = centred_p(z[256 + 4 * v + e]);
        for (int e = 0; e < 8; e++) d[ntt::TW_P3 + (4 + e) * 64 + v] = centred_p(z[512 + 8 * v + e]);
    }
}

// digit table: entry e = (e as a signed 6-bit value) * c mod P, centred
void ntt_fill_digits(double* d, uint64_t zeta1) {
    for (int e = 0; e < ntt::DIGITS; e++) {
        const int sdig = e < ntt::DIGITS / 2 ? e : e - ntt::DIGITS;
        const uint64_t mag = mulmod_p((uint64_t)(sdig < 0 ? -sdig : sdig), zeta1);
        d[e] = centred_p(sdig < 0 ? (ntt::P_U64 - mag) % ntt::P_U64 : mag);
    }
}

// N = 1024: zeta_k = psi^bitrev(k), psi a primitive 2048-th root of unity (22 generates F_P^*)
std::vector<double> ntt_device_table() {
    const uint64_t psi = powmod_p(22, (ntt::P_U64 - 1) / (2 * ntt::N));
    std::vector<uint64_t> zeta(ntt::N), zinv(ntt::N);
    for (int k = 1; k < ntt::N; k++) { zeta[k] = powmod_p(psi, (uint64_t)bitrev(k, 10)); zinv[k] = powmod_p(zeta[k], ntt::P_U64 - 2); }
    std::vector<double> t(ntt::TW_TOTAL, 0.0);
    ntt_fill_table(t.data(), zeta);
    ntt_fill_table(t.data() + ntt::TW_DIR_PAD, zinv);
    // the five digit tables of the first two stages: zeta_1; zeta_2, zeta_3 (the two blocks of stage 2); zeta_2 zeta_1, zeta_3 zeta_1
    const uint64_t coeff[ntt::DIG_TABLES] = {zeta[1], zeta[2], zeta[3], mulmod_p(zeta[2], zeta[1]), mulmod_p(zeta[3], zeta[1])};
    for (int k = 0; k < ntt::DIG_TABLES; k++) ntt_fill_digits(t.data() + ntt::TW_DIG + k * ntt::DIGITS, coeff[k]);
    return t;
}

// N = 2048 (rtfhe_kernels_ntt_halves.hpp; scripts/ntt/model2048.py): [half][1024] forward tables; the 1024-point transform of
// half H uses zeta_{k' + (1 + H) 2^floor(log2 k')} of the 2048-point table; the pad entry holds zeta_1 (the stage across the halves)
std::vector<double> ntt_halves_device_table() {
    constexpr int N2 = 2048;
    const uint64_t psi = powmod_p(22, (ntt::P_U64 - 1) / (2 * N2));
    std::vector<uint64_t> zeta(N2);
    for (int k = 1; k < N2; k++) zeta[k] = powmod_p(psi, (uint64_t)bitrev(k, 11));
    std::vector<double> t(NttHalvesTw::TOTAL, 0.0);
    for (int H = 0; H < 2; H++) {
        std::vector<uint64_t> sub(ntt::N);
        for (int kp = 1; kp < ntt::N; kp++) {
            int top = 0; while ((2 << top) <= kp) top++;
            sub[kp] = zeta[kp + ((1 + H) << top)];
        }
        double* d = t.data() + (size_t)H * NttHalvesTw::TABLE;
        ntt_fill_table(d, sub);
        d[NttHalvesTw::CROSS] = centred_p(zeta[1]);
    }
    ntt_fill_digits(t.data() + NttHalvesTw::DIG, zeta[1]);
    return t;
}

template <int W>
int launch_bootstrap_ntt_w(rtfhe_ctx* ctx, BootstrapArgs b, hipStream_t s) {
    auto k = k_bootstrap_ntt<3, 6, 8, 2, KSQ, W>;
    const size_t lds = ntt_lds_bytes(W, b.npad);
    if (int rc = allow_lds(ctx, k, lds)) return rc;
    NttBootstrapArgs a{b, ctx->d_ntt_tw, ctx->d_ntt_bk};
    hipLaunchKernelGGL(k, dim3((b.count + W - 1) / W), dim3(64 * W), lds, s, a);
    HIPCHECK(ctx, hipGetLastError());
    ctx->launches++;
    return 0;
}

template <int GATES>
int launch_bootstrap_ntt_pair_g(rtfhe_ctx* ctx, BootstrapArgs b, hipStream_t s) {
    auto k = k_bootstrap_ntt_pair<3, 6, 8, 2, KSQ, GATES>;
    const size_t lds = NttPairLds::bytes(GATES, b.npad);
    if (int rc = allow_lds(ctx, k, lds)) return rc;
    NttBootstrapArgs a{b, ctx->d_ntt_tw, ctx->d_ntt_bk};
    hipLaunchKernelGGL(k, dim3((b.count + GATES - 1) / GATES), dim3(128 * GATES), lds, s, a);
    HIPCHECK(ctx, hipGetLastError());
    ctx->launches++;
    return 0;
}

// NTT backend, one gate per 8-wave workgroup (the latency shape)
int launch_bootstrap_ntt_wg(rtfhe_ctx* ctx, BootstrapArgs b, hipStream_t s) {
    auto k = k_bootstrap_ntt_wg<3, 6, 8, 2, KSQ>;
    const size_t lds = NttWgLds::bytes(b.npad);
    if (int rc = allow_lds(ctx, k, lds)) return rc;
    NttBootstrapArgs a{b, ctx->d_ntt_tw, ctx->d_ntt_bk};
    hipLaunchKernelGGL(k, dim3(b.count), dim3(512), lds, s, a);
    HIPCHECK(ctx, hipGetLastError());
    ctx->launches++;
    return 0;
}

// NTT backend, two waves per gate.  Whole rounds of 4 gates per CU in one launch; a remainder runs with 1 / 2 / 3 gates per
// workgroup (one workgroup per CU): with fewer gates per CU a gate's two waves share their SIMDs with fewer other waves -- a
// circuit wave of 1-3 gates takes 0.67 x the time of a full round instead of all of it.
int launch_bootstrap_ntt_pair(rtfhe_ctx* ctx, BootstrapArgs a, hipStream_t s) {
    if (split_ok(ctx, a, s)) return launch_split(ctx, a, s, launch_bootstrap_ntt_pair);
    const size_t out_words = mode_out_words(a, ntt::N);
    const size_t cus = (size_t)ctx->num_cus, round = 4 * cus, count = (size_t)a.count;
    const size_t full = count / round * round, rem = count - full;
    if (full)
        if (int rc = launch_bootstrap_ntt_pair_g<4>(ctx, batch_segment(ctx, a, 0, full, out_words), s)) return rc;
    if (!rem) return 0;
    const BootstrapArgs tail = batch_segment(ctx, a, full, rem, out_words);
    if (rem <= cus) return ctx->force_waves == 2 ? launch_bootstrap_ntt_pair_g<1>(ctx, tail, s) : launch_bootstrap_ntt_wg(ctx, tail, s);
    if (rem <= 2 * cus) return launch_bootstrap_ntt_pair_g<2>(ctx, tail, s);
    if (rem <= 3 * cus) return launch_bootstrap_ntt_pair_g<3>(ctx, tail, s);
    return launch_bootstrap_ntt_pair_g<4>(ctx, tail, s);
}

template <int GATES>
int launch_bootstrap_ntt_halves_g(rtfhe_ctx* ctx, BootstrapArgs b, hipStream_t s) {
    auto k = k_bootstrap_ntt_halves<3, 6, 8, 2, KSQ, GATES>;
    const size_t lds = NttHalvesLds::bytes(GATES, b.npad);
    if (int rc = allow_lds(ctx, k, lds)) return rc;
    NttHalvesArgs a{b, ctx->d_ntt_tw, ctx->d_ntt_bk};
    hipLaunchKernelGGL(k, dim3((b.count + GATES - 1) / GATES), dim3(128 * GATES), lds, s, a);
    HIPCHECK(ctx, hipGetLastError());
    ctx->launches++;
    return 0;
}

// NTT backend at N = 2048: the same ladder (NTT_HALVES_ROUND gates per CU in whole rounds, fewer per workgroup for a remainder)
constexpr int NTT_HALVES_ROUND = 4;
int launch_bootstrap_ntt_halves(rtfhe_ctx* ctx, BootstrapArgs a, hipStream_t s) {
    if (split_ok(ctx, a, s)) return launch_split(ctx, a, s, launch_bootstrap_ntt_halves);
    const size_t out_words = mode_out_words(a, 2048);
    const size_t cus = (size_t)ctx->num_cus, round = NTT_HALVES_ROUND * cus, count = (size_t)a.count;
    const size_t full = count / round * round, rem = count - full;
    if (full)
        if (int rc = launch_bootstrap_ntt_halves_g<NTT_HALVES_ROUND>(ctx, batch_segment(ctx, a, 0, full, out_words), s)) return rc;
    if (!rem) return 0;
    const BootstrapArgs tail = batch_segment(ctx, a, full, rem, out_words);
    if (rem <= cus || NTT_HALVES_ROUND == 1) return launch_bootstrap_ntt_halves_g<1>(ctx, tail, s);
    if (rem <= 2 * cus || NTT_HALVES_ROUND == 2) return launch_bootstrap_ntt_halves_g<2>(ctx, tail, s);
    if (rem <= 3 * cus || NTT_HALVES_ROUND == 3) return launch_bootstrap_ntt_halves_g<3>(ctx, tail, s);
    return launch_bootstrap_ntt_halves_g<4>(ctx, tail, s);
}

}  // namespace

namespace rtfhe_host {

int ntt_prepare(rtfhe_ctx* ctx) {
    if (ctx->ntt_ready) return 0;
    if (!ctx->d_bk_torus) return fail(ctx, RTFHE_ERR_STATE, "the NTT backend needs the bootstrapping key in torus form (rtfhe_load_bk_torus)");
    const bool halves = ctx->logn == 11;
    if (!ctx->d_ntt_tw) {
        std::vector<double> t = halves ? ntt_halves_device_table() : ntt_device_table();
        HIPCHECK(ctx, hipMalloc((void**)&ctx->d_ntt_tw, t.size() * sizeof(double)));
        HIPCHECK(ctx, hipMemcpy(ctx->d_ntt_tw, t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    const size_t words = bk_word_count(ctx->p);
    if (!ctx->d_ntt_bk) HIPCHECK(ctx, hipMalloc((void**)&ctx->d_ntt_bk, words * sizeof(double)));
    constexpr int W = 4;
    const int32_t polys = (int32_t)(words / ctx->p.N);
    const double ninv = centred_p(powmod_p((uint64_t)ctx->p.N, ntt::P_U64 - 2));
    int grid = (polys + W - 1) / W; if (grid > 2048) grid = 2048;
    if (halves) {
        NttHalvesBkArgs a{ctx->d_ntt_tw, ctx->d_bk_torus, ctx->d_ntt_bk, polys, 2 * ctx->p.l, ninv};
        const size_t lds = (size_t)(NttHalvesTw::TOTAL + W * ntt::XSLOTS) * sizeof(double);
        if (int rc = allow_lds(ctx, k_ntt_bk_halves<W>, lds)) return rc;
        hipLaunchKernelGGL(k_ntt_bk_halves<W>, dim3(grid), dim3(64 * W), lds, ctx->stream, a);
    } else {
        NttBkArgs a{ctx->d_ntt_tw, ctx->d_bk_torus, ctx->d_ntt_bk, polys, 2 * ctx->p.l, ninv};
        const size_t lds = (size_t)(ntt::TW_DIR_PAD + W * ntt::XSLOTS) * sizeof(double);
        if (int rc = allow_lds(ctx, k_ntt_bk<W>, lds)) return rc;
        hipLaunchKernelGGL(k_ntt_bk<W>, dim3(grid), dim3(64 * W), lds, ctx->stream, a);
    }
    HIPCHECK(ctx, hipGetLastError());
    HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->ntt_ready = true;
    return 0;
}

int launch_bootstrap_ntt(rtfhe_ctx* ctx, BootstrapArgs a, hipStream_t s) {
    // two waves per gate: 11.5 ms per 1024 gates vs 13.3 ms one wave per gate in 4-wave workgroups (RTFHE_FORCE_WAVES=4);
    // 6-wave workgroups of the latter measured slower still (64 k vs 76 k gates/s): LDS-bound
    if (ctx->logn == 11) return launch_bootstrap_ntt_halves(ctx, a, s);
    if (ctx->force_waves == 4) return launch_bootstrap_ntt_w<4>(ctx, a, s);
    return launch_bootstrap_ntt_pair(ctx, a, s);
}

// external product of `count` TRLWE samples with bk[idx[g]] on the NTT backend (stage-level entry point)
int launch_extprod_ntt(rtfhe_ctx* ctx, const int32_t* d_idx, const uint32_t* d_in, uint32_t* d_out, int32_t count, hipStream_t s) {
    if (ctx->logn == 11) {
        NttHalvesExtProdArgs a{ctx->d_ntt_tw, ctx->d_ntt_bk, d_idx, d_in, d_out, count};
        const size_t lds = NttHalvesLds::TW + (size_t)2 * 2048 * 4 + 2 * NttHalvesLds::XB;
        if (int rc = allow_lds(ctx, k_external_product_ntt_halves<3, 6>, lds)) return rc;
        hipLaunchKernelGGL((k_external_product_ntt_halves<3, 6>), dim3(a.count), dim3(128), lds, s, a);
    } else {
        constexpr int W = 4;
        NttExtProdArgs a{ctx->d_ntt_tw, ctx->d_ntt_bk, d_idx, d_in, d_out, count};
        const size_t lds = ntt_lds_bytes(W, 0);
        if (int rc = allow_lds(ctx, k_external_product_ntt<3, 6, W>, lds)) return rc;
        hipLaunchKernelGGL((k_external_product_ntt<3, 6, W>), dim3((a.count + W - 1) / W), dim3(64 * W), lds, s, a);
    }
    HIPCHECK(ctx, hipGetLastError());
    return 0;
}

int prime_ntt_kernels(rtfhe_ctx* ctx) {
    const int npad = (ctx->p.n + 1 + 63) / 64 * 64;
    if (ctx->logn == 10) {
        if (int rc = allow_lds(ctx, k_bootstrap_ntt_pair<3, 6, 8, 2, KSQ, 4>, NttPairLds::bytes(4, npad))) return rc;
        if (int rc = allow_lds(ctx, k_bootstrap_ntt_pair<3, 6, 8, 2, KSQ, 3>, NttPairLds::bytes(3, npad))) return rc;
        if (int rc = allow_lds(ctx, k_bootstrap_ntt_pair<3, 6, 8, 2, KSQ, 2>, NttPairLds::bytes(2, npad))) return rc;
        if (int rc = allow_lds(ctx, k_bootstrap_ntt_pair<3, 6, 8, 2, KSQ, 1>, NttPairLds::bytes(1, npad))) return rc;
        if (int rc = allow_lds(ctx, k_bootstrap_ntt<3, 6, 8, 2, KSQ, 4>, ntt_lds_bytes(4, npad))) return rc;
        if (int rc = allow_lds(ctx, k_bootstrap_ntt_wg<3, 6, 8, 2, KSQ>, NttWgLds::bytes(npad))) return rc;
    } else {
        if (int rc = allow_lds(ctx, k_bootstrap_ntt_halves<3, 6, 8, 2, KSQ, 4>, NttHalvesLds::bytes(4, npad))) return rc;
        if (int rc = allow_lds(ctx, k_bootstrap_ntt_halves<3, 6, 8, 2, KSQ, 3>, NttHalvesLds::bytes(3, npad))) return rc;
        if (int rc = allow_lds(ctx, k_bootstrap_ntt_halves<3, 6, 8, 2, KSQ, 2>, NttHalvesLds::bytes(2, npad))) return rc;
        if (int rc = allow_lds(ctx, k_bootstrap_ntt_halves<3, 6, 8, 2, KSQ, 1>, NttHalvesLds::bytes(1, npad))) return rc;
    }
    return 0;
}

}  // namespace rtfhe_host
