// rtfhe_dispatch_xfft.hip -- the split-FFT exact backend (rtfhe_xfft.hpp; N = 1024: rtfhe_kernels_xfft.hpp, N = 2048: rtfhe_kernels_xfft2.hpp): host
// tables, the split key spectra, kernel shapes per batch.
#include "rtfhe_host.hpp"

#include <cmath>

#include "rtfhe_kernels_xfft.hpp"
#include "rtfhe_kernels_xfft2.hpp"
#include "rtfhe_kernels_xfft_rr.hpp"

using namespace rtfhe;
using namespace rtfhe_host;

namespace {

typedef xfft::XTw XTw;

cplx unit(long double angle) { return make_double2((double)cosl(angle), (double)sinl(angle)); }

// The device table (layout: XTw).  Angles are formed in long double and rounded once: an entry is within 1 ulp of the true value, which is all
// the error analysis assumes (scripts/xfft/model.py) -- no bit of this table has to match anything.
//   forward, stage s (1..9), block B:  w = exp(i theta_{s,B} / 2),  theta_{1,0} = pi/2,  theta_{s+1,2B} = theta_{s,B}/2,  theta_{s+1,2B+1} = theta_{s,B}/2 + pi
//   inverse, stage t (1..9), q < 2^(t-1):  w = exp(-2 pi i q / 2^t)
//   untwist, j < 512:  exp(-i (pi/2) j / 512) / 512
const long double PI = 3.141592653589793238462643383279502884L;

// root_theta: the ring of the 512-point forward transform is C[X]/(X^512 - exp(i root_theta)): pi/2 at N = 1024; at N = 2048 the two halves of the
// 1024-point transform are such rings with pi/4 and pi/4 + pi (rtfhe_kernels_xfft2.hpp)
std::vector<cplx> xfft_device_table(long double root_theta = PI / 2) {
    constexpr int n = 512, LOG = 9;
    std::vector<std::vector<long double>> theta(LOG + 2);
    theta[1] = {root_theta};
    for (int s = 1; s <= LOG; s++) {
        theta[s + 1].resize(theta[s].size() * 2);
        for (size_t B = 0; B < theta[s].size(); B++) { theta[s + 1][2 * B] = theta[s][B] / 2; theta[s + 1][2 * B + 1] = theta[s][B] / 2 + PI; }
    }
    auto fw = [&](int s, int B) { return unit(theta[s][B] / 2); };
    auto iw = [&](int t, int q) { return unit(-2 * PI * (long double)q / (long double)(1 << t)); };
    std::vector<cplx> tb(XTw::TOTAL, make_double2(0.0, 0.0));
    // entry e of a forward pass whose first stage is s0: e = 0: (s0, c), 1 + q: (s0 + 1, 2 c + q), 3 + q: (s0 + 2, 4 c + q); c = the block of stage s0
    auto fwd_entries = [&](int base, int s0, int classes) {
        for (int c = 0; c < classes; c++) {
            tb[base + 0 * classes + c] = fw(s0, c);
            for (int q = 0; q < 2; q++) tb[base + (1 + q) * classes + c] = fw(s0 + 1, 2 * c + q);
            for (int q = 0; q < 4; q++) tb[base + (3 + q) * classes + c] = fw(s0 + 2, 4 * c + q);
        }
    };
    fwd_entries(XTw::F1, 1, 1);
    fwd_entries(XTw::F2, 4, 8);
    fwd_entries(XTw::F3, 7, 64);
    // entry e of an inverse pass whose first stage is t0 (half-size h0 = 2^(t0-1)): lane class r < h0; e = 0: (t0, r), 1 + q: (t0 + 1, r + h0 q), 3 + q: (t0 + 2, r + h0 q)
    auto inv_entries = [&](int base, int t0, int classes) {
        for (int r = 0; r < classes; r++) {
            tb[base + 0 * classes + r] = iw(t0, r);
            for (int q = 0; q < 2; q++) tb[base + (1 + q) * classes + r] = iw(t0 + 1, r + classes * q);
            for (int q = 0; q < 4; q++) tb[base + (3 + q) * classes + r] = iw(t0 + 2, r + classes * q);
        }
    };
    inv_entries(XTw::I2, 4, 8);
    inv_entries(XTw::I3, 7, 64);
    for (int j = 0; j < n; j++) {
        const cplx u = unit(-(PI / 2) * (long double)j / (long double)n);
        tb[XTw::UT + j] = make_double2(u.x / n, u.y / n);      // [m][lane] with j = lane + 64 m
    }
    return tb;
}

// N = 2048 (XTw2): the forward tables of the two halves, the inverse tables (the standard DIT twiddles, as at N = 1024) and the last inverse
// stage's U_q = psi^-q / n, V_q = omega^-q U_q (psi = exp(i (pi/2) / n), omega = exp(2 pi i / n), n = 1024), cut by the half that uses them:
// half h reads q = lane + 64 (4 h + k), k < 4
std::vector<cplx> xfft2_device_table() {
    typedef xfft::XTw2 T2;
    std::vector<cplx> tb(T2::TOTAL, make_double2(0.0, 0.0));
    const std::vector<cplx> inv = xfft_device_table();
    for (int h = 0; h < 2; h++) {
        const std::vector<cplx> half = xfft_device_table(PI / 4 + (long double)h * PI);
        for (int e = 0; e < T2::FH; e++) tb[T2::F + h * T2::FH + e] = half[XTw::F1 + e];
    }
    for (int e = 0; e < 7 * 8; e++) tb[T2::I2 + e] = inv[XTw::I2 + e];
    for (int e = 0; e < 7 * 64; e++) tb[T2::I3 + e] = inv[XTw::I3 + e];
    constexpr int n = 1024;
    for (int h = 0; h < 2; h++)
        for (int k = 0; k < 4; k++)
            for (int lane = 0; lane < 64; lane++) {
                const int q = lane + 64 * (4 * h + k);
                const long double ua = -(PI / 2) * (long double)q / (long double)n, va = ua - 2 * PI * (long double)q / (long double)n;
                const cplx u = unit(ua), v = unit(va);
                tb[T2::UV + ((h * 2 + 0) * 4 + k) * 64 + lane] = make_double2(u.x / n, u.y / n);
                tb[T2::UV + ((h * 2 + 1) * 4 + k) * 64 + lane] = make_double2(v.x / n, v.y / n);
            }
    return tb;
}

template <int GATES>
int launch_xpair_g(rtfhe_ctx* ctx, BootstrapArgs b, hipStream_t s) {
    auto k = k_bootstrap_xpair<3, 6, 8, 2, KSQ, GATES>;
    const size_t lds = XPairLds::bytes(GATES, b.npad);
    if (int rc = allow_lds(ctx, k, lds)) return rc;
    XBootstrapArgs a{b, ctx->d_xtw, ctx->d_xbk};
    hipLaunchKernelGGL(k, dim3((b.count + GATES - 1) / GATES), dim3(128 * GATES), lds, s, a);
    HIPCHECK(ctx, hipGetLastError());
    ctx->launches++;
    return 0;
}

template <int GATES>
int launch_xquad_g(rtfhe_ctx* ctx, BootstrapArgs b, hipStream_t s) {
    auto k = k_bootstrap_xquad<3, 6, 8, 2, KSQ, GATES>;
    const size_t lds = XQuadLds::bytes(GATES, b.npad);
    if (int rc = allow_lds(ctx, k, lds)) return rc;
    XBootstrapArgs a{b, ctx->d_xtw, ctx->d_xbk};
    hipLaunchKernelGGL(k, dim3((b.count + GATES - 1) / GATES), dim3(256 * GATES), lds, s, a);
    HIPCHECK(ctx, hipGetLastError());
    ctx->launches++;
    return 0;
}

// N = 2048: whole rounds of 2 gates per CU (four waves per gate: two waves per SIMD); a remainder of at most one gate per CU runs one gate per workgroup
int launch_xquad(rtfhe_ctx* ctx, BootstrapArgs a, hipStream_t s) {
    if (split_ok(ctx, a, s)) return launch_split(ctx, a, s, launch_xquad);
    const size_t out_words = mode_out_words(a, 2048);
    const size_t cus = (size_t)ctx->num_cus, round = 2 * cus, count = (size_t)a.count;
    const size_t full = count / round * round, rem = count - full;
    if (full)
        if (int rc = launch_xquad_g<2>(ctx, batch_segment(ctx, a, 0, full, out_words), s)) return rc;
    if (!rem) return 0;
    const BootstrapArgs tail = batch_segment(ctx, a, full, rem, out_words);
    return rem <= cus ? launch_xquad_g<1>(ctx, tail, s) : launch_xquad_g<2>(ctx, tail, s);
}

// 4 x CUs < count <= 6 x CUs gates on the four wave pairs of every CU, time-sliced (rtfhe_kernels_xfft_rr.hpp; the mirror backend's
// k_bootstrap_pair_rr on this backend's arithmetic)
int launch_xpair_rr(rtfhe_ctx* ctx, BootstrapArgs b, hipStream_t s) {
    auto k = k_bootstrap_xpair_rr<3, 6, 8, 2, KSQ>;
    const int wgs = ctx->num_cus, most = (b.count + wgs - 1) / wgs;
    if (b.count < 4 * wgs || most > XPairRrLds::GMAX)
        return fail(ctx, RTFHE_ERR_STATE, "k_bootstrap_xpair_rr: " + std::to_string(b.count) + " gates on " + std::to_string(wgs) + " CUs is not a shape it serves");
    const size_t lds = XPairRrLds::bytes(most, b.npad);
    if (int rc = allow_lds(ctx, k, lds)) return rc;
    XBootstrapArgs a{b, ctx->d_xtw, ctx->d_xbk};
    hipLaunchKernelGGL(k, dim3(wgs), dim3(512), lds, s, a);
    HIPCHECK(ctx, hipGetLastError());
    ctx->launches++;
    return 0;
}

// whole rounds of 4 gates per CU in one launch; a remainder with 1 / 2 / 3 gates per workgroup, one workgroup per CU (as the other two-waves-per-gate
// kernels); behind at least one whole round, a remainder of up to 2 gates per CU rides with the last whole round in one time-sliced launch
int launch_xpair(rtfhe_ctx* ctx, BootstrapArgs a, hipStream_t s) {
    if (split_ok(ctx, a, s)) return launch_split(ctx, a, s, launch_xpair);
    const size_t out_words = mode_out_words(a, 1024);
    const size_t cus = (size_t)ctx->num_cus, round = 4 * cus, count = (size_t)a.count;
    const size_t full = count / round * round, rem = count - full;
    const int rr = ctx->xrr > XPairRrLds::GMAX ? XPairRrLds::GMAX : ctx->xrr;
    if (rr > 4 && full && rem && round + rem <= (size_t)rr * cus) {
        if (full > round)
            if (int rc = launch_xpair_g<4>(ctx, batch_segment(ctx, a, 0, full - round, out_words), s)) return rc;
        return launch_xpair_rr(ctx, batch_segment(ctx, a, full - round, round + rem, out_words), s);
    }
    if (full)
        if (int rc = launch_xpair_g<4>(ctx, batch_segment(ctx, a, 0, full, out_words), s)) return rc;
    if (!rem) return 0;
    const BootstrapArgs tail = batch_segment(ctx, a, full, rem, out_words);
    if (rem <= cus) return launch_xpair_g<1>(ctx, tail, s);
    if (rem <= 2 * cus) return launch_xpair_g<2>(ctx, tail, s);
    if (rem <= 3 * cus) return launch_xpair_g<3>(ctx, tail, s);
    return launch_xpair_g<4>(ctx, tail, s);
}

}  // namespace

namespace rtfhe_host {

int xfft_prepare(rtfhe_ctx* ctx) {
    if (ctx->xfft_ready) return 0;
    if (!ctx->d_bk_torus) return fail(ctx, RTFHE_ERR_STATE, "the split-FFT exact backend needs the bootstrapping key in torus form (rtfhe_load_bk_torus)");
    const bool big = ctx->logn == 11;
    if (!ctx->d_xtw) {
        const std::vector<cplx> t = big ? xfft2_device_table() : xfft_device_table();
        HIPCHECK(ctx, hipMalloc((void**)&ctx->d_xtw, t.size() * sizeof(cplx)));
        HIPCHECK(ctx, hipMemcpy(ctx->d_xtw, t.data(), t.size() * sizeof(cplx), hipMemcpyHostToDevice));
    }
    const size_t polys = bk_word_count(ctx->p) / ctx->p.N;
    if (!ctx->d_xbk) HIPCHECK(ctx, hipMalloc((void**)&ctx->d_xbk, 2 * bk_cplx_count(ctx->p) * sizeof(cplx)));
    constexpr int W = 4;
    XBkArgs a{ctx->d_xtw, ctx->d_bk_torus, ctx->d_xbk, (int32_t)polys, 2 * ctx->p.l};
    if (big) {      // a work item = (polynomial, half of its spectrum)
        int grid = (int)((2 * polys + W - 1) / W); if (grid > 2048) grid = 2048;
        const size_t lds = (size_t)xfft::XTw2::TOTAL * sizeof(cplx) + (size_t)W * 2 * Geo<10>::XSLOTS * sizeof(double);
        if (int rc = allow_lds(ctx, k_xbk_build2<W>, lds)) return rc;
        hipLaunchKernelGGL(k_xbk_build2<W>, dim3(grid), dim3(64 * W), lds, ctx->stream, a);
    } else {
        int grid = (int)((polys + W - 1) / W); if (grid > 2048) grid = 2048;
        const size_t lds = (size_t)XTw::TOTAL * sizeof(cplx) + (size_t)W * 2 * Geo<10>::XSLOTS * sizeof(double);
        if (int rc = allow_lds(ctx, k_xbk_build<W>, lds)) return rc;
        hipLaunchKernelGGL(k_xbk_build<W>, dim3(grid), dim3(64 * W), lds, ctx->stream, a);
    }
    HIPCHECK(ctx, hipGetLastError());
    HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->xfft_ready = true;
    return 0;
}

int launch_bootstrap_xfft(rtfhe_ctx* ctx, BootstrapArgs a, hipStream_t s) { return ctx->logn == 11 ? launch_xquad(ctx, a, s) : launch_xpair(ctx, a, s); }

// external product of `count` TRLWE samples with bk[idx[g]] on this backend (stage-level entry point)
int launch_extprod_xfft(rtfhe_ctx* ctx, const int32_t* d_idx, const uint32_t* d_in, uint32_t* d_out, int32_t count, hipStream_t s) {
    if (ctx->logn == 11) {      // one workgroup of two waves per sample
        XExtProdArgs a{ctx->d_xtw, ctx->d_xbk, d_idx, d_in, d_out, count};
        const size_t lds = (size_t)xfft::XTw2::TOTAL * sizeof(cplx) + (size_t)2 * 2 * Geo<10>::XSLOTS * sizeof(double) + (size_t)2 * 2 * 4 * 64 * sizeof(cplx);
        if (int rc = allow_lds(ctx, k_external_product_xfft2<3, 6>, lds)) return rc;
        hipLaunchKernelGGL((k_external_product_xfft2<3, 6>), dim3(count), dim3(128), lds, s, a);
        HIPCHECK(ctx, hipGetLastError());
        return 0;
    }
    constexpr int W = 2;
    XExtProdArgs a{ctx->d_xtw, ctx->d_xbk, d_idx, d_in, d_out, count};
    const size_t lds = (size_t)XTw::TOTAL * sizeof(cplx) + (size_t)W * 2 * Geo<10>::XSLOTS * sizeof(double);
    if (int rc = allow_lds(ctx, k_external_product_xfft<3, 6, W>, lds)) return rc;
    hipLaunchKernelGGL((k_external_product_xfft<3, 6, W>), dim3((count + W - 1) / W), dim3(64 * W), lds, s, a);
    HIPCHECK(ctx, hipGetLastError());
    return 0;
}

int prime_xfft_kernels(rtfhe_ctx* ctx) {
    const int npad = (ctx->p.n + 1 + 63) / 64 * 64;
    if (ctx->logn == 11) {
        if (int rc = allow_lds(ctx, k_bootstrap_xquad<3, 6, 8, 2, KSQ, 2>, XQuadLds::bytes(2, npad))) return rc;
        if (int rc = allow_lds(ctx, k_bootstrap_xquad<3, 6, 8, 2, KSQ, 1>, XQuadLds::bytes(1, npad))) return rc;
        return 0;
    }
    if (int rc = allow_lds(ctx, k_bootstrap_xpair<3, 6, 8, 2, KSQ, 4>, XPairLds::bytes(4, npad))) return rc;
    {   // the time-sliced launch: five or six gates per CU, as many as this mask length leaves room for in 160 KiB of LDS
        int fit = 0;
        for (int g = 5; g <= XPairRrLds::GMAX; g++)
            if (XPairRrLds::bytes(g, npad) <= (size_t)160 * 1024) fit = g;
        ctx->xrr = ctx->rr < fit ? ctx->rr : fit;
        if (ctx->xrr >= 5)
            if (int rc = allow_lds(ctx, k_bootstrap_xpair_rr<3, 6, 8, 2, KSQ>, XPairRrLds::bytes(ctx->xrr, npad))) return rc;
    }
    if (int rc = allow_lds(ctx, k_bootstrap_xpair<3, 6, 8, 2, KSQ, 3>, XPairLds::bytes(3, npad))) return rc;
    if (int rc = allow_lds(ctx, k_bootstrap_xpair<3, 6, 8, 2, KSQ, 2>, XPairLds::bytes(2, npad))) return rc;
    if (int rc = allow_lds(ctx, k_bootstrap_xpair<3, 6, 8, 2, KSQ, 1>, XPairLds::bytes(1, npad))) return rc;
    return 0;
}

}  // namespace rtfhe_host
