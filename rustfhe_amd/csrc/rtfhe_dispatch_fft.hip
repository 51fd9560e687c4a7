// rtfhe_dispatch_fft.hip -- the FP64 mirror backend's bootstrap kernels and the choice of kernel shape per batch.
#include "rtfhe_host.hpp"

#include "rtfhe_kernels_wg.hpp"
#include "rtfhe_kernels_pair.hpp"
#include "rtfhe_kernels_pair4.hpp"
#include "rtfhe_kernels_pair_rr.hpp"
#include "rtfhe_kernels_eo.hpp"
#include "rtfhe_kernels_eo4.hpp"

using namespace rtfhe;
using namespace rtfhe_host;

namespace {

template <int LOGN, int W>
int launch_bootstrap_w(rtfhe_ctx* ctx, BootstrapArgs a, hipStream_t s) {
    auto k = k_bootstrap<LOGN, 3, 6, 8, 2, KSQ, W>;
    const size_t lds = bootstrap_lds_bytes<LOGN>(W, a.npad, bootstrap_dual_xbuf(LOGN, W));
    if (int rc = allow_lds(ctx, k, lds)) return rc;
    const int grid = (a.count + W - 1) / W;
    hipLaunchKernelGGL(k, dim3(grid), dim3(64 * W), lds, s, a);
    HIPCHECK(ctx, hipGetLastError());
    ctx->launches++;
    return 0;
}

int launch_bootstrap_wg10(rtfhe_ctx* ctx, BootstrapArgs a, hipStream_t s) {
    auto k = k_bootstrap_wg<10, 3, 6, 8, 2, KSQ>;
    const size_t lds = WgLds<10, 3>::bytes(a.npad);
    if (int rc = allow_lds(ctx, k, lds)) return rc;
    hipLaunchKernelGGL(k, dim3(a.count), dim3(512), lds, s, a);
    HIPCHECK(ctx, hipGetLastError());
    ctx->launches++;
    return 0;
}

template <int GATES>
int launch_bootstrap_pair10_g(rtfhe_ctx* ctx, BootstrapArgs a, hipStream_t s) {
    auto k = k_bootstrap_pair<3, 6, 8, 2, KSQ, GATES>;
    const size_t lds = PairLds::bytes(GATES, a.npad);
    if (int rc = allow_lds(ctx, k, lds)) return rc;
    hipLaunchKernelGGL(k, dim3((a.count + GATES - 1) / GATES), dim3(128 * GATES), lds, s, a);
    HIPCHECK(ctx, hipGetLastError());
    ctx->launches++;
    return 0;
}
int launch_bootstrap_pair10(rtfhe_ctx* ctx, BootstrapArgs a, hipStream_t s) { return launch_bootstrap_pair10_g<4>(ctx, a, s); }
// 4 x CUs < count <= rr x CUs gates on the four wave pairs of every CU, time-sliced (rtfhe_kernels_pair_rr.hpp)
bool rr_applies(const rtfhe_ctx* ctx, size_t count) {
    const size_t cus = (size_t)ctx->num_cus, round = 4 * cus, rem = count % round;
    const int rr = ctx->rr > PairRrLds::GMAX ? PairRrLds::GMAX : ctx->rr;
    return rr > 4 && !ctx->force_waves && count > round && rem != 0 && round + rem <= (size_t)rr * cus;
}
int launch_bootstrap_pair_rr(rtfhe_ctx* ctx, BootstrapArgs a, hipStream_t s) {
    auto k = k_bootstrap_pair_rr<3, 6, 8, 2, KSQ>;
    const int wgs = ctx->num_cus, most = (a.count + wgs - 1) / wgs;
    if (a.count < 4 * wgs || most > PairRrLds::GMAX)
        return fail(ctx, RTFHE_ERR_STATE, "k_bootstrap_pair_rr: " + std::to_string(a.count) + " gates on " + std::to_string(wgs) + " CUs is not a shape it serves");
    const size_t lds = PairRrLds::bytes(most, a.npad);
    if (int rc = allow_lds(ctx, k, lds)) return rc;
    hipLaunchKernelGGL(k, dim3(wgs), dim3(512), lds, s, a);
    HIPCHECK(ctx, hipGetLastError());
    ctx->launches++;
    return 0;
}
// four waves per gate, (polynomial, parity): up to two gates per CU (rtfhe_kernels_pair4.hpp); no fused key switch
template <int GATES>
int launch_bootstrap_pair4_g(rtfhe_ctx* ctx, BootstrapArgs b, hipStream_t s) {
    auto k = k_bootstrap_pair4<3, 6, GATES>;
    const size_t lds = Pair4Lds::bytes(GATES, b.npad);
    if (int rc = allow_lds(ctx, k, lds)) return rc;
    Pair4Args a{b, ctx->d_p4bk};
    hipLaunchKernelGGL(k, dim3((b.count + GATES - 1) / GATES), dim3(256 * GATES), lds, s, a);
    HIPCHECK(ctx, hipGetLastError());
    ctx->launches++;
    return 0;
}

// N = 2048: two waves per transform, split by the parity of the point index (rtfhe_kernels_eo.hpp)
template <int GATES>
int launch_bootstrap_eo11_g(rtfhe_ctx* ctx, BootstrapArgs b, hipStream_t s) {
    auto k = k_bootstrap_eo<3, 6, 8, 2, KSQ, GATES>;
    const size_t lds = EoLds::bytes(GATES, b.npad);
    if (int rc = allow_lds(ctx, k, lds)) return rc;
    EoArgs a{b, ctx->d_etw, ctx->d_ebk};
    hipLaunchKernelGGL(k, dim3((b.count + GATES - 1) / GATES), dim3(128 * GATES), lds, s, a);
    HIPCHECK(ctx, hipGetLastError());
    ctx->launches++;
    return 0;
}
// four waves per gate, (polynomial, parity): batches of up to two gates per CU (rtfhe_kernels_eo4.hpp); no fused key switch
template <int GATES>
int launch_bootstrap_eo4_g(rtfhe_ctx* ctx, BootstrapArgs b, hipStream_t s) {
    auto k = k_bootstrap_eo4<3, 6, GATES>;
    const size_t lds = Eo4Lds::bytes(GATES, b.npad);
    if (int rc = allow_lds(ctx, k, lds)) return rc;
    EoArgs a{b, ctx->d_etw, ctx->d_ebk};
    hipLaunchKernelGGL(k, dim3((b.count + GATES - 1) / GATES), dim3(256 * GATES), lds, s, a);
    HIPCHECK(ctx, hipGetLastError());
    ctx->launches++;
    return 0;
}
template <int GATES>
int launch_bootstrap_n2048_g(rtfhe_ctx* ctx, BootstrapArgs b, hipStream_t s) {
    // up to two gates per CU: four waves per gate, so that no SIMD is left with a lone wave (single gate 9.78 -> 5.84 ms, 512 gates 9.91 -> 7.8 ms,
    // profiles/r04/n2048_four_waves_per_gate_ab.log); the fused key switch and a forced split stay on the two-wave kernels
    if constexpr (GATES <= 2) {
        if (ctx->eo4 && (b.mode == MODE_EXTRACT || b.mode == MODE_BLIND_ROTATE)) return launch_bootstrap_eo4_g<GATES>(ctx, b, s);
    }
    return launch_bootstrap_eo11_g<GATES>(ctx, b, s);
}

// Kernel shape by batch size (N = 1024), measured in profiles/r01_pair/shape_sweep.log:
//   whole rounds of 4 gates per CU : two waves per gate, 8-wave workgroups (k_bootstrap_pair) -- best throughput at every size
//   a remainder <= 1 gate per CU   : one gate per 8-wave workgroup (k_bootstrap_wg): ~2.4x lower latency
//   a remainder <= 2 / 3 gates per CU : the two-waves-per-gate kernel with 2 / 3 gates per workgroup, one workgroup per CU -- every
//                                    gate still has its two waves, which then share their SIMDs with fewer (or no) other waves
//   a larger remainder             : one more (partly filled) round of 4 gates per CU
// The segments are queued back to back on the caller's stream.  RTFHE_FORCE_WAVES=1|2|4|8 forces one shape for the
// whole batch (4, 8: one gate per wave in 4- / 8-wave workgroups).
template <int LOGN>
int launch_bootstrap_t(rtfhe_ctx* ctx, BootstrapArgs a, hipStream_t s) {
    if constexpr (LOGN == 10) {
        const int force = ctx->force_waves;
        if (force == 1) return launch_bootstrap_wg10(ctx, a, s);
        if (force == 2) return launch_bootstrap_pair10(ctx, a, s);
        if (force == 8) return launch_bootstrap_w<10, 8>(ctx, a, s);
        if (force == 4) return launch_bootstrap_w<10, 4>(ctx, a, s);
        if (split_ok(ctx, a, s)) return launch_split(ctx, a, s, launch_bootstrap_t<10>);     // comes back here in MODE_EXTRACT
        const size_t out_words = mode_out_words(a, 1 << LOGN);
        const size_t round = (size_t)4 * ctx->num_cus, count = (size_t)a.count;
        const size_t full = count / round * round, rem = count - full;
        // four waves per gate (k_bootstrap_pair4) where k_bootstrap_pair would leave SIMDs a lone wave: tails of more than wg_max gates and up to
        // two / three gates per CU (7 % on 257-512-gate and 1-4 % on 513-768-gate batches; at four gates per CU it loses 12 %, profiles/r04/pair4_ab.log)
        const bool p4 = ctx->p4bk_valid && (a.mode == MODE_EXTRACT || a.mode == MODE_BLIND_ROTATE);
        if (rr_applies(ctx, count)) {
            // the last whole round and the remainder as ONE launch of five or six gates per CU: (4 + rem / CUs) / 4 rounds instead of 2
            // (1,280 gates 9.3 -> 8.2 ms; profiles/r06/pair_rr_sweep.log)
            if (full > round)
                if (int rc = launch_bootstrap_pair10(ctx, batch_segment(ctx, a, 0, full - round, out_words), s)) return rc;
            return launch_bootstrap_pair_rr(ctx, batch_segment(ctx, a, full - round, round + rem, out_words), s);
        }
        if (full)
            if (int rc = launch_bootstrap_pair10(ctx, batch_segment(ctx, a, 0, full, out_words), s)) return rc;
        if (rem) {
            const BootstrapArgs tail = batch_segment(ctx, a, full, rem, out_words);
            if (rem <= (size_t)ctx->wg_max) return launch_bootstrap_wg10(ctx, tail, s);
            if (rem <= (size_t)2 * ctx->num_cus) return (p4 && ctx->pair4 >= 2) ? launch_bootstrap_pair4_g<2>(ctx, tail, s) : launch_bootstrap_pair10_g<2>(ctx, tail, s);
            if (rem <= (size_t)3 * ctx->num_cus) return (p4 && ctx->pair4 >= 3) ? launch_bootstrap_pair4_g<3>(ctx, tail, s) : launch_bootstrap_pair10_g<3>(ctx, tail, s);
            return launch_bootstrap_pair10(ctx, tail, s);
        }
        return 0;
    } else {
        // two waves per transform (two waves per SIMD, no AGPR traffic); RTFHE_FORCE_WAVES=4 selects one wave per gate
        // (inverse pass-1/untwist twiddles in global memory: 4 gates per CU fit)
        if (!(ctx->d_etw && ctx->ebk_valid) || ctx->force_waves == 4) return launch_bootstrap_w<11, 4>(ctx, a, s);
        // whole rounds of 4 gates per CU in one launch; a remainder with 1 / 2 / 3 gates per workgroup, one workgroup per CU
        // (a gate's two waves then share their SIMDs with fewer other waves: a single gate takes 0.67 x a full round)
        if (split_ok(ctx, a, s)) return launch_split(ctx, a, s, launch_bootstrap_t<11>);
        const size_t out_words = mode_out_words(a, 1 << LOGN);
        const size_t cus = (size_t)ctx->num_cus, round = 4 * cus, count = (size_t)a.count;
        const size_t full = count / round * round, rem = count - full;
        if (full)
            if (int rc = launch_bootstrap_n2048_g<4>(ctx, batch_segment(ctx, a, 0, full, out_words), s)) return rc;
        if (!rem) return 0;
        const BootstrapArgs tail = batch_segment(ctx, a, full, rem, out_words);
        if (rem <= cus) return launch_bootstrap_n2048_g<1>(ctx, tail, s);
        if (rem <= 2 * cus) return launch_bootstrap_n2048_g<2>(ctx, tail, s);
        if (rem <= 3 * cus) return launch_bootstrap_n2048_g<3>(ctx, tail, s);
        return launch_bootstrap_n2048_g<4>(ctx, tail, s);
    }
}

// which second key layout the dispatch above reads for a batch of `count` gates in `mode` (0 = none)
enum { LAYOUT_NONE = 0, LAYOUT_P4 = 1, LAYOUT_EO = 2 };
int layout_needed(const rtfhe_ctx* ctx, size_t count, int mode) {
    if (ctx->backend != RTFHE_BACKEND_FFT64_MIRROR) return LAYOUT_NONE;
    if (ctx->logn == 11) return ctx->force_waves == 4 ? LAYOUT_NONE : LAYOUT_EO;
    if (ctx->force_waves || ctx->pair4 < 2) return LAYOUT_NONE;
    const size_t cus = (size_t)ctx->num_cus, rem = count % (4 * cus);
    if (rr_applies(ctx, count)) return LAYOUT_NONE;      // the remainder rides with the last whole round (k_bootstrap_pair_rr reads d_bk)
    // (a MODE_GATE batch reaches the four-wave kernel through the split path only, as MODE_EXTRACT: without the matrix form of the key it stays on the fused kernels)
    if (mode == MODE_GATE && !(ctx->d_ksmat && ctx->ks_mm_min > 0 && count >= (size_t)ctx->ks_mm_min)) return LAYOUT_NONE;
    return (rem > (size_t)ctx->wg_max && rem <= (size_t)(ctx->pair4 < 3 ? 2 : 3) * cus) ? LAYOUT_P4 : LAYOUT_NONE;
}

}  // namespace

// ---- the device tables of the kernel families of this unit (HostTw: rtfhe_host.hpp, rtfhe_twiddles.hip) ----
// table of k_bootstrap_eo (N = 2048; layout: EoTw): wave H owns the points i = 2 j + H and runs nine of the ten stages on the 512-point
// sub-sequence j; the twiddle of pair (i, i + halfnn) is entry i mod halfnn = 2 (j mod halfnn / 2) + H of the reference's stage table
std::vector<cplx> HostTw::eo_table() const {
    typedef Geo<10> G;
    std::vector<cplx> t(EoTw::TOTAL, make_double2(0.0, 0.0));
    const double fold = 2.0 / (double)N;      // the inverse's input scaling (fft_processor_spqlios.cpp:158), exact, folded into the untwist
    for (int H = 0; H < 2; H++) {
        for (int m = 0; m < 8; m++)
            for (int lane = 0; lane < 64; lane++) {
                const int i = 2 * (lane + 64 * m) + H;
                t[EoTw::TWIST + (H * 8 + m) * 64 + lane] = make_double2(twist_c[i], twist_s[i]);
                t[EoTw::IUNTW + (H * 8 + m) * 64 + lane] = make_double2(untw_c[i] * fold, untw_s[i] * fold);
            }
        for (int mb = G::LR - 1; mb >= 0; mb--) {
            const int h = 1 << mb;
            for (int q = 0; q < h; q++) {
                const int e = G::R - 2 * h + q;
                for (int lane = 0; lane < 64; lane++) {           // pass 1: j-halfnn 64 h = i-halfnn 128 h
                    const int k = 2 * (lane + 64 * q) + H;
                    t[EoTw::P1 + (H * 7 + e) * 64 + lane] = make_double2(fwd_c[fwd_off(128 * h) + k], fwd_s[fwd_off(128 * h) + k]);
                    t[EoTw::IP1 + (H * 7 + e) * 64 + lane] = make_double2(inv_c[inv_off(128 * h) + k], inv_s[inv_off(128 * h) + k]);
                }
                for (int r = 0; r < G::NLOW; r++) {               // pass 2: j-halfnn 8 h = i-halfnn 16 h
                    const int k = 2 * ((q << G::LOW) | r) + H;
                    t[EoTw::P2 + (H * 7 + e) * G::NLOW + r] = make_double2(fwd_c[fwd_off(16 * h) + k], fwd_s[fwd_off(16 * h) + k]);
                    t[EoTw::IP2 + (H * 7 + e) * G::NLOW + r] = make_double2(inv_c[inv_off(16 * h) + k], inv_s[inv_off(16 * h) + k]);
                }
            }
        }
        for (int q = 0; q < 4; q++) {                             // pass 3: i-halfnn 8 (entries 0..3) and 4 (entries 4..5)
            t[EoTw::P3 + H * 8 + q] = make_double2(fwd_c[fwd_off(8) + 2 * q + H], fwd_s[fwd_off(8) + 2 * q + H]);
            t[EoTw::IP3 + H * 8 + q] = make_double2(inv_c[inv_off(8) + 2 * q + H], inv_s[inv_off(8) + 2 * q + H]);
        }
        for (int q = 0; q < 2; q++) {
            t[EoTw::P3 + H * 8 + 4 + q] = make_double2(fwd_c[fwd_off(4) + 2 * q + H], fwd_s[fwd_off(4) + 2 * q + H]);
            t[EoTw::IP3 + H * 8 + 4 + q] = make_double2(inv_c[inv_off(4) + 2 * q + H], inv_s[inv_off(4) + 2 * q + H]);
        }
    }
    return t;
}

// tables of the parity sub-networks of the latency kernel (N = 1024; layout: Q4Tw, rtfhe_sub256.hpp): wave H owns the points i = 2 j + H of a
// 512-point transform and runs its twiddled stages on the 256-point sub-sequence j; the twiddle of pair (i, i + halfnn) is entry
// i mod halfnn = 2 (j mod halfnn / 2) + H of the reference's stage table
std::vector<cplx> HostTw::q4_table() const {
    std::vector<cplx> t(Q4Tw::TOTAL, make_double2(0.0, 0.0));
    const double fold = 2.0 / (double)N;      // the inverse's input scaling (fft_processor_spqlios.cpp:158), exact, folded into the untwist
    for (int dir = 0; dir < 2; dir++)
        for (int H = 0; H < 2; H++) {
            cplx* d = t.data() + Q4Tw::off(dir, H);
            const double* sc = dir ? inv_c.data() : fwd_c.data();
            const double* ss = dir ? inv_s.data() : fwd_s.data();
            auto off = [&](int halfnn) { return dir ? inv_off(halfnn) : fwd_off(halfnn); };
            auto entry = [&](int halfnn, int k) { return make_double2(sc[off(halfnn) + k], ss[off(halfnn) + k]); };
            for (int m = 0; m < 4; m++)
                for (int lane = 0; lane < 64; lane++) {
                    const int i = 2 * (lane + 64 * m) + H;
                    d[Q4Tw::TW + m * 64 + lane] = dir ? make_double2(untw_c[i] * fold, untw_s[i] * fold) : make_double2(twist_c[i], twist_s[i]);
                }
            for (int mb = 1; mb >= 0; mb--) {
                const int h = 1 << mb;
                for (int q = 0; q < h; q++) {
                    const int e = 4 - 2 * h + q;
                    for (int lane = 0; lane < 64; lane++) d[Q4Tw::P1 + e * 64 + lane] = entry(128 * h, 2 * (lane + 64 * q) + H);     // pass 1: j-halfnn 64 h
                    for (int r = 0; r < 16; r++) d[Q4Tw::P2 + e * 16 + r] = entry(32 * h, 2 * ((q << 4) | r) + H);                  // pass 2: j-halfnn 16 h
                    for (int c = 0; c < 4; c++) d[Q4Tw::P3 + e * 4 + c] = entry(8 * h, 2 * ((q << 2) | c) + H);                     // pass 3: j-halfnn 4 h
                }
            }
            for (int q = 0; q < 2; q++) d[Q4Tw::P4 + q] = entry(4, 2 * q + H);                                                      // pass 4: i-halfnn 4
        }
    return t;
}

namespace rtfhe_host {

int launch_bootstrap_fft(rtfhe_ctx* ctx, BootstrapArgs a, hipStream_t s) {
    return ctx->logn == 10 ? launch_bootstrap_t<10>(ctx, a, s) : launch_bootstrap_t<11>(ctx, a, s);
}

// The key spectra once more in the layout a kernel family reads (derived from d_bk on this context's device), built by the first batch whose
// dispatch needs it -- a context that never runs such a batch never pays for the copy (N = 2048: 125 MB, N = 1024: 62 MB).  Called outside
// stream captures only (it allocates and synchronises); a dispatch inside a capture that finds the layout absent takes the kernels that read d_bk.
static int build_bk_layout(rtfhe_ctx* ctx, int which) {
    cplx** dst = which == LAYOUT_EO ? &ctx->d_ebk : &ctx->d_p4bk;
    bool* valid = which == LAYOUT_EO ? &ctx->ebk_valid : &ctx->p4bk_valid;
    const size_t polys = bk_word_count(ctx->p) / ctx->p.N;
    if (!*dst) HIPCHECK(ctx, hipMalloc((void**)dst, bk_cplx_count(ctx->p) * sizeof(cplx)));
    if (which == LAYOUT_EO) hipLaunchKernelGGL(k_bk_to_eo, dim3(2048), dim3(256), 0, ctx->stream, (const cplx*)ctx->d_bk, *dst, polys);
    else hipLaunchKernelGGL(k_bk_to_p4, dim3(2048), dim3(256), 0, ctx->stream, (const cplx*)ctx->d_bk, *dst, polys);
    HIPCHECK(ctx, hipGetLastError());
    HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
    *valid = true;
    return 0;
}

int ensure_bk_layouts(rtfhe_ctx* ctx, size_t count, int mode) {
    const int need = layout_needed(ctx, count, mode);
    if (need == LAYOUT_NONE || !ctx->d_bk) return 0;
    if (need == LAYOUT_EO ? ctx->ebk_valid : ctx->p4bk_valid) return 0;
    return build_bk_layout(ctx, need);
}

// A new key has just been put into d_bk (and d_bk_torus): every derived form of the key whose BUFFER ALREADY EXISTS is rebuilt in place, now,
// synchronously.  Such a buffer's address may be baked into a HIP graph -- one of this context's circuits, or a capture the caller took around a
// *_dev call -- and a replay must find the new key there, not the old key's spectra until some later eager batch happens to rebuild them
// (advisor r5).  Forms that were never built stay unbuilt (built on demand, ensure_bk_layouts / ntt_prepare / xfft_prepare).  The exact backends'
// forms derive from the torus form of the key: a key that came as spectra (rtfhe_load_bk_fft) has none, the old forms cannot be rebuilt, and
// the circuits recorded on those backends are marked stale (rtfhe_circuit_launch then fails with RTFHE_ERR_STATE instead of computing with
// the old key).
int rebuild_derived_keys(rtfhe_ctx* ctx) {
    if (int rc = use(ctx)) return rc;
    if (ctx->d_ebk) if (int rc = build_bk_layout(ctx, LAYOUT_EO)) return rc;
    if (ctx->d_p4bk) if (int rc = build_bk_layout(ctx, LAYOUT_P4)) return rc;
    if (ctx->d_bk_torus) {
        if (ctx->d_ntt_bk) if (int rc = ntt_prepare(ctx)) return rc;
        if (ctx->d_xbk) if (int rc = xfft_prepare(ctx)) return rc;
    } else if (ctx->d_ntt_bk || ctx->d_xbk) {
        for (rtfhe_circuit* c : ctx->circuits)
            if (c->backend != RTFHE_BACKEND_FFT64_MIRROR) c->stale = true;
    }
    return 0;
}

// grants every bootstrap kernel of this context's parameter set its dynamic LDS once, at context creation
int prime_fft_kernels(rtfhe_ctx* ctx) {
    const int npad = (ctx->p.n + 1 + 63) / 64 * 64;
    if (ctx->logn == 10) {
        if (int rc = allow_lds(ctx, k_bootstrap_pair<3, 6, 8, 2, KSQ, 4>, PairLds::bytes(4, npad))) return rc;
        if (int rc = allow_lds(ctx, k_bootstrap_pair<3, 6, 8, 2, KSQ, 3>, PairLds::bytes(3, npad))) return rc;
        if (int rc = allow_lds(ctx, k_bootstrap_pair<3, 6, 8, 2, KSQ, 2>, PairLds::bytes(2, npad))) return rc;
        if (int rc = allow_lds(ctx, k_bootstrap_wg<10, 3, 6, 8, 2, KSQ>, WgLds<10, 3>::bytes(npad))) return rc;
        // the time-sliced launch: as many gates per CU (five or six) as this mask length leaves room for in 160 KiB of LDS
        {
            int fit = 0;
            for (int g = 5; g <= PairRrLds::GMAX; g++)
                if (PairRrLds::bytes(g, npad) <= (size_t)160 * 1024) fit = g;
            if (ctx->rr > fit) ctx->rr = fit;
            if (ctx->rr >= 5)
                if (int rc = allow_lds(ctx, k_bootstrap_pair_rr<3, 6, 8, 2, KSQ>, PairRrLds::bytes(ctx->rr, npad))) return rc;
        }
        if (int rc = allow_lds(ctx, k_bootstrap_pair4<3, 6, 3>, Pair4Lds::bytes(3, npad))) return rc;
        if (int rc = allow_lds(ctx, k_bootstrap_pair4<3, 6, 2>, Pair4Lds::bytes(2, npad))) return rc;
        if (int rc = allow_lds(ctx, k_bootstrap<10, 3, 6, 8, 2, KSQ, 4>, bootstrap_lds_bytes<10>(4, npad, bootstrap_dual_xbuf(10, 4)))) return rc;
        if (int rc = allow_lds(ctx, k_bootstrap<10, 3, 6, 8, 2, KSQ, 8>, bootstrap_lds_bytes<10>(8, npad, bootstrap_dual_xbuf(10, 8)))) return rc;
    } else {
        if (int rc = allow_lds(ctx, k_bootstrap<11, 3, 6, 8, 2, KSQ, 4>, bootstrap_lds_bytes<11>(4, npad, bootstrap_dual_xbuf(11, 4)))) return rc;
        if (int rc = allow_lds(ctx, k_bootstrap_eo<3, 6, 8, 2, KSQ, 4>, EoLds::bytes(4, npad))) return rc;
        if (int rc = allow_lds(ctx, k_bootstrap_eo<3, 6, 8, 2, KSQ, 3>, EoLds::bytes(3, npad))) return rc;
        if (int rc = allow_lds(ctx, k_bootstrap_eo<3, 6, 8, 2, KSQ, 2>, EoLds::bytes(2, npad))) return rc;
        if (int rc = allow_lds(ctx, k_bootstrap_eo<3, 6, 8, 2, KSQ, 1>, EoLds::bytes(1, npad))) return rc;
        if (int rc = allow_lds(ctx, k_bootstrap_eo4<3, 6, 2>, Eo4Lds::bytes(2, npad))) return rc;
        if (int rc = allow_lds(ctx, k_bootstrap_eo4<3, 6, 1>, Eo4Lds::bytes(1, npad))) return rc;
    }
    return 0;
}

}  // namespace rtfhe_host
