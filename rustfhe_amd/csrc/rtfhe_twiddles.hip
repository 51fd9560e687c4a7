// rtfhe_twiddles.hip -- host twiddle tables and the device tables of every kernel family built from them.
// Values follow the reference's table builders so that a context created against the same libm as the reference holds the same bits:
//   accurate_cos / accurate_sin   utils/src/spqlios/spqlios-fft-impl.cpp:99-113
//   new_ifft_table                utils/src/spqlios/spqlios-fft-impl.cpp:400-437
//   new_fft_table                 utils/src/spqlios/spqlios-fft-impl.cpp:158-193
#include "rtfhe_host.hpp"

#include <cmath>

using namespace rtfhe;

namespace {

double quad_cos(int i, int n) {
    i = ((i % n) + n) % n;
    if (i >= 3 * n / 4) return std::cos(2. * M_PI * (n - i) / double(n));
    if (i >= 2 * n / 4) return -std::cos(2. * M_PI * (i - n / 2) / double(n));
    if (i >= 1 * n / 4) return -std::cos(2. * M_PI * (n / 2 - i) / double(n));
    return std::cos(2. * M_PI * (i) / double(n));
}
double quad_sin(int i, int n) {
    i = ((i % n) + n) % n;
    if (i >= 3 * n / 4) return -std::sin(2. * M_PI * (n - i) / double(n));
    if (i >= 2 * n / 4) return -std::sin(2. * M_PI * (i - n / 2) / double(n));
    if (i >= 1 * n / 4) return std::sin(2. * M_PI * (n / 2 - i) / double(n));
    return std::sin(2. * M_PI * (i) / double(n));
}

// reference memory layout: per stage, blocks | c0 c1 c2 c3 | s0 s1 s2 s3 |
size_t put(double* dst, const double* c, const double* s, int cnt) {
    size_t w = 0;
    for (int i = 0; i < cnt; i += 4) {
        for (int k = 0; k < 4; k++) dst[w++] = c[i + k];
        for (int k = 0; k < 4; k++) dst[w++] = s[i + k];
    }
    return w;
}
size_t get(const double* src, double* c, double* s, int cnt) {
    size_t r = 0;
    for (int i = 0; i < cnt; i += 4) {
        for (int k = 0; k < 4; k++) c[i + k] = src[r++];
        for (int k = 0; k < 4; k++) s[i + k] = src[r++];
    }
    return r;
}

}  // namespace

void HostTw::build(int N_) {
    N = N_;
    const int n = 2 * N, P = N / 2;
    twist_c.assign(P, 0); twist_s.assign(P, 0); untw_c.assign(P, 0); untw_s.assign(P, 0);
    fwd_c.assign(P, 0); fwd_s.assign(P, 0); inv_c.assign(P, 0); inv_s.assign(P, 0);
    for (int j = 0; j < P; j++) {
        twist_c[j] = quad_cos(j, n);  twist_s[j] = quad_sin(j, n);
        untw_c[j] = quad_cos(-j, n);  untw_s[j] = quad_sin(-j, n);
    }
    for (int halfnn = P / 2; halfnn >= 4; halfnn /= 2) {
        const int j = n / (2 * halfnn);
        for (int k = 0; k < halfnn; k++) {
            fwd_c[fwd_off(halfnn) + k] = quad_cos(j * k, n);
            fwd_s[fwd_off(halfnn) + k] = quad_sin(j * k, n);
            inv_c[inv_off(halfnn) + k] = quad_cos(-j * k, n);
            inv_s[inv_off(halfnn) + k] = quad_sin(-j * k, n);
        }
    }
}
void HostTw::export_ref(double* ifft_table, double* fft_table) const {
    const int P = N / 2;
    std::memset(ifft_table, 0, sizeof(double) * 2 * N);
    std::memset(fft_table, 0, sizeof(double) * 2 * N);
    size_t w = put(ifft_table, twist_c.data(), twist_s.data(), P);
    for (int h = P / 2; h >= 4; h /= 2) w += put(ifft_table + w, fwd_c.data() + fwd_off(h), fwd_s.data() + fwd_off(h), h);
    w = 0;
    for (int h = 4; h <= P / 2; h *= 2) w += put(fft_table + w, inv_c.data() + inv_off(h), inv_s.data() + inv_off(h), h);
    put(fft_table + w, untw_c.data(), untw_s.data(), P);
}
void HostTw::import_ref(const double* ifft_table, const double* fft_table) {
    const int P = N / 2;
    size_t r = get(ifft_table, twist_c.data(), twist_s.data(), P);
    for (int h = P / 2; h >= 4; h /= 2) r += get(ifft_table + r, fwd_c.data() + fwd_off(h), fwd_s.data() + fwd_off(h), h);
    r = 0;
    for (int h = 4; h <= P / 2; h *= 2) r += get(fft_table + r, inv_c.data() + inv_off(h), inv_s.data() + inv_off(h), h);
    get(fft_table + r, untw_c.data(), untw_s.data(), P);
}

// device table: per direction [twist R*64][pass1 (R-1)*64][pass2 (R-1)*NLOW][pass3 NLOW-4]
template <int LOGN>
static std::vector<cplx> device_table_t(const HostTw& T) {
    typedef Geo<LOGN> G;
    std::vector<cplx> t(G::TW_TOTAL);
    for (int dir = 0; dir < 2; dir++) {
        cplx* d = t.data() + dir * G::TW_DIR;
        const double* tc = dir ? T.untw_c.data() : T.twist_c.data();
        const double* ts = dir ? T.untw_s.data() : T.twist_s.data();
        const double* sc = dir ? T.inv_c.data() : T.fwd_c.data();
        const double* ss = dir ? T.inv_s.data() : T.fwd_s.data();
        auto off = [&](int halfnn) { return dir ? T.inv_off(halfnn) : T.fwd_off(halfnn); };
        // The reference scales the inverse transform's INPUT by 2/N (fft_processor_spqlios.cpp:158,166-180).  2/N is a
        // power of two and IEEE rounding is invariant under exact power-of-two scaling (no under/overflow anywhere
        // near these magnitudes), so folding the factor into the final untwist twiddles gives bit-identical
        // outputs and saves one multiply per point.
        const double fold = dir ? 2.0 / (double)T.N : 1.0;
        for (int m = 0; m < G::R; m++)
            for (int lane = 0; lane < 64; lane++) {
                const int pos = G::pos1(lane, m);
                d[G::TW_TWIST + m * 64 + lane] = make_double2(tc[pos] * fold, ts[pos] * fold);
            }
        for (int mb = G::LR - 1; mb >= 0; mb--) {
            const int h = 1 << mb;
            for (int q = 0; q < h; q++) {
                const int e = G::R - 2 * h + q;
                for (int lane = 0; lane < 64; lane++) {           // pass 1: global halfnn = 64 h
                    const int idx = lane + 64 * q;
                    d[G::TW_P1 + e * 64 + lane] = make_double2(sc[off(64 * h) + idx], ss[off(64 * h) + idx]);
                }
                for (int r = 0; r < G::NLOW; r++) {               // pass 2: global halfnn = NLOW h
                    const int idx = (q << G::LOW) | r;
                    d[G::TW_P2 + e * G::NLOW + r] = make_double2(sc[off(G::NLOW * h) + idx], ss[off(G::NLOW * h) + idx]);
                }
            }
        }
        for (int mb = G::LOW - 1; mb >= 2; mb--) {                // pass 3: global halfnn = h, wave-uniform
            const int h = 1 << mb;
            for (int q = 0; q < h; q++)
                d[G::TW_P3 + G::NLOW - 2 * h + q] = make_double2(sc[off(h) + q], ss[off(h) + q]);
        }
    }
    return t;
}
std::vector<cplx> HostTw::device_table(int logn) const { return logn == 10 ? device_table_t<10>(*this) : device_table_t<11>(*this); }

namespace rtfhe_host {

// The bootstrap kernels skip the multiplies of the butterfly whose twiddle is the first entry of the halfnn = 4 stage (BOOT_TRIV,
// rtfhe_device.hpp): that entry must be exactly (1, +-0) -- cos(0), sin(0), which every libm returns exactly and every table the reference's
// builders produce holds.  A table imported through rtfhe_set_twiddles is checked here and refused otherwise.
bool unit_twiddles_ok(const HostTw& tw) {
    auto is_zero = [](double v) { return v == 0.0; };
    // (k_bootstrap_eo, N = 2048, also skips the first butterfly of the halfnn = 8 stage of the even-point sub-network)
    for (int h : {4, 8})
        if (!(tw.fwd_c[tw.fwd_off(h)] == 1.0 && is_zero(tw.fwd_s[tw.fwd_off(h)]) && tw.inv_c[tw.inv_off(h)] == 1.0 && is_zero(tw.inv_s[tw.inv_off(h)]))) return false;
    return true;
}

int upload_twiddles(rtfhe_ctx* ctx) {
    if (BOOT_TRIV && !unit_twiddles_ok(ctx->tw))
        return fail(ctx, RTFHE_ERR_INVALID, "twiddle table: the first entry of the halfnn = 4 and 8 stages must be exactly (1, 0) in both directions "
                                            "(cos 0, sin 0: true of every table the reference builds)");
    std::vector<cplx> t = ctx->tw.device_table(ctx->logn);
    if (ctx->logn == 10) {      // the latency kernel's parity tables ride behind the table every kernel stages into LDS (k_bootstrap_wg reads them from there)
        const std::vector<cplx> q = ctx->tw.q4_table();
        t.insert(t.end(), q.begin(), q.end());
    }
    if (!ctx->d_tw) HIPCHECK(ctx, hipMalloc((void**)&ctx->d_tw, t.size() * sizeof(cplx)));
    HIPCHECK(ctx, hipMemcpy(ctx->d_tw, t.data(), t.size() * sizeof(cplx), hipMemcpyHostToDevice));
    if (ctx->logn == 11) {
        std::vector<cplx> e = ctx->tw.eo_table();
        if (!ctx->d_etw) HIPCHECK(ctx, hipMalloc((void**)&ctx->d_etw, e.size() * sizeof(cplx)));
        HIPCHECK(ctx, hipMemcpy(ctx->d_etw, e.data(), e.size() * sizeof(cplx), hipMemcpyHostToDevice));
    }
    return 0;
}

}  // namespace rtfhe_host
