// rtfhe_api.hip -- C ABI (include/rtfhe.h) over the gfx950 kernels.  No CPU fallback: every compute
// entry point runs HIP kernels or fails with an error code.
#include "../../include/rtfhe.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "rtfhe_kernels.hpp"
#include "rtfhe_kernels_wg.hpp"
#include "rtfhe_kernels_pair.hpp"
#include "rtfhe_kernels_pair4.hpp"
#include "rtfhe_kernels_halves.hpp"
#include "rtfhe_kernels_eo.hpp"
#include "rtfhe_kernels_eo4.hpp"
#include "rtfhe_kernels_ntt.hpp"
#include "rtfhe_kernels_ntt_halves.hpp"
#include "rtfhe_kernels_anyn.hpp"
#include "rtfhe_kernels_ntt_wg.hpp"
#include "rtfhe_kernels_ksmm.hpp"

using namespace rtfhe;

namespace {

thread_local std::string g_last_error;

// ------------------------------------------------------------------------------------------------
// twiddle tables (host).  Values follow the reference's table builders so that a context created in
// the same process / against the same libm as the reference holds the same bits:
//   accurate_cos / accurate_sin   utils/src/spqlios/spqlios-fft-impl.cpp:99-113
//   new_ifft_table                utils/src/spqlios/spqlios-fft-impl.cpp:400-437
//   new_fft_table                 utils/src/spqlios/spqlios-fft-impl.cpp:158-193
// ------------------------------------------------------------------------------------------------
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

struct HostTw {
    int N = 0;
    // per-stage natural order; forward stages concatenated halfnn = P/2 .. 4, inverse halfnn = 4 .. P/2
    std::vector<double> twist_c, twist_s, untw_c, untw_s, fwd_c, fwd_s, inv_c, inv_s;
    int fwd_off(int halfnn) const { return N / 2 - 2 * halfnn; }
    int inv_off(int halfnn) const { return halfnn - 4; }

    void build(int N_) {
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
    // reference memory layout: per stage, blocks | c0 c1 c2 c3 | s0 s1 s2 s3 |
    static size_t put(double* dst, const double* c, const double* s, int cnt) {
        size_t w = 0;
        for (int i = 0; i < cnt; i += 4) {
            for (int k = 0; k < 4; k++) dst[w++] = c[i + k];
            for (int k = 0; k < 4; k++) dst[w++] = s[i + k];
        }
        return w;
    }
    static size_t get(const double* src, double* c, double* s, int cnt) {
        size_t r = 0;
        for (int i = 0; i < cnt; i += 4) {
            for (int k = 0; k < 4; k++) c[i + k] = src[r++];
            for (int k = 0; k < 4; k++) s[i + k] = src[r++];
        }
        return r;
    }
    void export_ref(double* ifft_table, double* fft_table) const {
        const int P = N / 2;
        std::memset(ifft_table, 0, sizeof(double) * 2 * N);
        std::memset(fft_table, 0, sizeof(double) * 2 * N);
        size_t w = put(ifft_table, twist_c.data(), twist_s.data(), P);
        for (int h = P / 2; h >= 4; h /= 2) w += put(ifft_table + w, fwd_c.data() + fwd_off(h), fwd_s.data() + fwd_off(h), h);
        w = 0;
        for (int h = 4; h <= P / 2; h *= 2) w += put(fft_table + w, inv_c.data() + inv_off(h), inv_s.data() + inv_off(h), h);
        put(fft_table + w, untw_c.data(), untw_s.data(), P);
    }
    void import_ref(const double* ifft_table, const double* fft_table) {
        const int P = N / 2;
        size_t r = get(ifft_table, twist_c.data(), twist_s.data(), P);
        for (int h = P / 2; h >= 4; h /= 2) r += get(ifft_table + r, fwd_c.data() + fwd_off(h), fwd_s.data() + fwd_off(h), h);
        r = 0;
        for (int h = 4; h <= P / 2; h *= 2) r += get(fft_table + r, inv_c.data() + inv_off(h), inv_s.data() + inv_off(h), h);
        get(fft_table + r, untw_c.data(), untw_s.data(), P);
    }

    // table of k_bootstrap_halves (N = 2048; layout: HalvesTw): per direction the twist of the 16 inputs of a lane, the
    // twiddles of the stage with halfnn = 512, and the 512-point sub-transform's stage tables in the geometry of Geo<10>.
    // (The two directions cannot share one table: the reference's inverse entries are the conjugates of the forward ones except
    // at the quarter turn of every stage, where cos is -6.1e-17 forward and +6.1e-17 inverse.)
    std::vector<cplx> halves_table() const {
        typedef Geo<10> G;
        std::vector<cplx> t(HalvesTw::TOTAL);
        const double fold = 2.0 / (double)N;      // the inverse's input scaling (fft_processor_spqlios.cpp:158), exact, folded into the untwist
        for (int k = 0; k < 16; k++)
            for (int lane = 0; lane < 64; lane++) {
                const int p = (k < 8) ? lane + 64 * k : 512 + lane + 64 * (k - 8);
                t[HalvesTw::TWIST + k * 64 + lane] = make_double2(twist_c[p], twist_s[p]);
                t[HalvesTw::IUNTW + k * 64 + lane] = make_double2(untw_c[p] * fold, untw_s[p] * fold);
            }
        for (int m = 0; m < 8; m++)
            for (int lane = 0; lane < 64; lane++) {
                const int q = lane + 64 * m;
                t[HalvesTw::ST1 + m * 64 + lane] = make_double2(fwd_c[fwd_off(512) + q], fwd_s[fwd_off(512) + q]);
                t[HalvesTw::IST1 + m * 64 + lane] = make_double2(inv_c[inv_off(512) + q], inv_s[inv_off(512) + q]);
            }
        for (int mb = G::LR - 1; mb >= 0; mb--) {
            const int h = 1 << mb;
            for (int q = 0; q < h; q++) {
                const int e = G::R - 2 * h + q;
                for (int lane = 0; lane < 64; lane++) {           // pass 1: halfnn = 64 h
                    const int idx = lane + 64 * q;
                    t[HalvesTw::P1 + e * 64 + lane] = make_double2(fwd_c[fwd_off(64 * h) + idx], fwd_s[fwd_off(64 * h) + idx]);
                    t[HalvesTw::IP1 + e * 64 + lane] = make_double2(inv_c[inv_off(64 * h) + idx], inv_s[inv_off(64 * h) + idx]);
                }
                for (int r = 0; r < G::NLOW; r++) {               // pass 2: halfnn = 8 h
                    const int idx = (q << G::LOW) | r;
                    t[HalvesTw::P2 + e * G::NLOW + r] = make_double2(fwd_c[fwd_off(G::NLOW * h) + idx], fwd_s[fwd_off(G::NLOW * h) + idx]);
                    t[HalvesTw::IP2 + e * G::NLOW + r] = make_double2(inv_c[inv_off(G::NLOW * h) + idx], inv_s[inv_off(G::NLOW * h) + idx]);
                }
            }
        }
        for (int q = 0; q < 4; q++) {                             // pass 3: halfnn = 4
            t[HalvesTw::P3 + q] = make_double2(fwd_c[fwd_off(4) + q], fwd_s[fwd_off(4) + q]);
            t[HalvesTw::IP3 + q] = make_double2(inv_c[inv_off(4) + q], inv_s[inv_off(4) + q]);
        }
        return t;
    }

    // table of k_bootstrap_eo (N = 2048; layout: EoTw): wave H owns the points i = 2 j + H and runs nine of the ten stages on the 512-point
    // sub-sequence j; the twiddle of pair (i, i + halfnn) is entry i mod halfnn = 2 (j mod halfnn / 2) + H of the reference's stage table
    std::vector<cplx> eo_table() const {
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
    std::vector<cplx> q4_table() const {
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

    // device table: per direction [twist R*64][pass1 (R-1)*64][pass2 (R-1)*NLOW][pass3 NLOW-4]
    template <int LOGN>
    std::vector<cplx> device_table() const {
        typedef Geo<LOGN> G;
        std::vector<cplx> t(G::TW_TOTAL);
        for (int dir = 0; dir < 2; dir++) {
            cplx* d = t.data() + dir * G::TW_DIR;
            const double* tc = dir ? untw_c.data() : twist_c.data();
            const double* ts = dir ? untw_s.data() : twist_s.data();
            const double* sc = dir ? inv_c.data() : fwd_c.data();
            const double* ss = dir ? inv_s.data() : fwd_s.data();
            auto off = [&](int halfnn) { return dir ? inv_off(halfnn) : fwd_off(halfnn); };
            // The reference scales the inverse transform's INPUT by 2/N (fft_processor_spqlios.cpp:158,166-180).  2/N is a
            // power of two and IEEE rounding is invariant under exact power-of-two scaling (no under/overflow anywhere
            // near these magnitudes), so folding the factor into the final untwist twiddles gives bit-identical
            // outputs and saves one multiply per point.
            const double fold = dir ? 2.0 / (double)N : 1.0;
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
};

}  // namespace

struct rtfhe_circuit;
struct rtfhe_ctx {
    rtfhe_params p{};
    int device = 0;
    int logn = 10;
    HostTw tw;
    cplx* d_tw = nullptr;
    cplx* d_bk = nullptr;
    cplx* d_htw = nullptr;            // N = 2048: tables of k_bootstrap_halves
    cplx* d_hbk = nullptr;            // N = 2048: key spectra in the halves layout
    cplx* d_etw = nullptr;            // N = 2048: tables of k_bootstrap_eo
    cplx* d_ebk = nullptr;            // N = 2048: key spectra in the even / odd layout
    cplx* d_p4bk = nullptr;           // N = 1024: key spectra in the layout of k_bootstrap_pair4
    int pair4 = 3;                    // N = 1024, four waves per gate (k_bootstrap_pair4) for batches and tails of more than wg_max gates and up to
                                      // `pair4` gates per CU (RTFHE_PAIR4: 0 = never, 2, 3 = default, 4 = A/B: full rounds too, 9 = A/B: also instead of the latency kernel)
    unsigned long long tune = 0;      // tuning builds only (rtfhe_debug_set_tune): handed to the kernels as BootstrapArgs::tune
    int eo4 = 1;                      // N = 2048, up to two gates per CU: 1 = four waves per gate (k_bootstrap_eo4), 0 = two (RTFHE_N2048_EO4)
    int n2048_kernel = -1;            // -1 = by launch shape (below), 0 = parity split (k_bootstrap_eo), 1 = top-bit split (k_bootstrap_halves);
                                      // RTFHE_N2048_KERNEL=eo|halves
    int backend = RTFHE_BACKEND_FFT64_MIRROR;
    uint32_t* d_bk_torus = nullptr;   // kept when the key came in torus form: source for the NTT-domain key
    double* d_ntt_bk = nullptr;
    double* d_ntt_tw = nullptr;
    bool ntt_ready = false;
    uint32_t* d_ksk = nullptr;
    int ksw = 0;
    uint4* d_ksmat = nullptr;         // the key-switching key as signed byte limbs in i8-MFMA operand order (rtfhe_kernels_ksmm.hpp)
    // lvl1 samples between the two launches of the split path: ONE buffer per stream a batch was ever launched on (launches of one stream are
    // ordered, launches on different streams of one context may overlap and must not share it), plus a circuit's own during its capture
    struct Tlwe1 { uint32_t* d = nullptr; size_t cap = 0; };     // cap in gates
    std::unordered_map<hipStream_t, Tlwe1> tlwe1;
    Tlwe1* tlwe1_capture = nullptr;   // set by rtfhe_circuit_create around its capture: the circuit's buffer
    bool foreign_capture = false;     // set by launch_bootstrap for the duration of a call made inside a stream capture that is NOT
                                      // rtfhe_circuit_create's: such a batch stays on the fused kernel (see split_ok)
    int ks_mm_min = 1;                // batches of at least this many gates take the split path (0 = never: fused kernel); RTFHE_KS_MM_MIN
    bool has_bk = false, has_ksk = false;
    void* d_a = nullptr; void* d_b = nullptr; void* d_c = nullptr;   // device staging for host-pointer calls
    size_t cap_a = 0, cap_b = 0, cap_c = 0;
    void* h_pin[3] = {nullptr, nullptr, nullptr};                     // pinned host staging (pageable caller buffers go through it)
    size_t cap_pin[3] = {0, 0, 0};
    bool stage_pinned = false;                                        // RTFHE_STAGING=1: stage pageable caller buffers through h_pin (measured slower
                                                                      // than the runtime's own pageable path: +3.4 % vs +1.5 % at 1024 gates)
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int64_t launches = 0;
    // between rtfhe_timer_begin and _end every batch key switch of the split path is bracketed by a pair of events of its own, so
    // that the timer can report the blind-rotation kernel's and the key-switch kernel's device time separately
    bool timing = false;
    std::vector<hipEvent_t> ks_events;     // pool, pairs (before memset + k_key_switch_mm, after)
    size_t ks_events_used = 0;
    int32_t* d_fault = nullptr;            // set by a kernel that skipped a netlist gate (bad wire index / opcode)
    unsigned long long* d_dbg = nullptr;   // RTFHE_WG_STAMPS builds: 128 words of phase timings
    std::unordered_map<const void*, size_t> lds_allowed;   // kernel -> dynamic LDS bytes already granted on this device
    // multi-device context (rtfhe_ctx_create_multi): one full single-device context per further device; `this` is device 0 of
    // the set.  Keys are loaded once on this context and copied device-to-device; host-pointer batches are sharded.
    std::vector<rtfhe_ctx*> peers;
    std::vector<rtfhe_circuit*> circuits;   // live HIP-graph circuits of this context: orphaned (not freed) by rtfhe_ctx_destroy
    void* h_mux[2] = {nullptr, nullptr};   // device intermediates of rtfhe_mux_batch
    size_t cap_mux = 0;
    int num_cus = 256;
    int force_waves = 0;   // RTFHE_FORCE_WAVES=1|4|8 (tuning knob: 1 = workgroup-per-gate kernel)
    int wg_max = 512;      // RTFHE_WG_MAX_GATES: largest batch routed to the workgroup-per-gate kernel
    std::string err;
};

// a whole levelised netlist recorded into a HIP graph (rtfhe_circuit_create)
struct rtfhe_circuit {
    rtfhe_ctx* ctx = nullptr;  // null once the context has been destroyed (the handle then only remains to be freed)
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    int device = 0;            // kept here: the context may be destroyed before the circuit
    uint32_t* d_samples = nullptr;   // the circuit's own lvl1 sample buffer (split path): replays on any stream never share one with other work
    int32_t waves = 0;
    int64_t launches = 0;      // kernel launches one replay stands for
};

// releases the graph objects of a circuit (the handle itself stays valid for rtfhe_circuit_destroy)
static void circuit_release(rtfhe_circuit* c) {
    (void)hipSetDevice(c->device);
    if (c->exec) (void)hipGraphExecDestroy(c->exec);
    if (c->graph) (void)hipGraphDestroy(c->graph);
    if (c->d_samples) (void)hipFree(c->d_samples);
    c->exec = nullptr; c->graph = nullptr; c->d_samples = nullptr;
}

namespace {

int fail(rtfhe_ctx* ctx, int code, const std::string& msg) {
    g_last_error = msg;
    if (ctx) ctx->err = msg;
    return code;
}

#define HIPCHECK(ctx, expr)                                                                         \
    do {                                                                                            \
        hipError_t e__ = (expr);                                                                    \
        if (e__ != hipSuccess)                                                                      \
            return fail(ctx, RTFHE_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));    \
    } while (0)

int waves_for(int logn) { return logn == 10 ? 4 : 2; }
constexpr int KSQ = 3;   // uint4 loads per lane per key-switch row: rows up to 768 words

size_t bk_cplx_count(const rtfhe_params& p) { return (size_t)p.n * 2 * 2 * p.l * (p.N / 2); }
size_t bk_word_count(const rtfhe_params& p) { return (size_t)p.n * 2 * 2 * p.l * p.N; }
size_t ksk_rows(const rtfhe_params& p) { return (size_t)p.N * p.ks_t * ((1 << p.ks_basebit) - 1); }

int ensure(rtfhe_ctx* ctx, void** ptr, size_t* cap, size_t bytes) {
    if (*cap >= bytes && *ptr) return 0;
    if (*ptr) HIPCHECK(ctx, hipFree(*ptr));
    *ptr = nullptr; *cap = 0;
    HIPCHECK(ctx, hipMalloc(ptr, bytes ? bytes : 16));
    *cap = bytes;
    return 0;
}

// The dynamic-LDS limit of a kernel is a per-function, per-DEVICE attribute shared by every context of the process: it is only
// ever raised (a second context with a smaller mask would otherwise lower it under the first one's launches) and remembered
// process-wide, so that a launch costs no runtime call beyond the launch itself.
std::mutex g_lds_mutex;
std::map<std::pair<int, const void*>, size_t> g_lds_granted;     // (device, kernel) -> largest dynamic LDS granted so far

template <typename K>
int allow_lds(rtfhe_ctx* ctx, K kernel, size_t bytes) {
    const void* key = reinterpret_cast<const void*>(kernel);
    auto it = ctx->lds_allowed.find(key);
    if (it != ctx->lds_allowed.end() && it->second >= bytes) return 0;          // this context has already seen >= bytes granted
    std::lock_guard<std::mutex> lock(g_lds_mutex);
    size_t& granted = g_lds_granted[std::make_pair(ctx->device, key)];
    if (granted < bytes) {
        const hipError_t e = hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) {
            hipFuncAttributes fa{};
            const hipError_t e2 = hipFuncGetAttributes(&fa, key);
            return fail(ctx, RTFHE_ERR_HIP, std::string("hipFuncSetAttribute(MaxDynamicSharedMemorySize = ") + std::to_string(bytes) + "): " + hipGetErrorString(e) +
                        (e2 == hipSuccess ? " [kernel: static LDS " + std::to_string(fa.sharedSizeBytes) + ", regs " + std::to_string(fa.numRegs) +
                                            ", max threads " + std::to_string(fa.maxThreadsPerBlock) + "]" : std::string(" [hipFuncGetAttributes: ") + hipGetErrorString(e2) + "]"));
        }
        granted = bytes;
    }
    ctx->lds_allowed[key] = granted;
    return 0;
}

// The split path of a plain batch (whole rounds of the two-waves-per-gate kernels): blind rotation + sample extract of every gate
// (the bootstrap kernel in MODE_EXTRACT, launched by `blind_rotate`), then the key switch of the whole batch as one exact i8
// contraction on the matrix pipe (k_key_switch_mm) -- two launches back to back on the caller's stream, the lvl1 samples in between
// stay in HBM (4 MB per 1024 gates at N = 1024).
rtfhe_ctx::Tlwe1* tlwe1_of(rtfhe_ctx* ctx, hipStream_t s) {
    if (ctx->tlwe1_capture) return ctx->tlwe1_capture;
    auto it = ctx->tlwe1.find(s);
    return it == ctx->tlwe1.end() ? nullptr : &it->second;
}
bool split_ok(rtfhe_ctx* ctx, const BootstrapArgs& a, hipStream_t s) {
    if (!(a.mode == MODE_GATE && ctx->d_ksmat && ctx->ks_mm_min > 0 && (size_t)a.count >= (size_t)ctx->ks_mm_min)) return false;
    // A caller's own capture would bake THIS stream's scratch pointer into a graph the library does not own: a later, larger eager batch
    // on the stream frees and reallocates that buffer (ensure_tlwe1) and a replay then writes freed memory; a replay on another stream
    // would share the scratch with eager work on this one.  Only rtfhe_circuit_create's captures (which own their sample buffer) split.
    if (ctx->foreign_capture) return false;
    const rtfhe_ctx::Tlwe1* b = tlwe1_of(ctx, s);
    return b && (size_t)a.count <= b->cap;      // (the sample buffer is sized by ensure_tlwe1 before any launch or capture)
}
// lvl1 sample buffer of the split path for stream s: sized outside launches (hipMalloc is not allowed inside a stream capture)
int ensure_tlwe1(rtfhe_ctx* ctx, rtfhe_ctx::Tlwe1& b, size_t gates) {
    if (b.cap >= gates) return 0;
    HIPCHECK(ctx, hipDeviceSynchronize());            // earlier launches may still read the old buffer
    if (b.d) HIPCHECK(ctx, hipFree(b.d));
    b.d = nullptr; b.cap = 0;
    const size_t cap = gates < 1024 ? 1024 : gates;
    HIPCHECK(ctx, hipMalloc((void**)&b.d, cap * ((size_t)ctx->p.N + 1) * 4));
    b.cap = cap;
    return 0;
}
// blind_rotate(ctx, a', s) launches the bootstrap kernel(s) of the whole batch with a'.mode = MODE_EXTRACT: every gate's lvl1 sample goes to
// a'.ext (segments of a batch advance it by N + 1 words per gate) and its output row is zeroed for the key switch's atomics
template <typename F>
int launch_split(rtfhe_ctx* ctx, BootstrapArgs a, hipStream_t s, F blind_rotate) {
    uint32_t* samples = tlwe1_of(ctx, s)->d;
    a.mode = MODE_EXTRACT; a.ext = samples;
    if (int rc = blind_rotate(ctx, a, s)) return rc;
    const int colgroups = (ctx->p.n + 1 + 15) / 16, mgroups = (a.count + 16 * KSMM_MT - 1) / (16 * KSMM_MT);
    // K-slices: enough single-wave blocks to give every SIMD a few (the slices of one launch add into the zeroed output)
    int splitk = 1;
    while (splitk < 8 && (size_t)mgroups * colgroups * splitk < (size_t)8 * ctx->num_cus && (ctx->p.N / 4) % (8 * splitk) == 0) splitk *= 2;
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    // (no bracketing inside rtfhe_circuit_create's capture: a recorded event would become a graph node and rtfhe_timer_end would then ask
    // a never-recorded event for its time; and a timer that is never ended stops taking events at 4096 pairs)
    if (ctx->timing && !ctx->tlwe1_capture && ctx->ks_events_used + 2 <= 8192) {
        while (ctx->ks_events.size() < ctx->ks_events_used + 2) { hipEvent_t e; HIPCHECK(ctx, hipEventCreate(&e)); ctx->ks_events.push_back(e); }
        ev_a = ctx->ks_events[ctx->ks_events_used]; ev_b = ctx->ks_events[ctx->ks_events_used + 1];
        ctx->ks_events_used += 2;
        HIPCHECK(ctx, hipEventRecord(ev_a, s));
    }
    KsMmArgs k{samples, ctx->d_ksmat, a.out, a.count, ctx->p.n, ctx->p.N, colgroups, splitk, a.ops, a.idx0, a.idx1, a.idx_out, a.num_wires};
    hipLaunchKernelGGL((k_key_switch_mm<8, 2>), dim3(mgroups * colgroups * splitk), dim3(64), 0, s, k);
    HIPCHECK(ctx, hipGetLastError());
    if (ev_b) HIPCHECK(ctx, hipEventRecord(ev_b, s));
    ctx->launches++;
    return 0;
}

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

// N = 2048: two waves per transform (rtfhe_kernels_halves.hpp)
template <int GATES>
int launch_bootstrap_halves11_g(rtfhe_ctx* ctx, BootstrapArgs b, hipStream_t s) {
    auto k = k_bootstrap_halves<3, 6, 8, 2, KSQ, GATES>;
    const size_t lds = HalvesLds::bytes(GATES, b.npad);
    if (int rc = allow_lds(ctx, k, lds)) return rc;
    HalvesArgs a{b, ctx->d_htw, ctx->d_hbk};
    hipLaunchKernelGGL(k, dim3((b.count + GATES - 1) / GATES), dim3(128 * GATES), lds, s, a);
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
        if (ctx->n2048_kernel < 0 && ctx->eo4 && (b.mode == MODE_EXTRACT || b.mode == MODE_BLIND_ROTATE)) return launch_bootstrap_eo4_g<GATES>(ctx, b, s);
    }
    // Measured (profiles/r04/n2048_parity_split_ab.log, same process, identical outputs): the parity split is faster at every launch shape --
    // 4-10 % where a workgroup holds 1-2 gates (its trades are covered by arithmetic), 1.3 % at 4 gates per workgroup once its priority raise
    // sits inside the wait's assembly statement (no spills), level at 3.  RTFHE_N2048_KERNEL=halves keeps the top-bit split selectable.
    const bool eo = ctx->n2048_kernel < 0 ? true : ctx->n2048_kernel == 0;
    return eo ? launch_bootstrap_eo11_g<GATES>(ctx, b, s) : launch_bootstrap_halves11_g<GATES>(ctx, b, s);
}

// words per gate of the output buffer, by mode (MODE_EXTRACT: the final TLWE rows; the lvl1 samples go to `ext`)
size_t mode_out_words(const BootstrapArgs& a, int N) {
    return a.mode == MODE_BLIND_ROTATE ? (size_t)2 * N : (size_t)a.n + 1;
}


// `cnt` gates of a batch starting at gate `off` (plain batches advance the ciphertext pointers, netlist waves the index arrays)
BootstrapArgs batch_segment(const rtfhe_ctx* ctx, BootstrapArgs a, size_t off, size_t cnt, size_t out_words) {
    if (a.idx0) { a.ops += off; a.idx0 += off; a.idx1 += off; a.idx_out += off; }
    else { a.in0 += off * ((size_t)a.n + 1); a.in1 += off * ((size_t)a.n + 1); a.out += off * out_words; }
    if (a.ext) a.ext += off * ((size_t)ctx->p.N + 1);
    a.count = (int32_t)cnt;
    return a;
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
        if (full) {
            const BootstrapArgs seg = batch_segment(ctx, a, 0, full, out_words);
            if (ctx->pair4 >= 4 && ctx->pair4 != 9 && ctx->d_p4bk && (seg.mode == MODE_EXTRACT || seg.mode == MODE_BLIND_ROTATE)) { if (int rc = launch_bootstrap_pair4_g<4>(ctx, seg, s)) return rc; }      // A/B: RTFHE_PAIR4=4
            else if (int rc = launch_bootstrap_pair10(ctx, seg, s)) return rc;
        }
        if (rem) {
            const BootstrapArgs tail = batch_segment(ctx, a, full, rem, out_words);
            if (rem <= (size_t)ctx->wg_max) {
                if (ctx->pair4 == 9 && ctx->d_p4bk && (tail.mode == MODE_EXTRACT || tail.mode == MODE_BLIND_ROTATE)) return launch_bootstrap_pair4_g<1>(ctx, tail, s);     // A/B: RTFHE_PAIR4=9
                return launch_bootstrap_wg10(ctx, tail, s);
            }
            if (rem <= (size_t)2 * ctx->num_cus) {
                if (ctx->pair4 >= 2 && ctx->d_p4bk && (tail.mode == MODE_EXTRACT || tail.mode == MODE_BLIND_ROTATE)) return launch_bootstrap_pair4_g<2>(ctx, tail, s);
                return launch_bootstrap_pair10_g<2>(ctx, tail, s);
            }
            if (rem <= (size_t)3 * ctx->num_cus) {
                if (ctx->pair4 >= 3 && ctx->d_p4bk && (tail.mode == MODE_EXTRACT || tail.mode == MODE_BLIND_ROTATE)) return launch_bootstrap_pair4_g<3>(ctx, tail, s);
                return launch_bootstrap_pair10_g<3>(ctx, tail, s);
            }
            if (ctx->pair4 >= 4 && ctx->pair4 != 9 && ctx->d_p4bk && (tail.mode == MODE_EXTRACT || tail.mode == MODE_BLIND_ROTATE)) return launch_bootstrap_pair4_g<4>(ctx, tail, s);     // A/B: RTFHE_PAIR4=4
            return launch_bootstrap_pair10(ctx, tail, s);
        }
        return 0;
    } else {
        // two waves per transform (two waves per SIMD, no AGPR traffic); RTFHE_FORCE_WAVES=4 selects one wave per gate
        // (inverse pass-1/untwist twiddles in global memory: 4 gates per CU fit)
        if (!(ctx->d_htw && ctx->d_hbk) || ctx->force_waves == 4) return launch_bootstrap_w<11, 4>(ctx, a, s);
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

// digit table: entry e = (e as a signed 6-bit value) * zeta_1 mod P, centred
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
    ntt_fill_digits(t.data() + ntt::TW_DIG, zeta[1]);
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
#ifndef NTT_HALVES_ROUND
#define NTT_HALVES_ROUND 4
#endif
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

int launch_bootstrap(rtfhe_ctx* ctx, int op, int mode, int steps, const void* d_in0, const void* d_in1, void* d_out,
                     size_t count, hipStream_t s, const int32_t* d_ops = nullptr, const int32_t* d_idx0 = nullptr,
                     const int32_t* d_idx1 = nullptr, const int32_t* d_idx_out = nullptr, int32_t num_wires = 0) {
    if (!ctx->has_bk) return fail(ctx, RTFHE_ERR_STATE, "bootstrapping key not loaded");
    if (mode == MODE_GATE && !ctx->has_ksk) return fail(ctx, RTFHE_ERR_STATE, "key-switching key not loaded");
    if (count == 0) return 0;
    if (count > 0x7fffffff) return fail(ctx, RTFHE_ERR_INVALID, "count too large");
    BootstrapArgs a{};
    a.tw = ctx->d_tw; a.bk = ctx->d_bk; a.ksk = ctx->d_ksk;
    a.in0 = (const uint32_t*)d_in0; a.in1 = (const uint32_t*)(d_in1 ? d_in1 : d_in0); a.out = (uint32_t*)d_out;
    a.count = (int)count; a.op = op; a.n = ctx->p.n; a.steps = steps; a.mode = mode; a.ksw = ctx->ksw;
    a.npad = (ctx->p.n + 1 + 63) / 64 * 64;
    a.ops = d_ops; a.idx0 = d_idx0; a.idx1 = d_idx1; a.idx_out = d_idx_out;
    a.num_wires = num_wires; a.fault = ctx->d_fault;
    a.dbg = ctx->d_dbg;
    a.tune = ctx->tune;
    a.ext = nullptr;
    struct Unset { bool& f; ~Unset() { f = false; } } unset{ctx->foreign_capture};
    if (mode == MODE_GATE && ctx->ks_mm_min > 0 && ctx->d_ksmat && !ctx->tlwe1_capture) {
        // the split path's sample buffer of this stream is created / grows here, outside any stream capture; inside a capture that is not
        // rtfhe_circuit_create's own the batch stays on the fused kernel whatever buffer the stream already has (split_ok)
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusActive; }
        if (cs != hipStreamCaptureStatusNone) {
            ctx->foreign_capture = true;
        } else {
            const rtfhe_ctx::Tlwe1* have = tlwe1_of(ctx, s);
            if (!have || count > have->cap)
                if (int rc = ensure_tlwe1(ctx, ctx->tlwe1[s], count)) return rc;
        }
    }
    if (ctx->backend == RTFHE_BACKEND_NTT_EXACT) {
        if (int rc = ntt_prepare(ctx)) return rc;
        // two waves per gate: 11.5 ms per 1024 gates vs 13.3 ms one wave per gate in 4-wave workgroups (RTFHE_FORCE_WAVES=4);
        // 6-wave workgroups of the latter measured slower still (64 k vs 76 k gates/s): LDS-bound
        if (ctx->logn == 11) return launch_bootstrap_ntt_halves(ctx, a, s);
        if (ctx->force_waves == 4) return launch_bootstrap_ntt_w<4>(ctx, a, s);
        return launch_bootstrap_ntt_pair(ctx, a, s);
    }
    return ctx->logn == 10 ? launch_bootstrap_t<10>(ctx, a, s) : launch_bootstrap_t<11>(ctx, a, s);
}

template <int LOGN>
int launch_fft_t(rtfhe_ctx* ctx, bool forward, FftArgs a, hipStream_t s) {
    constexpr int W = 4;
    typedef Geo<LOGN> G;
    const size_t lds = (size_t)G::TW_DIR * sizeof(cplx) + (size_t)W * G::XSLOTS * sizeof(double);
    int grid = (a.count + W - 1) / W;
    if (grid > 2048) grid = 2048;
    if (forward) {
        auto k = k_fft_forward<LOGN, W>;
        if (int rc = allow_lds(ctx, k, lds)) return rc;
        hipLaunchKernelGGL(k, dim3(grid), dim3(64 * W), lds, s, a);
    } else {
        auto k = k_fft_inverse<LOGN, W>;
        if (int rc = allow_lds(ctx, k, lds)) return rc;
        hipLaunchKernelGGL(k, dim3(grid), dim3(64 * W), lds, s, a);
    }
    HIPCHECK(ctx, hipGetLastError());
    return 0;
}
int launch_fft(rtfhe_ctx* ctx, bool forward, FftArgs a, hipStream_t s) {
    if (a.count == 0) return 0;
    return ctx->logn == 10 ? launch_fft_t<10>(ctx, forward, a, s) : launch_fft_t<11>(ctx, forward, a, s);
}

template <int LOGN>
int launch_extprod_t(rtfhe_ctx* ctx, ExtProdArgs a, hipStream_t s) {
    constexpr int W = 4;
    auto k = k_external_product<LOGN, 3, 6, W>;
    const size_t lds = bootstrap_lds_bytes<LOGN>(W, 0);
    if (int rc = allow_lds(ctx, k, lds)) return rc;
    hipLaunchKernelGGL(k, dim3((a.count + W - 1) / W), dim3(64 * W), lds, s, a);
    HIPCHECK(ctx, hipGetLastError());
    return 0;
}

template <int LOGN>
int launch_keyswitch_t(rtfhe_ctx* ctx, KeySwitchArgs a, hipStream_t s) {
    constexpr int W = 4;
    auto k = k_key_switch<LOGN, 8, 2, KSQ, W>;
    const size_t lds = (size_t)W * (1 << LOGN) * 4;
    hipLaunchKernelGGL(k, dim3((a.count + W - 1) / W), dim3(64 * W), lds, s, a);
    HIPCHECK(ctx, hipGetLastError());
    return 0;
}

template <int LOGN>
int launch_permute_t(rtfhe_ctx* ctx, const double* src, double* dst, size_t count, int dir, int rows, hipStream_t s) {
    hipLaunchKernelGGL(k_bk_permute<LOGN>, dim3(2048), dim3(256), 0, s, src, dst, count, dir, rows);
    HIPCHECK(ctx, hipGetLastError());
    return 0;
}

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
    std::vector<cplx> t = ctx->logn == 10 ? ctx->tw.device_table<10>() : ctx->tw.device_table<11>();
    if (ctx->logn == 10) {      // the latency kernel's parity tables ride behind the table every kernel stages into LDS (k_bootstrap_wg reads them from there)
        const std::vector<cplx> q = ctx->tw.q4_table();
        t.insert(t.end(), q.begin(), q.end());
    }
    if (!ctx->d_tw) HIPCHECK(ctx, hipMalloc((void**)&ctx->d_tw, t.size() * sizeof(cplx)));
    HIPCHECK(ctx, hipMemcpy(ctx->d_tw, t.data(), t.size() * sizeof(cplx), hipMemcpyHostToDevice));
    if (ctx->logn == 11) {
        std::vector<cplx> h = ctx->tw.halves_table();
        if (!ctx->d_htw) HIPCHECK(ctx, hipMalloc((void**)&ctx->d_htw, h.size() * sizeof(cplx)));
        HIPCHECK(ctx, hipMemcpy(ctx->d_htw, h.data(), h.size() * sizeof(cplx), hipMemcpyHostToDevice));
        std::vector<cplx> e = ctx->tw.eo_table();
        if (!ctx->d_etw) HIPCHECK(ctx, hipMalloc((void**)&ctx->d_etw, e.size() * sizeof(cplx)));
        HIPCHECK(ctx, hipMemcpy(ctx->d_etw, e.data(), e.size() * sizeof(cplx), hipMemcpyHostToDevice));
    }
    return 0;
}

// N = 2048: the key spectra once more in the layout of k_bootstrap_halves (derived from d_bk on this context's device)
int build_halves_bk(rtfhe_ctx* ctx) {
    if (ctx->logn == 10 && ctx->d_bk) {      // N = 1024: the key spectra once more in the layout of k_bootstrap_pair4
        HIPCHECK(ctx, hipSetDevice(ctx->device));
        const size_t polys10 = bk_word_count(ctx->p) / ctx->p.N;
        if (!ctx->d_p4bk) HIPCHECK(ctx, hipMalloc((void**)&ctx->d_p4bk, bk_cplx_count(ctx->p) * sizeof(cplx)));
        hipLaunchKernelGGL(k_bk_to_p4, dim3(2048), dim3(256), 0, ctx->stream, (const cplx*)ctx->d_bk, ctx->d_p4bk, polys10);
        HIPCHECK(ctx, hipGetLastError());
        HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
        return 0;
    }
    if (ctx->logn != 11 || !ctx->d_bk) return 0;
    HIPCHECK(ctx, hipSetDevice(ctx->device));
    const size_t polys = bk_word_count(ctx->p) / ctx->p.N;
    if (!ctx->d_hbk) HIPCHECK(ctx, hipMalloc((void**)&ctx->d_hbk, bk_cplx_count(ctx->p) * sizeof(cplx)));
    hipLaunchKernelGGL(k_bk_to_halves, dim3(2048), dim3(256), 0, ctx->stream, (const cplx*)ctx->d_bk, ctx->d_hbk, polys, 1.0);
    HIPCHECK(ctx, hipGetLastError());
    if (!ctx->d_ebk) HIPCHECK(ctx, hipMalloc((void**)&ctx->d_ebk, bk_cplx_count(ctx->p) * sizeof(cplx)));
    hipLaunchKernelGGL(k_bk_to_eo, dim3(2048), dim3(256), 0, ctx->stream, (const cplx*)ctx->d_bk, ctx->d_ebk, polys);
    HIPCHECK(ctx, hipGetLastError());
    HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

// TRGSWRepF::from (trgsw.rs:68-76): ifft_torus = forward transform of the key words viewed as signed i32, from the
// device copy of the torus-form key into the device spectra
int transform_bk_from_torus(rtfhe_ctx* ctx) {
    const size_t words = bk_word_count(ctx->p);
    FftArgs a{ctx->d_tw, ctx->d_bk_torus, ctx->d_bk, (int32_t)(words / ctx->p.N), 1, 2 * ctx->p.l, 0};
    if (int rc = launch_fft(ctx, true, a, ctx->stream)) return rc;
    HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
    return build_halves_bk(ctx);
}

template <int LOGN>
int launch_poly_mul_t(rtfhe_ctx* ctx, PolyMulArgs a, hipStream_t s) {
    constexpr int W = 4;
    typedef Geo<LOGN> G;
    auto k = k_poly_mul<LOGN, W>;
    const size_t lds = (size_t)G::TW_TOTAL * sizeof(cplx) + (size_t)W * G::XSLOTS * sizeof(double);
    if (int rc = allow_lds(ctx, k, lds)) return rc;
    int grid = (a.count + W - 1) / W;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(k, dim3(grid), dim3(64 * W), lds, s, a);
    HIPCHECK(ctx, hipGetLastError());
    return 0;
}

int use(rtfhe_ctx* ctx) {
    if (!ctx) return fail(nullptr, RTFHE_ERR_INVALID, "null context");
    HIPCHECK(ctx, hipSetDevice(ctx->device));
    return 0;
}

// grants every bootstrap kernel of this context's parameter set its dynamic LDS once, at context creation
int prime_kernel_attributes(rtfhe_ctx* ctx) {
    const int npad = (ctx->p.n + 1 + 63) / 64 * 64;
    if (ctx->logn == 10) {
        if (int rc = allow_lds(ctx, k_bootstrap_pair<3, 6, 8, 2, KSQ, 4>, PairLds::bytes(4, npad))) return rc;
        if (int rc = allow_lds(ctx, k_bootstrap_pair<3, 6, 8, 2, KSQ, 3>, PairLds::bytes(3, npad))) return rc;
        if (int rc = allow_lds(ctx, k_bootstrap_pair<3, 6, 8, 2, KSQ, 2>, PairLds::bytes(2, npad))) return rc;
        if (int rc = allow_lds(ctx, k_bootstrap_wg<10, 3, 6, 8, 2, KSQ>, WgLds<10, 3>::bytes(npad))) return rc;
        if (int rc = allow_lds(ctx, k_bootstrap_pair4<3, 6, 3>, Pair4Lds::bytes(3, npad))) return rc;
        if (int rc = allow_lds(ctx, k_bootstrap_pair4<3, 6, 2>, Pair4Lds::bytes(2, npad))) return rc;
        if (int rc = allow_lds(ctx, k_bootstrap_pair4<3, 6, 1>, Pair4Lds::bytes(1, npad))) return rc;
        if (int rc = allow_lds(ctx, k_bootstrap<10, 3, 6, 8, 2, KSQ, 4>, bootstrap_lds_bytes<10>(4, npad, bootstrap_dual_xbuf(10, 4)))) return rc;
        if (int rc = allow_lds(ctx, k_bootstrap<10, 3, 6, 8, 2, KSQ, 8>, bootstrap_lds_bytes<10>(8, npad, bootstrap_dual_xbuf(10, 8)))) return rc;
        if (int rc = allow_lds(ctx, k_bootstrap_ntt_pair<3, 6, 8, 2, KSQ, 4>, NttPairLds::bytes(4, npad))) return rc;
        if (int rc = allow_lds(ctx, k_bootstrap_ntt_pair<3, 6, 8, 2, KSQ, 3>, NttPairLds::bytes(3, npad))) return rc;
        if (int rc = allow_lds(ctx, k_bootstrap_ntt_pair<3, 6, 8, 2, KSQ, 2>, NttPairLds::bytes(2, npad))) return rc;
        if (int rc = allow_lds(ctx, k_bootstrap_ntt_pair<3, 6, 8, 2, KSQ, 1>, NttPairLds::bytes(1, npad))) return rc;
        if (int rc = allow_lds(ctx, k_bootstrap_ntt<3, 6, 8, 2, KSQ, 4>, ntt_lds_bytes(4, npad))) return rc;
        if (int rc = allow_lds(ctx, k_bootstrap_ntt_wg<3, 6, 8, 2, KSQ>, NttWgLds::bytes(npad))) return rc;
    } else {
        if (int rc = allow_lds(ctx, k_bootstrap<11, 3, 6, 8, 2, KSQ, 4>, bootstrap_lds_bytes<11>(4, npad, bootstrap_dual_xbuf(11, 4)))) return rc;
        if (int rc = allow_lds(ctx, k_bootstrap_halves<3, 6, 8, 2, KSQ, 4>, HalvesLds::bytes(4, npad))) return rc;
        if (int rc = allow_lds(ctx, k_bootstrap_halves<3, 6, 8, 2, KSQ, 3>, HalvesLds::bytes(3, npad))) return rc;
        if (int rc = allow_lds(ctx, k_bootstrap_halves<3, 6, 8, 2, KSQ, 2>, HalvesLds::bytes(2, npad))) return rc;
        if (int rc = allow_lds(ctx, k_bootstrap_halves<3, 6, 8, 2, KSQ, 1>, HalvesLds::bytes(1, npad))) return rc;
        if (int rc = allow_lds(ctx, k_bootstrap_eo<3, 6, 8, 2, KSQ, 4>, EoLds::bytes(4, npad))) return rc;
        if (int rc = allow_lds(ctx, k_bootstrap_eo<3, 6, 8, 2, KSQ, 3>, EoLds::bytes(3, npad))) return rc;
        if (int rc = allow_lds(ctx, k_bootstrap_eo<3, 6, 8, 2, KSQ, 2>, EoLds::bytes(2, npad))) return rc;
        if (int rc = allow_lds(ctx, k_bootstrap_eo<3, 6, 8, 2, KSQ, 1>, EoLds::bytes(1, npad))) return rc;
        if (int rc = allow_lds(ctx, k_bootstrap_eo4<3, 6, 2>, Eo4Lds::bytes(2, npad))) return rc;
        if (int rc = allow_lds(ctx, k_bootstrap_eo4<3, 6, 1>, Eo4Lds::bytes(1, npad))) return rc;
        if (int rc = allow_lds(ctx, k_bootstrap_ntt_halves<3, 6, 8, 2, KSQ, 4>, NttHalvesLds::bytes(4, npad))) return rc;
        if (int rc = allow_lds(ctx, k_bootstrap_ntt_halves<3, 6, 8, 2, KSQ, 3>, NttHalvesLds::bytes(3, npad))) return rc;
        if (int rc = allow_lds(ctx, k_bootstrap_ntt_halves<3, 6, 8, 2, KSQ, 2>, NttHalvesLds::bytes(2, npad))) return rc;
        if (int rc = allow_lds(ctx, k_bootstrap_ntt_halves<3, 6, 8, 2, KSQ, 1>, NttHalvesLds::bytes(1, npad))) return rc;
    }
    return 0;
}

// A *_dev entry point must never launch on a pointer the GPU cannot dereference (a host pointer passed by mistake would fault
// the device): memory of the context's own device, managed and pinned-host allocations pass, memory of another GPU only with peer
// access, anything else is refused before the launch.
bool gpu_accessible(const rtfhe_ctx* ctx, const void* p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (a.type == hipMemoryTypeManaged || a.type == hipMemoryTypeHost) return true;
    if (a.type != hipMemoryTypeDevice) return false;
    if (a.device == ctx->device) return true;
    // memory of ANOTHER GPU: only when this device has peer access to it (a kernel on ctx->device would otherwise fault on it)
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, ctx->device, a.device) != hipSuccess || !can) { (void)hipGetLastError(); return false; }
    const hipError_t e = hipDeviceEnablePeerAccess(a.device, 0);          // current device = ctx->device (use())
    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); return false; }
    (void)hipGetLastError();
    return true;
}

// true when `p` is host memory the GPU can DMA from directly (hipHostMalloc / hipHostRegister, e.g. rtfhe_host_alloc)
bool is_pinned_host(const void* p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}

int ensure_pinned(rtfhe_ctx* ctx, int slot, size_t bytes) {
    if (ctx->cap_pin[slot] >= bytes && ctx->h_pin[slot]) return 0;
    if (ctx->h_pin[slot]) HIPCHECK(ctx, hipHostFree(ctx->h_pin[slot]));
    ctx->h_pin[slot] = nullptr; ctx->cap_pin[slot] = 0;
    HIPCHECK(ctx, hipHostMalloc(&ctx->h_pin[slot], bytes ? bytes : 16, hipHostMallocDefault));
    ctx->cap_pin[slot] = bytes;
    return 0;
}

// host -> device on ctx->stream.  Caller-pinned memory (rtfhe_host_alloc) is DMA'd as it is; pageable memory is handed to the
// runtime's pageable path, or -- RTFHE_STAGING=1 -- goes through the context's own pinned staging buffer `slot`.
int copy_in(rtfhe_ctx* ctx, void* dst, const void* src, size_t bytes, int slot) {
    if (ctx->stage_pinned && !is_pinned_host(src)) {
        if (int rc = ensure_pinned(ctx, slot, bytes)) return rc;
        std::memcpy(ctx->h_pin[slot], src, bytes);
        src = ctx->h_pin[slot];
    }
    HIPCHECK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return 0;
}

// device -> host on ctx->stream, synchronous on return
int copy_out(rtfhe_ctx* ctx, void* dst, const void* src, size_t bytes, int slot) {
    if (ctx->stage_pinned && !is_pinned_host(dst)) {
        if (int rc = ensure_pinned(ctx, slot, bytes)) return rc;
        HIPCHECK(ctx, hipMemcpyAsync(ctx->h_pin[slot], src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
        std::memcpy(dst, ctx->h_pin[slot], bytes);
        return 0;
    }
    HIPCHECK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

// contiguous range of device d of n_dev (sizes differ by at most one)
inline size_t shard_begin(size_t count, int d, int n_dev) { return count * (size_t)d / (size_t)n_dev; }

// runs fn(context of device d, d) for every device of a multi-device context, one host thread per device; first error wins
template <typename F>
int for_each_device(rtfhe_ctx* ctx, F fn) {
    const int n_dev = 1 + (int)ctx->peers.size();
    std::vector<int> rcs(n_dev, 0);
    std::vector<std::thread> th;
    for (int d = 1; d < n_dev; d++) th.emplace_back([&, d]() { rcs[d] = fn(ctx->peers[d - 1], d); });
    rcs[0] = fn(ctx, 0);
    for (auto& t : th) t.join();
    for (int d = 0; d < n_dev; d++)
        if (rcs[d]) return d == 0 ? rcs[0] : fail(ctx, rcs[d], "device " + std::to_string(ctx->peers[d - 1]->device) + ": " + ctx->peers[d - 1]->err);
    return 0;
}

// device 0's copy of a key -> every peer (device-to-device; xGMI between the GPUs of one node)
int replicate(rtfhe_ctx* ctx, rtfhe_ctx* peer, const void* src, void** dst_of_peer, size_t bytes) {
    if (!*dst_of_peer) {
        HIPCHECK(ctx, hipSetDevice(peer->device));
        HIPCHECK(ctx, hipMalloc(dst_of_peer, bytes));
    }
    HIPCHECK(ctx, hipMemcpyPeer(*dst_of_peer, peer->device, src, ctx->device, bytes));
    HIPCHECK(ctx, hipSetDevice(ctx->device));
    return 0;
}

}  // namespace

extern "C" {

void rtfhe_default_params(rtfhe_params* p) {
    p->n = 635; p->N = 1024; p->nbit = 10; p->l = 3; p->bgbit = 6; p->ks_t = 8; p->ks_basebit = 2;
}

const char* rtfhe_version(void) { return "rtfhe-hip 0.1 (gfx950, fft64-mirror)"; }

const char* rtfhe_last_error(const rtfhe_ctx* ctx) { return ctx ? ctx->err.c_str() : g_last_error.c_str(); }

int rtfhe_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static int create_single(const rtfhe_params* p, int device_id, rtfhe_ctx** out) {
    if (!p || !out) return fail(nullptr, RTFHE_ERR_INVALID, "null argument");
    *out = nullptr;
    if (p->N != 1024 && p->N != 2048) return fail(nullptr, RTFHE_ERR_INVALID, "supported TRLWE degrees: N = 1024, 2048");
    if (p->nbit != ilog2(p->N)) return fail(nullptr, RTFHE_ERR_INVALID, "nbit must be log2(N)");
    if (p->l != 3 || p->bgbit != 6) return fail(nullptr, RTFHE_ERR_INVALID, "supported gadget: l = 3, bgbit = 6");
    if (p->ks_t != 8 || p->ks_basebit != 2) return fail(nullptr, RTFHE_ERR_INVALID, "supported key switch: t = 8, basebit = 2");
    if (p->n < 1 || p->n + 1 > 256 * KSQ) return fail(nullptr, RTFHE_ERR_INVALID, "supported TLWE dimension: 1 <= n <= 767");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, RTFHE_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device_id < 0 || device_id >= ndev) return fail(nullptr, RTFHE_ERR_INVALID, "device_id out of range");
    rtfhe_ctx* ctx = new (std::nothrow) rtfhe_ctx();
    if (!ctx) return fail(nullptr, RTFHE_ERR_NOMEM, "out of host memory");
    ctx->p = *p; ctx->device = device_id; ctx->logn = p->nbit;
    ctx->ksw = (p->n + 1 + 3) / 4 * 4;
    ctx->tw.build(p->N);
    if (const char* e = std::getenv("RTFHE_TEST_PERTURB_TWIDDLE")) {
        // TEST ONLY: stands in for a host whose libm rounds one cos differently (2 of 2040 entries were an ulp off the correctly
        // rounded value in the survey's container).  Entry `e` of the forward stage table moves by one ulp.
        const size_t k = (size_t)std::atoi(e) % ctx->tw.fwd_c.size();
        ctx->tw.fwd_c[k] = std::nextafter(ctx->tw.fwd_c[k], 2.0);
    }
    int rc = use(ctx);
    if (!rc) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0) ctx->num_cus = prop.multiProcessorCount;
        ctx->wg_max = ctx->num_cus;
        if (const char* e = std::getenv("RTFHE_FORCE_WAVES")) ctx->force_waves = std::atoi(e);
        if (const char* e = std::getenv("RTFHE_WG_MAX_GATES")) ctx->wg_max = std::atoi(e);
        if (const char* e = std::getenv("RTFHE_STAGING")) ctx->stage_pinned = std::atoi(e) != 0;
        if (const char* e = std::getenv("RTFHE_KS_MM_MIN")) ctx->ks_mm_min = std::atoi(e);
        if (const char* e = std::getenv("RTFHE_N2048_EO4")) ctx->eo4 = std::atoi(e) != 0;
        if (const char* e = std::getenv("RTFHE_PAIR4")) ctx->pair4 = std::atoi(e);
        if (const char* e = std::getenv("RTFHE_N2048_KERNEL")) ctx->n2048_kernel = std::string(e) == "halves" ? 1 : std::string(e) == "eo" ? 0 : -1;
    }
    if (!rc) rc = prime_kernel_attributes(ctx);
    if (!rc) rc = upload_twiddles(ctx);
    if (!rc && (hipMalloc((void**)&ctx->d_fault, 4) != hipSuccess || hipMemset(ctx->d_fault, 0, 4) != hipSuccess))
        rc = fail(ctx, RTFHE_ERR_HIP, "hipMalloc failed");
    if (!rc && hipStreamCreate(&ctx->stream) != hipSuccess) rc = fail(ctx, RTFHE_ERR_HIP, "hipStreamCreate failed");
    if (!rc && (hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess))
        rc = fail(ctx, RTFHE_ERR_HIP, "hipEventCreate failed");
    if (rc) { g_last_error = ctx->err; rtfhe_ctx_destroy(ctx); return rc; }
#ifdef RTFHE_WG_STAMPS
    if (hipMalloc((void**)&ctx->d_dbg, 128 * 8) == hipSuccess) (void)hipMemset(ctx->d_dbg, 0, 128 * 8);
#endif
    *out = ctx;
    return 0;
}

int rtfhe_ctx_create(const rtfhe_params* p, int device_id, rtfhe_ctx** out) { return create_single(p, device_id, out); }

// One context over several GPUs of the node: device_ids[0] is the primary.  Keys loaded into the context are
// transformed once on the primary and copied device-to-device to the others; host-pointer batch calls (rtfhe_gate_batch,
// rtfhe_mux_batch, rtfhe_bootstrap_batch) shard contiguous gate ranges over the devices, one host thread and one stream per
// device, with direct host<->device copies per device.  *_dev and stage-level calls run on the primary device.
int rtfhe_ctx_create_multi(const rtfhe_params* p, const int* device_ids, int n_dev, rtfhe_ctx** out) {
    if (!p || !out || !device_ids) return fail(nullptr, RTFHE_ERR_INVALID, "null argument");
    *out = nullptr;
    if (n_dev < 1 || n_dev > 64) return fail(nullptr, RTFHE_ERR_INVALID, "n_dev out of range");
    // TEST ONLY: RTFHE_TEST_ALLOW_DUP_DEVICES=1 lets one GPU appear several times, so that the multi-device paths (per-device host
    // threads, device-to-device key replication, shard offsets, error aggregation) run on a one-GPU box
    const char* dup = std::getenv("RTFHE_TEST_ALLOW_DUP_DEVICES");
    if (!(dup && std::atoi(dup) != 0))
        for (int a = 0; a < n_dev; a++)
            for (int b = a + 1; b < n_dev; b++)
                if (device_ids[a] == device_ids[b]) return fail(nullptr, RTFHE_ERR_INVALID, "device_ids must be distinct");
    rtfhe_ctx* ctx = nullptr;
    if (int rc = create_single(p, device_ids[0], &ctx)) return rc;
    for (int d = 1; d < n_dev; d++) {
        rtfhe_ctx* peer = nullptr;
        if (int rc = create_single(p, device_ids[d], &peer)) { rtfhe_ctx_destroy(ctx); return rc; }
        ctx->peers.push_back(peer);
    }
    (void)hipSetDevice(ctx->device);
    *out = ctx;
    return 0;
}

int rtfhe_ctx_device_count(const rtfhe_ctx* ctx) { return ctx ? 1 + (int)ctx->peers.size() : 0; }

// the contiguous gate range entry d of an n_dev-device context takes of a host batch of `count` gates (pure arithmetic: what
// run_host_bootstrap / rtfhe_mux_batch use, exported so that a caller can lay out per-device buffers; no GPU needed)
int rtfhe_shard_range(size_t count, int d, int n_dev, size_t* begin, size_t* end) {
    if (n_dev < 1 || n_dev > 64 || d < 0 || d >= n_dev || !begin || !end) return fail(nullptr, RTFHE_ERR_INVALID, "rtfhe_shard_range: bad argument");
    *begin = shard_begin(count, d, n_dev);
    *end = shard_begin(count, d + 1, n_dev);
    return 0;
}

// pinned host memory for ciphertext buffers: host-pointer calls DMA straight from / into it (no staging copy)
void* rtfhe_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
void rtfhe_host_free(void* p) { if (p) (void)hipHostFree(p); }

// tuning builds only (-DPAIR_PRIO_RUNTIME / -DHALVES_PRIO_RUNTIME): the priority schedule the next launches try (scripts/tune_prio.py)
extern "C" int rtfhe_debug_set_tune(rtfhe_ctx* ctx, unsigned long long tune) {
    if (!ctx) return RTFHE_ERR_INVALID;
    ctx->tune = tune;
    return 0;
}

#ifdef RTFHE_WG_STAMPS
extern "C" int rtfhe_debug_read_stamps(rtfhe_ctx* ctx, unsigned long long* out128) {
    if (!ctx || !ctx->d_dbg) return RTFHE_ERR_STATE;
    return hipMemcpy(out128, ctx->d_dbg, 128 * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : RTFHE_ERR_HIP;
}
#endif

void rtfhe_ctx_destroy(rtfhe_ctx* ctx) {
    if (!ctx) return;
    for (rtfhe_ctx* peer : ctx->peers) rtfhe_ctx_destroy(peer);
    ctx->peers.clear();
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    // circuits that outlive their context: their graphs go now, the handles stay valid for rtfhe_circuit_destroy (which then
    // only frees them) and rtfhe_circuit_launch (which then fails with RTFHE_ERR_STATE)
    for (rtfhe_circuit* c : ctx->circuits) { circuit_release(c); c->ctx = nullptr; }
    ctx->circuits.clear();
    if (ctx->d_tw) (void)hipFree(ctx->d_tw);
    if (ctx->d_fault) (void)hipFree(ctx->d_fault);
    if (ctx->d_bk) (void)hipFree(ctx->d_bk);
    if (ctx->d_htw) (void)hipFree(ctx->d_htw);
    if (ctx->d_hbk) (void)hipFree(ctx->d_hbk);
    if (ctx->d_etw) (void)hipFree(ctx->d_etw);
    if (ctx->d_ebk) (void)hipFree(ctx->d_ebk);
    if (ctx->d_p4bk) (void)hipFree(ctx->d_p4bk);
    if (ctx->d_bk_torus) (void)hipFree(ctx->d_bk_torus);
    if (ctx->d_ntt_bk) (void)hipFree(ctx->d_ntt_bk);
    if (ctx->d_ntt_tw) (void)hipFree(ctx->d_ntt_tw);
    if (ctx->d_ksk) (void)hipFree(ctx->d_ksk);
    if (ctx->d_ksmat) (void)hipFree(ctx->d_ksmat);
    for (auto& kv : ctx->tlwe1) if (kv.second.d) (void)hipFree(kv.second.d);
    if (ctx->d_a) (void)hipFree(ctx->d_a);
    if (ctx->d_b) (void)hipFree(ctx->d_b);
    if (ctx->d_c) (void)hipFree(ctx->d_c);
    for (void* h : ctx->h_pin) if (h) (void)hipHostFree(h);
    for (void* m : ctx->h_mux) if (m) (void)hipFree(m);
    for (hipEvent_t e : ctx->ks_events) (void)hipEventDestroy(e);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int rtfhe_set_backend(rtfhe_ctx* ctx, int backend) {
    if (int rc = use(ctx)) return rc;
    if (backend != RTFHE_BACKEND_FFT64_MIRROR && backend != RTFHE_BACKEND_NTT_EXACT) return fail(ctx, RTFHE_ERR_INVALID, "unknown backend");
    ctx->backend = backend;
    for (rtfhe_ctx* peer : ctx->peers) peer->backend = backend;
    return 0;
}

int rtfhe_get_backend(const rtfhe_ctx* ctx) { return ctx ? ctx->backend : RTFHE_ERR_INVALID; }

int rtfhe_ctx_params(const rtfhe_ctx* ctx, rtfhe_params* p) {
    if (!ctx || !p) return fail(nullptr, RTFHE_ERR_INVALID, "null argument");
    *p = ctx->p;
    return 0;
}

int rtfhe_get_twiddles(const rtfhe_ctx* ctx, double* ifft_table, double* fft_table) {
    if (!ctx || !ifft_table || !fft_table) return fail(nullptr, RTFHE_ERR_INVALID, "null argument");
    ctx->tw.export_ref(ifft_table, fft_table);
    return 0;
}

int rtfhe_set_twiddles(rtfhe_ctx* ctx, const double* ifft_table, const double* fft_table) {
    if (int rc = use(ctx)) return rc;
    if (!ifft_table || !fft_table) return fail(ctx, RTFHE_ERR_INVALID, "null argument");
    const HostTw before = ctx->tw;
    ctx->tw.import_ref(ifft_table, fft_table);
    HIPCHECK(ctx, hipDeviceSynchronize());
    if (int rc = upload_twiddles(ctx)) { ctx->tw = before; return rc; }      // a refused table leaves the context as it was
    // a key loaded in torus form was transformed with the old tables: redo it (spectra loaded through rtfhe_load_bk_fft
    // are the caller's and stay as they are)
    if (ctx->has_bk && ctx->d_bk_torus)
        if (int rc = transform_bk_from_torus(ctx)) return rc;
    for (rtfhe_ctx* peer : ctx->peers) {
        if (int rc = rtfhe_set_twiddles(peer, ifft_table, fft_table)) return fail(ctx, rc, peer->err);
        HIPCHECK(ctx, hipSetDevice(ctx->device));
    }
    return 0;
}

// Twiddle tables as a file (the tables are libm-dependent DATA -- cos / sin of a double-rounded angle; two hosts' libms may differ
// by an ulp in a few entries, and one differing entry changes torus words).  A deployment that must reproduce a given reference build's bits
// ships that build's tables: rustfhe_amd/assets/ holds the tables of the reference build the golden vectors were made with.
// rtfhe_twiddles_load compares the file's tables with the context's (built with this host's libm at rtfhe_ctx_create) and, only if they
// differ, installs the file's (rtfhe_set_twiddles: a key loaded in torus form is re-transformed).  *entries_changed = table entries that
// differed (0: this host's libm agrees, nothing was done).  File I/O and checksum: rtfhe_wire.cpp.
int rtfhe_twiddles_write(const rtfhe_ctx* ctx, const char* path) {
    if (!ctx || !path) return fail(nullptr, RTFHE_ERR_INVALID, "null argument");
    std::vector<double> t((size_t)4 * ctx->p.N);
    ctx->tw.export_ref(t.data(), t.data() + (size_t)2 * ctx->p.N);
    return rtfhe_twiddles_file_write(path, ctx->p.N, t.data(), t.data() + (size_t)2 * ctx->p.N);
}

int rtfhe_twiddles_load(rtfhe_ctx* ctx, const char* path, int32_t* entries_changed) {
    if (!ctx || !path) return fail(nullptr, RTFHE_ERR_INVALID, "null argument");
    if (entries_changed) *entries_changed = 0;
    const size_t n2 = (size_t)2 * ctx->p.N;
    std::vector<double> t(2 * n2), cur(2 * n2);
    if (rtfhe_twiddles_file_read(path, ctx->p.N, t.data(), t.data() + n2) != 0)
        return fail(ctx, RTFHE_ERR_INVALID, "twiddle table file: unreadable, wrong degree or checksum mismatch");
    ctx->tw.export_ref(cur.data(), cur.data() + n2);
    int32_t diff = 0;
    for (size_t i = 0; i < t.size(); i++) diff += std::memcmp(&t[i], &cur[i], sizeof(double)) != 0;     // bits, not values: -0.0 vs +0.0 counts
    if (entries_changed) *entries_changed = diff;
    if (diff == 0) return 0;
    return rtfhe_set_twiddles(ctx, t.data(), t.data() + n2);
}

int rtfhe_load_bk_torus(rtfhe_ctx* ctx, const uint32_t* bk) {
    if (int rc = use(ctx)) return rc;
    if (!bk) return fail(ctx, RTFHE_ERR_INVALID, "null argument");
    const size_t words = bk_word_count(ctx->p);
    if (!ctx->d_bk) HIPCHECK(ctx, hipMalloc((void**)&ctx->d_bk, bk_cplx_count(ctx->p) * sizeof(cplx)));
    if (!ctx->d_bk_torus) HIPCHECK(ctx, hipMalloc((void**)&ctx->d_bk_torus, words * 4));
    HIPCHECK(ctx, hipMemcpy(ctx->d_bk_torus, bk, words * 4, hipMemcpyHostToDevice));
    ctx->ntt_ready = false;
    if (int rc = transform_bk_from_torus(ctx)) return rc;
    ctx->has_bk = true;
    for (rtfhe_ctx* peer : ctx->peers) {      // the transformed key and its torus form, device to device
        if (int rc = replicate(ctx, peer, ctx->d_bk, (void**)&peer->d_bk, bk_cplx_count(ctx->p) * sizeof(cplx))) return rc;
        if (int rc = replicate(ctx, peer, ctx->d_bk_torus, (void**)&peer->d_bk_torus, words * 4)) return rc;
        if (int rc = build_halves_bk(peer)) return fail(ctx, rc, peer->err);
        HIPCHECK(ctx, hipSetDevice(ctx->device));
        peer->ntt_ready = false; peer->has_bk = true;
    }
    return 0;
}

int rtfhe_load_bk_fft(rtfhe_ctx* ctx, const double* bk_f) {
    if (int rc = use(ctx)) return rc;
    if (!bk_f) return fail(ctx, RTFHE_ERR_INVALID, "null argument");
    const size_t words = bk_word_count(ctx->p);
    if (!ctx->d_bk) HIPCHECK(ctx, hipMalloc((void**)&ctx->d_bk, bk_cplx_count(ctx->p) * sizeof(cplx)));
    if (int rc = ensure(ctx, &ctx->d_a, &ctx->cap_a, words * 8)) return rc;
    HIPCHECK(ctx, hipMemcpy(ctx->d_a, bk_f, words * 8, hipMemcpyHostToDevice));
    const size_t polys = words / ctx->p.N;
    int rc = ctx->logn == 10 ? launch_permute_t<10>(ctx, (const double*)ctx->d_a, (double*)ctx->d_bk, polys, 0, 2 * ctx->p.l, ctx->stream)
                             : launch_permute_t<11>(ctx, (const double*)ctx->d_a, (double*)ctx->d_bk, polys, 0, 2 * ctx->p.l, ctx->stream);
    if (rc) return rc;
    HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_bk_torus) { (void)hipFree(ctx->d_bk_torus); ctx->d_bk_torus = nullptr; }   // no torus form of this key
    if (int rc = build_halves_bk(ctx)) return rc;
    ctx->ntt_ready = false;
    ctx->has_bk = true;
    for (rtfhe_ctx* peer : ctx->peers) {
        if (int rc = replicate(ctx, peer, ctx->d_bk, (void**)&peer->d_bk, bk_cplx_count(ctx->p) * sizeof(cplx))) return rc;
        if (int rc = build_halves_bk(peer)) return fail(ctx, rc, peer->err);
        HIPCHECK(ctx, hipSetDevice(ctx->device));
        if (peer->d_bk_torus) { (void)hipSetDevice(peer->device); (void)hipFree(peer->d_bk_torus); peer->d_bk_torus = nullptr; (void)hipSetDevice(ctx->device); }
        peer->ntt_ready = false; peer->has_bk = true;
    }
    return 0;
}

int rtfhe_export_bk_fft(rtfhe_ctx* ctx, double* bk_f) {
    if (int rc = use(ctx)) return rc;
    if (!bk_f) return fail(ctx, RTFHE_ERR_INVALID, "null argument");
    if (!ctx->has_bk) return fail(ctx, RTFHE_ERR_STATE, "bootstrapping key not loaded");
    const size_t words = bk_word_count(ctx->p);
    if (int rc = ensure(ctx, &ctx->d_a, &ctx->cap_a, words * 8)) return rc;
    const size_t polys = words / ctx->p.N;
    int rc = ctx->logn == 10 ? launch_permute_t<10>(ctx, (const double*)ctx->d_bk, (double*)ctx->d_a, polys, 1, 2 * ctx->p.l, ctx->stream)
                             : launch_permute_t<11>(ctx, (const double*)ctx->d_bk, (double*)ctx->d_a, polys, 1, 2 * ctx->p.l, ctx->stream);
    if (rc) return rc;
    HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHECK(ctx, hipMemcpy(bk_f, ctx->d_a, words * 8, hipMemcpyDeviceToHost));
    return 0;
}

int rtfhe_load_ksk(rtfhe_ctx* ctx, const uint32_t* ksk) {
    if (int rc = use(ctx)) return rc;
    if (!ksk) return fail(ctx, RTFHE_ERR_INVALID, "null argument");
    const size_t rows = ksk_rows(ctx->p), w = (size_t)ctx->p.n + 1, ksw = (size_t)ctx->ksw;
    // staging: rows padded to a multiple of 4 words (16-byte loads) + one all-zero row
    std::vector<uint32_t> padded((rows + 1) * ksw, 0u);
    for (size_t r = 0; r < rows; r++) std::memcpy(padded.data() + r * ksw, ksk + r * w, w * 4);
    if (int rc = ensure(ctx, &ctx->d_a, &ctx->cap_a, padded.size() * 4)) return rc;
    HIPCHECK(ctx, hipMemcpy(ctx->d_a, padded.data(), padded.size() * 4, hipMemcpyHostToDevice));
    // device layout: the rows of two adjacent levels pre-summed (see ks_accumulate) + one all-zero row that "both digits 0" selects
    const size_t dev_rows = (size_t)ks_dev_rows(ctx->p.N, ctx->p.ks_t, ctx->p.ks_basebit);
    if (!ctx->d_ksk) HIPCHECK(ctx, hipMalloc((void**)&ctx->d_ksk, (dev_rows + 1) * ksw * 4));
    KskCombineArgs a{(const uint32_t*)ctx->d_a, ctx->d_ksk, ctx->p.N, ctx->ksw};
    hipLaunchKernelGGL((k_ksk_combine<8, 2>), dim3(4096), dim3(256), 0, ctx->stream, a);
    HIPCHECK(ctx, hipGetLastError());
    HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
    // the same key as signed byte limbs in i8-MFMA operand order, for the batch key switch of the split path
    size_t ksmat_bytes = 0;
    if (ctx->ks_mm_min > 0) {
        const int colgroups = (ctx->p.n + 1 + 15) / 16;
        ksmat_bytes = (size_t)colgroups * (ctx->p.N / 2) * 4 * 64 * sizeof(uint4);
        if (!ctx->d_ksmat) HIPCHECK(ctx, hipMalloc((void**)&ctx->d_ksmat, ksmat_bytes));
        KsMatArgs m{(const uint32_t*)ctx->d_a, ctx->d_ksmat, ctx->p.N, ctx->p.n, ctx->ksw, colgroups};
        hipLaunchKernelGGL((k_ksmat_build<8, 2>), dim3(4096), dim3(256), 0, ctx->stream, m);
        HIPCHECK(ctx, hipGetLastError());
        HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
        if (int rc = ensure_tlwe1(ctx, ctx->tlwe1[ctx->stream], 8192)) return rc;     // the context's own stream (host-pointer calls)
    }
    ctx->has_ksk = true;
    for (rtfhe_ctx* peer : ctx->peers) {
        if (int rc = replicate(ctx, peer, ctx->d_ksk, (void**)&peer->d_ksk, (dev_rows + 1) * ksw * 4)) return rc;
        if (ksmat_bytes)
            if (int rc = replicate(ctx, peer, ctx->d_ksmat, (void**)&peer->d_ksmat, ksmat_bytes)) return rc;
        peer->has_ksk = true;
    }
    return 0;
}

// The reference's own container shape: KeySwitchingKey(Vec<[[TLWERep<M>; IKS_T]; IKS_L]>) with IKS_T = 2^IKS_BASEBIT = 4 entries per
// level (hom_nand/src/tlwe.rs:178-180, 243-245); entry t - 1 holds TLWE(t * s_i / 2^(basebit (l+1))) for t = 1 .. 4 (:252-274) and
// get(i, l, t) reads [i][l][t - 1] (:281-283).  identity_key_switch only ever asks for t = digit in 1 .. 3 (:43-73: a basebit-wide
// digit), so the 4th entry of every level is never read: it is dropped here and the rest goes through rtfhe_load_ksk.
int rtfhe_load_ksk_ref(rtfhe_ctx* ctx, const uint32_t* ksk_ref) {
    if (int rc = use(ctx)) return rc;
    if (!ksk_ref) return fail(ctx, RTFHE_ERR_INVALID, "null argument");
    const size_t w = (size_t)ctx->p.n + 1, base = (size_t)1 << ctx->p.ks_basebit, levels = (size_t)ctx->p.N * ctx->p.ks_t;
    std::vector<uint32_t> compact;
    try { compact.resize(levels * (base - 1) * w); } catch (const std::bad_alloc&) { return fail(ctx, RTFHE_ERR_NOMEM, "host staging for the key-switching key"); }
    for (size_t il = 0; il < levels; il++)
        std::memcpy(compact.data() + il * (base - 1) * w, ksk_ref + il * base * w, (base - 1) * w * sizeof(uint32_t));
    return rtfhe_load_ksk(ctx, compact.data());
}

int rtfhe_gate_batch_dev(rtfhe_ctx* ctx, int op, const void* d_in0, const void* d_in1, void* d_out, size_t count, void* stream) {
    if (int rc = use(ctx)) return rc;
    if (op < RTFHE_NAND || op > RTFHE_ANDNY) return fail(ctx, RTFHE_ERR_INVALID, "unknown gate");
    if (!d_in0 || !d_out) return fail(ctx, RTFHE_ERR_INVALID, "null argument");
    if (!gpu_accessible(ctx, d_in0) || (d_in1 && !gpu_accessible(ctx, d_in1)) || !gpu_accessible(ctx, d_out))
        return fail(ctx, RTFHE_ERR_INVALID, "rtfhe_gate_batch_dev needs device pointers (got memory the GPU cannot address)");
    return launch_bootstrap(ctx, op, MODE_GATE, ctx->p.n, d_in0, d_in1, d_out, count, (hipStream_t)stream);
}

int rtfhe_circuit_wave_dev(rtfhe_ctx* ctx, const void* d_ops, const void* d_idx0, const void* d_idx1, const void* d_idx_out,
                           void* d_wires, size_t num_wires, size_t count, void* stream) {
    if (int rc = use(ctx)) return rc;
    if (!d_ops || !d_idx0 || !d_idx1 || !d_idx_out || !d_wires) return fail(ctx, RTFHE_ERR_INVALID, "null argument");
    if (num_wires == 0 || num_wires > 0x7fffffff) return fail(ctx, RTFHE_ERR_INVALID, "num_wires out of range");
    if (!gpu_accessible(ctx, d_ops) || !gpu_accessible(ctx, d_idx0) || !gpu_accessible(ctx, d_idx1) || !gpu_accessible(ctx, d_idx_out) || !gpu_accessible(ctx, d_wires))
        return fail(ctx, RTFHE_ERR_INVALID, "rtfhe_circuit_wave_dev needs device pointers (got memory the GPU cannot address)");
    return launch_bootstrap(ctx, RTFHE_COPY, MODE_GATE, ctx->p.n, d_wires, d_wires, d_wires, count, (hipStream_t)stream,
                            (const int32_t*)d_ops, (const int32_t*)d_idx0, (const int32_t*)d_idx1, (const int32_t*)d_idx_out,
                            (int32_t)num_wires);
}

// ---- a whole levelised netlist as ONE submission: its dependency waves captured once into a HIP graph, replayed per run ----
int rtfhe_circuit_create(rtfhe_ctx* ctx, const void* d_ops, const void* d_idx0, const void* d_idx1, const void* d_idx_out,
                         const int32_t* wave_offsets, int32_t num_waves, void* d_wires, size_t num_wires, rtfhe_circuit** out) {
    if (int rc = use(ctx)) return rc;
    if (!out) return fail(ctx, RTFHE_ERR_INVALID, "null argument");
    *out = nullptr;
    if (!d_ops || !d_idx0 || !d_idx1 || !d_idx_out || !d_wires || !wave_offsets) return fail(ctx, RTFHE_ERR_INVALID, "null argument");
    if (num_waves < 1 || num_wires == 0 || num_wires > 0x7fffffff) return fail(ctx, RTFHE_ERR_INVALID, "num_waves / num_wires out of range");
    for (int32_t w = 0; w < num_waves; w++)
        if (wave_offsets[w] < 0 || wave_offsets[w + 1] <= wave_offsets[w]) return fail(ctx, RTFHE_ERR_INVALID, "wave_offsets must be strictly increasing from >= 0");
    if (!ctx->has_bk || !ctx->has_ksk) return fail(ctx, RTFHE_ERR_STATE, "keys not loaded");
    if (!gpu_accessible(ctx, d_ops) || !gpu_accessible(ctx, d_idx0) || !gpu_accessible(ctx, d_idx1) || !gpu_accessible(ctx, d_idx_out) || !gpu_accessible(ctx, d_wires))
        return fail(ctx, RTFHE_ERR_INVALID, "rtfhe_circuit_create needs device pointers (got memory the GPU cannot address)");
    if (ctx->backend == RTFHE_BACKEND_NTT_EXACT)
        if (int rc = ntt_prepare(ctx)) return rc;          // nothing but kernel launches may happen inside the capture
    rtfhe_ctx::Tlwe1 cbuf;                                 // ... so the circuit's own sample buffer (split path) is allocated now
    if (ctx->ks_mm_min > 0 && ctx->d_ksmat) {
        size_t widest = 0;
        for (int32_t w = 0; w < num_waves; w++) widest = std::max(widest, (size_t)(wave_offsets[w + 1] - wave_offsets[w]));
        if (int rc = ensure_tlwe1(ctx, cbuf, widest)) return rc;
    }
    rtfhe_circuit* c = new (std::nothrow) rtfhe_circuit();
    if (!c) { if (cbuf.d) (void)hipFree(cbuf.d); return fail(ctx, RTFHE_ERR_NOMEM, "out of host memory"); }
    c->ctx = ctx; c->device = ctx->device; c->waves = num_waves; c->d_samples = cbuf.d;
    const int64_t before = ctx->launches;
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { circuit_release(c); delete c; return fail(ctx, RTFHE_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(e)); }
    e = hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) { circuit_release(c); delete c; return fail(ctx, RTFHE_ERR_HIP, std::string("hipStreamBeginCapture: ") + hipGetErrorString(e)); }
    int rc = 0;
    ctx->tlwe1_capture = cbuf.d ? &cbuf : nullptr;
    for (int32_t w = 0; w < num_waves && !rc; w++) {
        const size_t off = (size_t)wave_offsets[w], cnt = (size_t)(wave_offsets[w + 1] - wave_offsets[w]);
        rc = launch_bootstrap(ctx, RTFHE_COPY, MODE_GATE, ctx->p.n, d_wires, d_wires, d_wires, cnt, ctx->stream,
                              (const int32_t*)d_ops + off, (const int32_t*)d_idx0 + off, (const int32_t*)d_idx1 + off,
                              (const int32_t*)d_idx_out + off, (int32_t)num_wires);
    }
    ctx->tlwe1_capture = nullptr;
    e = hipStreamEndCapture(ctx->stream, &c->graph);
    c->launches = ctx->launches - before;
    ctx->launches = before;
    if (rc) { circuit_release(c); delete c; return rc; }
    if (e != hipSuccess) { circuit_release(c); delete c; return fail(ctx, RTFHE_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e)); }
    e = hipGraphInstantiate(&c->exec, c->graph, nullptr, nullptr, 0);
    if (e != hipSuccess) { circuit_release(c); delete c; return fail(ctx, RTFHE_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e)); }
    ctx->circuits.push_back(c);
    *out = c;
    return 0;
}

int rtfhe_circuit_launch(rtfhe_circuit* c, void* stream) {
    if (!c) return fail(nullptr, RTFHE_ERR_INVALID, "null circuit");
    if (!c->ctx) return fail(nullptr, RTFHE_ERR_STATE, "the circuit's context has been destroyed");
    rtfhe_ctx* ctx = c->ctx;
    if (int rc = use(ctx)) return rc;
    HIPCHECK(ctx, hipGraphLaunch(c->exec, (hipStream_t)stream));
    ctx->launches += c->launches;
    return 0;
}

void rtfhe_circuit_destroy(rtfhe_circuit* c) {
    if (!c) return;
    if (c->ctx) {      // still attached: unregister (a context destroyed first has already released the graph and detached us)
        auto& v = c->ctx->circuits;
        for (size_t i = 0; i < v.size(); i++) if (v[i] == c) { v.erase(v.begin() + i); break; }
        circuit_release(c);
    }
    delete c;
}

int rtfhe_sync(rtfhe_ctx* ctx, void* stream) {
    if (int rc = use(ctx)) return rc;
    HIPCHECK(ctx, hipStreamSynchronize((hipStream_t)stream));
    // netlist waves validate their indices on the device; a skipped gate is reported here, once
    int32_t fault = 0;
    HIPCHECK(ctx, hipMemcpy(&fault, ctx->d_fault, 4, hipMemcpyDeviceToHost));
    if (fault) {
        HIPCHECK(ctx, hipMemset(ctx->d_fault, 0, 4));
        return fail(ctx, RTFHE_ERR_INVALID, "netlist wave: wire index or opcode out of range (those gates were skipped)");
    }
    return 0;
}

int rtfhe_timer_begin(rtfhe_ctx* ctx, void* stream) {
    if (int rc = use(ctx)) return rc;
    ctx->launches = 0;
    ctx->timing = true;
    ctx->ks_events_used = 0;
    HIPCHECK(ctx, hipEventRecord(ctx->ev0, (hipStream_t)stream));
    return 0;
}

// total device time between begin and end, and of it the time inside the batch key switches of the split path (memset +
// k_key_switch_mm; 0 when every launch was the fused kernel): total - key_switch = the blind-rotation kernels (+ launch gaps)
int rtfhe_timer_end_detail(rtfhe_ctx* ctx, void* stream, double* ms, double* key_switch_ms, int64_t* launches) {
    if (int rc = use(ctx)) return rc;
    ctx->timing = false;
    HIPCHECK(ctx, hipEventRecord(ctx->ev1, (hipStream_t)stream));
    HIPCHECK(ctx, hipEventSynchronize(ctx->ev1));
    float f = 0.f;
    HIPCHECK(ctx, hipEventElapsedTime(&f, ctx->ev0, ctx->ev1));
    if (ms) *ms = (double)f;
    double ks = 0.0;
    for (size_t i = 0; i + 1 < ctx->ks_events_used; i += 2) {
        float g = 0.f;
        HIPCHECK(ctx, hipEventElapsedTime(&g, ctx->ks_events[i], ctx->ks_events[i + 1]));
        ks += (double)g;
    }
    ctx->ks_events_used = 0;
    if (key_switch_ms) *key_switch_ms = ks;
    if (launches) *launches = ctx->launches;
    return 0;
}

int rtfhe_timer_end(rtfhe_ctx* ctx, void* stream, double* ms, int64_t* launches) {
    return rtfhe_timer_end_detail(ctx, stream, ms, nullptr, launches);
}

static int run_host_bootstrap_one(rtfhe_ctx* ctx, int op, int mode, int steps, const uint32_t* in0, const uint32_t* in1,
                                  uint32_t* out, size_t count, size_t out_words) {
    if (int rc = use(ctx)) return rc;
    if (count == 0) return 0;
    const size_t in_bytes = count * ((size_t)ctx->p.n + 1) * 4, out_bytes = count * out_words * 4;
    if (int rc = ensure(ctx, &ctx->d_a, &ctx->cap_a, in_bytes)) return rc;
    if (int rc = ensure(ctx, &ctx->d_c, &ctx->cap_c, out_bytes)) return rc;
    if (int rc = copy_in(ctx, ctx->d_a, in0, in_bytes, 0)) return rc;
    const void* d1 = nullptr;
    if (in1) {
        if (int rc = ensure(ctx, &ctx->d_b, &ctx->cap_b, in_bytes)) return rc;
        if (int rc = copy_in(ctx, ctx->d_b, in1, in_bytes, 1)) return rc;
        d1 = ctx->d_b;
    }
    if (int rc = launch_bootstrap(ctx, op, mode, steps, ctx->d_a, d1, ctx->d_c, count, ctx->stream)) return rc;
    return copy_out(ctx, out, ctx->d_c, out_bytes, 2);
}

// host-pointer batch: on a multi-device context device d bootstraps the contiguous range [count d / D, count (d+1) / D)
static int run_host_bootstrap(rtfhe_ctx* ctx, int op, int mode, int steps, const uint32_t* in0, const uint32_t* in1,
                              uint32_t* out, size_t count, size_t out_words) {
    if (int rc = use(ctx)) return rc;
    if (!in0 || !out) return fail(ctx, RTFHE_ERR_INVALID, "null argument");
    if (ctx->peers.empty()) return run_host_bootstrap_one(ctx, op, mode, steps, in0, in1, out, count, out_words);
    const int n_dev = 1 + (int)ctx->peers.size();
    const size_t w = (size_t)ctx->p.n + 1;
    return for_each_device(ctx, [&](rtfhe_ctx* c, int d) {
        const size_t b = shard_begin(count, d, n_dev), e = shard_begin(count, d + 1, n_dev);
        return run_host_bootstrap_one(c, op, mode, steps, in0 + b * w, in1 ? in1 + b * w : nullptr, out + b * out_words, e - b, out_words);
    });
}

int rtfhe_gate_batch(rtfhe_ctx* ctx, int op, const uint32_t* in0, const uint32_t* in1, uint32_t* out, size_t count) {
    if (!ctx) return fail(nullptr, RTFHE_ERR_INVALID, "null context");
    if (op < RTFHE_NAND || op > RTFHE_ANDNY) return fail(ctx, RTFHE_ERR_INVALID, "unknown gate");
    const bool unary = (op == RTFHE_NOT || op == RTFHE_COPY);
    if (!unary && !in1) return fail(ctx, RTFHE_ERR_INVALID, "binary gate needs two inputs");
    return run_host_bootstrap(ctx, op, MODE_GATE, ctx->p.n, in0, unary ? nullptr : in1, out, count, (size_t)ctx->p.n + 1);
}

int rtfhe_bootstrap_batch(rtfhe_ctx* ctx, const uint32_t* tlwe, uint32_t* out, size_t count) {
    return rtfhe_gate_batch(ctx, RTFHE_COPY, tlwe, nullptr, out, count);
}

int rtfhe_blind_rotate_batch(rtfhe_ctx* ctx, const uint32_t* tlwe, int32_t steps, uint32_t* acc, size_t count) {
    if (!ctx) return fail(nullptr, RTFHE_ERR_INVALID, "null context");
    if (steps < 0 || steps > ctx->p.n) return fail(ctx, RTFHE_ERR_INVALID, "steps out of range");
    return run_host_bootstrap(ctx, RTFHE_COPY, MODE_BLIND_ROTATE, steps, tlwe, nullptr, acc, count, (size_t)2 * ctx->p.N);
}

// hom_mux (tfhe.rs:27-40): i1 = AND(c, in1); i0 = AND(-c, in0); bootstrap(i1 + i0 + 1/8) -- the last line is hom_or(i1, i0).
// One copy in (c, in0, in1), three launches back to back on the context's stream with i1 / i0 kept on the device, one copy out.
static int mux_one(rtfhe_ctx* ctx, const uint32_t* c, const uint32_t* in0, const uint32_t* in1, uint32_t* out, size_t count) {
    if (int rc = use(ctx)) return rc;
    if (count == 0) return 0;
    const size_t bytes = count * ((size_t)ctx->p.n + 1) * 4;
    if (int rc = ensure(ctx, &ctx->d_a, &ctx->cap_a, bytes)) return rc;
    if (int rc = ensure(ctx, &ctx->d_b, &ctx->cap_b, bytes)) return rc;
    if (int rc = ensure(ctx, &ctx->d_c, &ctx->cap_c, bytes)) return rc;
    if (ctx->cap_mux < bytes) {
        for (void*& m : ctx->h_mux) { if (m) HIPCHECK(ctx, hipFree(m)); m = nullptr; }
        ctx->cap_mux = 0;
        for (void*& m : ctx->h_mux) HIPCHECK(ctx, hipMalloc(&m, bytes));
        ctx->cap_mux = bytes;
    }
    if (int rc = copy_in(ctx, ctx->d_a, c, bytes, 0)) return rc;
    if (int rc = copy_in(ctx, ctx->d_b, in1, bytes, 1)) return rc;
    if (int rc = copy_in(ctx, ctx->d_c, in0, bytes, 2)) return rc;
    const int n = ctx->p.n;
    if (int rc = launch_bootstrap(ctx, RTFHE_AND, MODE_GATE, n, ctx->d_a, ctx->d_b, ctx->h_mux[0], count, ctx->stream)) return rc;     // i1
    if (int rc = launch_bootstrap(ctx, RTFHE_ANDNY, MODE_GATE, n, ctx->d_a, ctx->d_c, ctx->h_mux[1], count, ctx->stream)) return rc;   // i0
    if (int rc = launch_bootstrap(ctx, RTFHE_OR, MODE_GATE, n, ctx->h_mux[0], ctx->h_mux[1], ctx->d_a, count, ctx->stream)) return rc;
    return copy_out(ctx, out, ctx->d_a, bytes, 0);
}

int rtfhe_mux_batch(rtfhe_ctx* ctx, const uint32_t* c, const uint32_t* in0, const uint32_t* in1, uint32_t* out, size_t count) {
    if (int rc = use(ctx)) return rc;
    if (!c || !in0 || !in1 || !out) return fail(ctx, RTFHE_ERR_INVALID, "null argument");
    if (!ctx->has_bk) return fail(ctx, RTFHE_ERR_STATE, "bootstrapping key not loaded");
    if (!ctx->has_ksk) return fail(ctx, RTFHE_ERR_STATE, "key-switching key not loaded");
    if (count > 0x7fffffff) return fail(ctx, RTFHE_ERR_INVALID, "count too large");
    if (ctx->peers.empty()) return mux_one(ctx, c, in0, in1, out, count);
    const int n_dev = 1 + (int)ctx->peers.size();
    const size_t w = (size_t)ctx->p.n + 1;
    return for_each_device(ctx, [&](rtfhe_ctx* cx, int d) {
        const size_t b = shard_begin(count, d, n_dev), e = shard_begin(count, d + 1, n_dev);
        return mux_one(cx, c + b * w, in0 + b * w, in1 + b * w, out + b * w, e - b);
    });
}

int rtfhe_external_product_batch(rtfhe_ctx* ctx, const int32_t* bk_index, const uint32_t* trlwe, uint32_t* out, size_t count) {
    if (int rc = use(ctx)) return rc;
    if (!bk_index || !trlwe || !out) return fail(ctx, RTFHE_ERR_INVALID, "null argument");
    if (!ctx->has_bk) return fail(ctx, RTFHE_ERR_STATE, "bootstrapping key not loaded");
    if (count == 0) return 0;
    if (count > 0x7fffffff) return fail(ctx, RTFHE_ERR_INVALID, "count too large");
    for (size_t g = 0; g < count; g++)
        if (bk_index[g] < 0 || bk_index[g] >= ctx->p.n) return fail(ctx, RTFHE_ERR_INVALID, "bk_index out of range");
    const size_t bytes = count * 2 * (size_t)ctx->p.N * 4;
    if (int rc = ensure(ctx, &ctx->d_a, &ctx->cap_a, bytes)) return rc;
    if (int rc = ensure(ctx, &ctx->d_b, &ctx->cap_b, count * 4)) return rc;
    if (int rc = ensure(ctx, &ctx->d_c, &ctx->cap_c, bytes)) return rc;
    HIPCHECK(ctx, hipMemcpyAsync(ctx->d_a, trlwe, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHECK(ctx, hipMemcpyAsync(ctx->d_b, bk_index, count * 4, hipMemcpyHostToDevice, ctx->stream));
    int rc;
    if (ctx->backend == RTFHE_BACKEND_NTT_EXACT) {
        if ((rc = ntt_prepare(ctx))) return rc;
        if (ctx->logn == 11) {
            NttHalvesExtProdArgs a{ctx->d_ntt_tw, ctx->d_ntt_bk, (const int32_t*)ctx->d_b, (const uint32_t*)ctx->d_a, (uint32_t*)ctx->d_c, (int32_t)count};
            const size_t lds = NttHalvesLds::TW + (size_t)2 * 2048 * 4 + 2 * NttHalvesLds::XB;
            if ((rc = allow_lds(ctx, k_external_product_ntt_halves<3, 6>, lds))) return rc;
            hipLaunchKernelGGL((k_external_product_ntt_halves<3, 6>), dim3(a.count), dim3(128), lds, ctx->stream, a);
            HIPCHECK(ctx, hipGetLastError());
            HIPCHECK(ctx, hipMemcpyAsync(out, ctx->d_c, bytes, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
            return 0;
        }
        constexpr int W = 4;
        NttExtProdArgs a{ctx->d_ntt_tw, ctx->d_ntt_bk, (const int32_t*)ctx->d_b, (const uint32_t*)ctx->d_a, (uint32_t*)ctx->d_c, (int32_t)count};
        const size_t lds = ntt_lds_bytes(W, 0);
        if ((rc = allow_lds(ctx, k_external_product_ntt<3, 6, W>, lds))) return rc;
        hipLaunchKernelGGL((k_external_product_ntt<3, 6, W>), dim3((a.count + W - 1) / W), dim3(64 * W), lds, ctx->stream, a);
        HIPCHECK(ctx, hipGetLastError());
    } else {
        ExtProdArgs a{ctx->d_tw, ctx->d_bk, (const int32_t*)ctx->d_b, (const uint32_t*)ctx->d_a, (uint32_t*)ctx->d_c, (int32_t)count};
        rc = ctx->logn == 10 ? launch_extprod_t<10>(ctx, a, ctx->stream) : launch_extprod_t<11>(ctx, a, ctx->stream);
        if (rc) return rc;
    }
    HIPCHECK(ctx, hipMemcpyAsync(out, ctx->d_c, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int rtfhe_key_switch_batch(rtfhe_ctx* ctx, const uint32_t* tlwe1, uint32_t* out, size_t count) {
    if (int rc = use(ctx)) return rc;
    if (!tlwe1 || !out) return fail(ctx, RTFHE_ERR_INVALID, "null argument");
    if (!ctx->has_ksk) return fail(ctx, RTFHE_ERR_STATE, "key-switching key not loaded");
    if (count == 0) return 0;
    if (count > 0x7fffffff) return fail(ctx, RTFHE_ERR_INVALID, "count too large");
    const size_t in_bytes = count * ((size_t)ctx->p.N + 1) * 4, out_bytes = count * ((size_t)ctx->p.n + 1) * 4;
    if (int rc = ensure(ctx, &ctx->d_a, &ctx->cap_a, in_bytes)) return rc;
    if (int rc = ensure(ctx, &ctx->d_c, &ctx->cap_c, out_bytes)) return rc;
    HIPCHECK(ctx, hipMemcpyAsync(ctx->d_a, tlwe1, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    KeySwitchArgs a{ctx->d_ksk, (const uint32_t*)ctx->d_a, (uint32_t*)ctx->d_c, (int32_t)count, ctx->p.n, ctx->ksw};
    int rc = ctx->logn == 10 ? launch_keyswitch_t<10>(ctx, a, ctx->stream) : launch_keyswitch_t<11>(ctx, a, ctx->stream);
    if (rc) return rc;
    HIPCHECK(ctx, hipMemcpyAsync(out, ctx->d_c, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

static int run_fft_batch(rtfhe_ctx* ctx, bool forward, bool f64_io, const void* src, void* res, size_t count) {
    if (int rc = use(ctx)) return rc;
    if (!src || !res) return fail(ctx, RTFHE_ERR_INVALID, "null argument");
    if (count == 0) return 0;
    if (count > 0x7fffffff) return fail(ctx, RTFHE_ERR_INVALID, "count too large");
    const size_t N = (size_t)ctx->p.N;
    const size_t in_bytes = count * N * ((forward && !f64_io) ? 4 : 8), out_bytes = count * N * ((forward || f64_io) ? 8 : 4);
    if (int rc = ensure(ctx, &ctx->d_a, &ctx->cap_a, in_bytes)) return rc;
    if (int rc = ensure(ctx, &ctx->d_c, &ctx->cap_c, out_bytes)) return rc;
    HIPCHECK(ctx, hipMemcpyAsync(ctx->d_a, src, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    FftArgs a{ctx->d_tw, ctx->d_a, ctx->d_c, (int32_t)count, 0, 0, f64_io ? 1 : 0};
    if (int rc = launch_fft(ctx, forward, a, ctx->stream)) return rc;
    HIPCHECK(ctx, hipMemcpyAsync(res, ctx->d_c, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int rtfhe_ifft_i32_batch(rtfhe_ctx* ctx, const int32_t* src, double* res, size_t count) {
    return run_fft_batch(ctx, true, false, src, res, count);
}

int rtfhe_fft_u32_batch(rtfhe_ctx* ctx, const double* src, uint32_t* res, size_t count) {
    return run_fft_batch(ctx, false, false, src, res, count);
}

int rtfhe_ifft_f64_batch(rtfhe_ctx* ctx, const double* src, double* res, size_t count) {
    return run_fft_batch(ctx, true, true, src, res, count);
}

int rtfhe_fft_f64_batch(rtfhe_ctx* ctx, const double* src, double* res, size_t count) {
    return run_fft_batch(ctx, false, true, src, res, count);
}

int rtfhe_poly_mul_batch(rtfhe_ctx* ctx, const uint32_t* a, const uint32_t* b, uint32_t* res, size_t count) {
    if (int rc = use(ctx)) return rc;
    if (!a || !b || !res) return fail(ctx, RTFHE_ERR_INVALID, "null argument");
    if (count == 0) return 0;
    if (count > 0x7fffffff) return fail(ctx, RTFHE_ERR_INVALID, "count too large");
    const size_t bytes = count * (size_t)ctx->p.N * 4;
    if (int rc = ensure(ctx, &ctx->d_a, &ctx->cap_a, bytes)) return rc;
    if (int rc = ensure(ctx, &ctx->d_b, &ctx->cap_b, bytes)) return rc;
    if (int rc = ensure(ctx, &ctx->d_c, &ctx->cap_c, bytes)) return rc;
    HIPCHECK(ctx, hipMemcpyAsync(ctx->d_a, a, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHECK(ctx, hipMemcpyAsync(ctx->d_b, b, bytes, hipMemcpyHostToDevice, ctx->stream));
    PolyMulArgs pa{ctx->d_tw, (const uint32_t*)ctx->d_a, (const uint32_t*)ctx->d_b, (uint32_t*)ctx->d_c, (int32_t)count};
    int rc = ctx->logn == 10 ? launch_poly_mul_t<10>(ctx, pa, ctx->stream) : launch_poly_mul_t<11>(ctx, pa, ctx->stream);
    if (rc) return rc;
    HIPCHECK(ctx, hipMemcpyAsync(res, ctx->d_c, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// rtfhe_fft_plan: the reference's transforms at ANY power of two 16 <= N <= 2048 (rtfhe_kernels_anyn.hpp).  The reference's FFI
// accepts every such N (Spqlios::new, utils/src/spqlios.rs:40-50; its unit test uses 16, :243-276); the gate path does not go
// through here.  Opaque handle, not thread-safe (one per thread, as the reference's thread_local FFT_MAP, math.rs:349-351).
// ------------------------------------------------------------------------------------------------
struct rtfhe_fft_plan {
    int32_t N = 0;
    int device = 0;
    HostTw tw;
    double* d_tab = nullptr;          // [8][N/2]
    void* d_in = nullptr; void* d_in2 = nullptr; void* d_out = nullptr;
    size_t cap = 0;                   // polynomials the staging buffers hold
    hipStream_t stream = nullptr;
};

namespace {

#define PLANCHECK(expr)                                                                              \
    do {                                                                                             \
        hipError_t e__ = (expr);                                                                     \
        if (e__ != hipSuccess)                                                                       \
            return fail(nullptr, RTFHE_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)); \
    } while (0)

int plan_upload(rtfhe_fft_plan* pl) {
    const int P = pl->N / 2;
    std::vector<double> t((size_t)8 * P);
    const std::vector<double>* src[8] = {&pl->tw.twist_c, &pl->tw.twist_s, &pl->tw.untw_c, &pl->tw.untw_s,
                                         &pl->tw.fwd_c, &pl->tw.fwd_s, &pl->tw.inv_c, &pl->tw.inv_s};
    for (int k = 0; k < 8; k++) std::memcpy(t.data() + (size_t)k * P, src[k]->data(), sizeof(double) * P);
    PLANCHECK(hipMemcpy(pl->d_tab, t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice));
    return 0;
}

int plan_run(rtfhe_fft_plan* pl, int mode, const void* src, const void* src2, void* res, size_t count) {
    if (!pl) return fail(nullptr, RTFHE_ERR_INVALID, "null plan");
    if (!src || !res || (mode == ANYN_POLY_MUL && !src2)) return fail(nullptr, RTFHE_ERR_INVALID, "null argument");
    if (count == 0) return 0;
    if (count > 0x7fffffff) return fail(nullptr, RTFHE_ERR_INVALID, "count too large");
    PLANCHECK(hipSetDevice(pl->device));
    const size_t N = (size_t)pl->N;
    if (pl->cap < count) {
        for (void** b : {&pl->d_in, &pl->d_in2, &pl->d_out}) { if (*b) PLANCHECK(hipFree(*b)); *b = nullptr; }
        pl->cap = 0;
        for (void** b : {&pl->d_in, &pl->d_in2, &pl->d_out}) PLANCHECK(hipMalloc(b, count * N * 8));
        pl->cap = count;
    }
    const size_t in_bytes = count * N * ((mode == ANYN_IFFT_I32 || mode == ANYN_POLY_MUL) ? 4 : 8);
    const size_t out_bytes = count * N * ((mode == ANYN_FFT_U32 || mode == ANYN_POLY_MUL) ? 4 : 8);
    PLANCHECK(hipMemcpyAsync(pl->d_in, src, in_bytes, hipMemcpyHostToDevice, pl->stream));
    if (mode == ANYN_POLY_MUL) PLANCHECK(hipMemcpyAsync(pl->d_in2, src2, in_bytes, hipMemcpyHostToDevice, pl->stream));
    AnyNArgs a{pl->d_tab, pl->d_in, pl->d_in2, pl->d_out, pl->N, (int32_t)count, mode};
    const unsigned grid = (unsigned)(count < 4096 ? count : 4096);
    hipLaunchKernelGGL(k_fft_anyn, dim3(grid), dim3(ANYN_THREADS), 0, pl->stream, a);
    PLANCHECK(hipGetLastError());
    PLANCHECK(hipMemcpyAsync(res, pl->d_out, out_bytes, hipMemcpyDeviceToHost, pl->stream));
    PLANCHECK(hipStreamSynchronize(pl->stream));
    return 0;
}

}  // namespace

extern "C" {

int rtfhe_fft_plan_create(int32_t N, int device_id, rtfhe_fft_plan** out) {
    if (!out) return fail(nullptr, RTFHE_ERR_INVALID, "null argument");
    *out = nullptr;
    if (N < 16 || N > 2 * ANYN_MAXP || (N & (N - 1))) return fail(nullptr, RTFHE_ERR_INVALID, "supported transform sizes: powers of two 16 <= N <= 2048");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, RTFHE_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device_id < 0 || device_id >= ndev) return fail(nullptr, RTFHE_ERR_INVALID, "device_id out of range");
    rtfhe_fft_plan* pl = new (std::nothrow) rtfhe_fft_plan();
    if (!pl) return fail(nullptr, RTFHE_ERR_NOMEM, "out of host memory");
    pl->N = N; pl->device = device_id;
    pl->tw.build(N);
    int rc = 0;
    if (hipSetDevice(device_id) != hipSuccess || hipMalloc((void**)&pl->d_tab, (size_t)8 * (N / 2) * sizeof(double)) != hipSuccess ||
        hipStreamCreate(&pl->stream) != hipSuccess)
        rc = fail(nullptr, RTFHE_ERR_HIP, "device set-up of the transform plan failed");
    if (!rc) rc = plan_upload(pl);
    if (rc) { rtfhe_fft_plan_destroy(pl); return rc; }
    *out = pl;
    return 0;
}

void rtfhe_fft_plan_destroy(rtfhe_fft_plan* pl) {
    if (!pl) return;
    (void)hipSetDevice(pl->device);
    if (pl->stream) { (void)hipStreamSynchronize(pl->stream); (void)hipStreamDestroy(pl->stream); }
    for (void* b : {(void*)pl->d_tab, pl->d_in, pl->d_in2, pl->d_out}) if (b) (void)hipFree(b);
    delete pl;
}

int32_t rtfhe_fft_plan_degree(const rtfhe_fft_plan* pl) { return pl ? pl->N : 0; }

int rtfhe_fft_plan_get_twiddles(const rtfhe_fft_plan* pl, double* ifft_table, double* fft_table) {
    if (!pl || !ifft_table || !fft_table) return fail(nullptr, RTFHE_ERR_INVALID, "null argument");
    pl->tw.export_ref(ifft_table, fft_table);
    return 0;
}

int rtfhe_fft_plan_set_twiddles(rtfhe_fft_plan* pl, const double* ifft_table, const double* fft_table) {
    if (!pl || !ifft_table || !fft_table) return fail(nullptr, RTFHE_ERR_INVALID, "null argument");
    PLANCHECK(hipSetDevice(pl->device));
    PLANCHECK(hipStreamSynchronize(pl->stream));
    pl->tw.import_ref(ifft_table, fft_table);
    return plan_upload(pl);
}

int rtfhe_fft_plan_ifft_i32(rtfhe_fft_plan* pl, const int32_t* src, double* res, size_t count) { return plan_run(pl, ANYN_IFFT_I32, src, nullptr, res, count); }
int rtfhe_fft_plan_ifft_f64(rtfhe_fft_plan* pl, const double* src, double* res, size_t count) { return plan_run(pl, ANYN_IFFT_F64, src, nullptr, res, count); }
int rtfhe_fft_plan_fft_u32(rtfhe_fft_plan* pl, const double* src, uint32_t* res, size_t count) { return plan_run(pl, ANYN_FFT_U32, src, nullptr, res, count); }
int rtfhe_fft_plan_fft_f64(rtfhe_fft_plan* pl, const double* src, double* res, size_t count) { return plan_run(pl, ANYN_FFT_F64, src, nullptr, res, count); }
int rtfhe_fft_plan_poly_mul(rtfhe_fft_plan* pl, const uint32_t* a, const uint32_t* b, uint32_t* res, size_t count) { return plan_run(pl, ANYN_POLY_MUL, a, b, res, count); }

}  // extern "C"
