// rtfhe_host.hpp -- host-side internals shared by the translation units behind the C ABI (include/rtfhe.h):
//   rtfhe_context.hip        contexts, keys (load / layouts on demand / footprint), twiddle-table calls, copies, error plumbing
//   rtfhe_twiddles.hip       host twiddle tables (the reference's builders restated) and the per-kernel device tables
//   rtfhe_dispatch_fft.hip   the FP64 mirror backend's bootstrap kernels: which kernel shape a batch runs on
//   rtfhe_dispatch_ntt.hip   the exact-integer NTT backend: host tables, key transform, kernel shapes
//   rtfhe_dispatch_xfft.hip  the split-FFT exact backend (exact products through an FMA-contracted FP64 FFT, N = 1024)
//   rtfhe_stages.hip         stage-level kernels and entry points (transforms, external product, key switch, key permutes), FFT plans at any N
//   rtfhe_batch.hip          batches of gates: the backend switch, split path, host-pointer and device-pointer batches, MUX, timers
//   rtfhe_circuit.hip        levelised netlists: one wave per call, or all waves recorded into a HIP graph
//   rtfhe_multi.hip          one context over several GPUs: key replication, sharding of host batches and of device-resident batches
// Every kernel is instantiated in exactly one of them.  No CPU fallback anywhere: an entry point runs HIP kernels or fails.
#pragma once

#include "../../include/rtfhe.h"

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "rtfhe_kernels.hpp"

// ------------------------------------------------------------------------------------------------
// twiddle tables (host).  Values follow the reference's table builders so that a context created in
// the same process / against the same libm as the reference holds the same bits:
//   accurate_cos / accurate_sin   utils/src/spqlios/spqlios-fft-impl.cpp:99-113
//   new_ifft_table                utils/src/spqlios/spqlios-fft-impl.cpp:400-437
//   new_fft_table                 utils/src/spqlios/spqlios-fft-impl.cpp:158-193
// ------------------------------------------------------------------------------------------------
struct HostTw {
    int N = 0;
    // per-stage natural order; forward stages concatenated halfnn = P/2 .. 4, inverse halfnn = 4 .. P/2
    std::vector<double> twist_c, twist_s, untw_c, untw_s, fwd_c, fwd_s, inv_c, inv_s;
    int fwd_off(int halfnn) const { return N / 2 - 2 * halfnn; }
    int inv_off(int halfnn) const { return halfnn - 4; }

    void build(int N_);
    // the reference's memory layout (2N doubles per direction: per stage, blocks | c0 c1 c2 c3 | s0 s1 s2 s3 |)
    void export_ref(double* ifft_table, double* fft_table) const;
    void import_ref(const double* ifft_table, const double* fft_table);
    // device tables, one per kernel family (rtfhe_twiddles.hip)
    std::vector<rtfhe::cplx> device_table(int logn) const;     // Geo<LOGN>: per direction [twist R*64][pass1 (R-1)*64][pass2 (R-1)*NLOW][pass3 NLOW-4]
    std::vector<rtfhe::cplx> eo_table() const;                 // EoTw: k_bootstrap_eo / _eo4 (N = 2048)
    std::vector<rtfhe::cplx> q4_table() const;                 // Q4Tw: the parity sub-networks of k_bootstrap_wg / _pair4 (N = 1024)
};

struct rtfhe_circuit;
struct rtfhe_ctx {
    rtfhe_params p{};
    int device = 0;
    int logn = 10;
    HostTw tw;
    rtfhe::cplx* d_tw = nullptr;
    rtfhe::cplx* d_bk = nullptr;             // key spectra, canonical device layout [n][2l][2][R][64] (every N)
    rtfhe::cplx* d_etw = nullptr;            // N = 2048: tables of k_bootstrap_eo
    // second layouts of the key spectra, built from d_bk by the first batch whose dispatch reads them (ensure_bk_layout) and dropped when the key changes
    rtfhe::cplx* d_ebk = nullptr;            // N = 2048: the even / odd layout of k_bootstrap_eo / _eo4
    rtfhe::cplx* d_p4bk = nullptr;           // N = 1024: the layout of k_bootstrap_pair4
    bool ebk_valid = false, p4bk_valid = false;
    int pair4 = 3;                    // N = 1024, four waves per gate (k_bootstrap_pair4) for batches and tails of more than wg_max gates and up to
                                      // `pair4` gates per CU (RTFHE_PAIR4: 0 = never, 2, 3 = default)
    int rr = 6;                       // N = 1024: a last whole round and the remainder behind it, up to `rr` gates per CU in all, as ONE time-sliced launch
                                      // (k_bootstrap_pair_rr; RTFHE_PAIR_RR: 0 = never, 5, 6 = default)
    int xrr = 0;                      // ... the same on the split-FFT exact backend (k_bootstrap_xpair_rr): min(rr, what fits), set when that backend's kernels are primed
    int eo4 = 1;                      // N = 2048, up to two gates per CU: 1 = four waves per gate (k_bootstrap_eo4), 0 = two (RTFHE_N2048_EO4)
    int backend = RTFHE_BACKEND_FFT64_MIRROR;
    uint32_t* d_bk_torus = nullptr;   // kept when the key came in torus form: source for the NTT-domain key
    double* d_ntt_bk = nullptr;
    double* d_ntt_tw = nullptr;
    bool ntt_ready = false;
    rtfhe::cplx* d_xbk = nullptr;     // split-FFT exact backend: the key as two 16-bit halves, each as spectra (2 x the canonical size)
    rtfhe::cplx* d_xtw = nullptr;
    bool xfft_ready = false;
    uint32_t* d_ksk = nullptr;
    int ksw = 0;
    uint4* d_ksmat = nullptr;         // the key-switching key as signed byte limbs in i8-MFMA operand order (rtfhe_kernels_ksmm.hpp)
    size_t ksk_bytes = 0, ksmat_bytes = 0;
    // lvl1 samples between the two launches of the split path: ONE buffer per stream a batch was ever launched on (launches of one stream are
    // ordered, launches on different streams of one context may overlap and must not share it), plus a circuit's own during its capture
    struct Tlwe1 { uint32_t* d = nullptr; size_t cap = 0; };     // cap in gates
    std::unordered_map<hipStream_t, Tlwe1> tlwe1;
    Tlwe1* tlwe1_capture = nullptr;   // set by rtfhe_circuit_create around its capture: the circuit's buffer
    bool foreign_capture = false;     // set by launch_bootstrap for the duration of a call made inside a stream capture that is NOT
                                      // rtfhe_circuit_create's: such a batch stays on the fused kernel (see split_ok)
    int ks_mm_min = 1;                // batches of at least this many gates take the split path (0 = never: fused kernel); RTFHE_KS_MM_MIN
    bool has_bk = false, has_ksk = false;
    void* d_a = nullptr; void* d_b = nullptr; void* d_c = nullptr;   // device staging for host-pointer calls (and a peer's shard of a device-resident batch)
    size_t cap_a = 0, cap_b = 0, cap_c = 0;
    void* h_pin[3] = {nullptr, nullptr, nullptr};                     // pinned host staging (pageable caller buffers go through it)
    size_t cap_pin[3] = {0, 0, 0};
    bool stage_pinned = false;                                        // RTFHE_STAGING=1: stage pageable caller buffers through h_pin (measured slower
                                                                      // than the runtime's own pageable path: +3.4 % vs +1.5 % at 1024 gates)
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev_shard = nullptr;    // multi-device, device-resident batches: "inputs ready" on the primary / "shard gathered" on a peer
    hipEvent_t ev_sh[3] = {nullptr, nullptr, nullptr};   // a peer's share of such a batch: before its pull, after it, after its bootstrap (ev_shard: after its push)
    bool shard_timed = false;         // ev_sh / ev_shard bracket a batch
    // what rtfhe_ctx_create_multi found out about this entry against the primary (rtfhe_ctx_peer_info)
    struct PeerLink { int same_device = 0, can_from = 0, can_to = 0, en_from = 0, en_to = 0; uint32_t link_type = 0xffffffffu, hops = 0; } link;
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
    // the set.  Keys are loaded once on this context and copied device-to-device; batches are sharded (rtfhe_multi.hip).
    std::vector<rtfhe_ctx*> peers;
    std::vector<rtfhe_circuit*> circuits;   // live HIP-graph circuits of this context: orphaned (not freed) by rtfhe_ctx_destroy
    // device intermediates (i1, i0) of a MUX batch: one pair per stream a MUX batch was ever launched on, as the lvl1 samples above (two MUX
    // batches on different streams of one context may overlap).  A pair whose addresses went into a caller's capture is never freed or replaced
    // while the context lives (`captured`): a later, larger eager batch on that stream gets a new pair and the old one moves to mux_retired.
    struct MuxBuf { void* m[2] = {nullptr, nullptr}; size_t cap = 0; bool captured = false; };
    std::unordered_map<hipStream_t, MuxBuf> mux;
    std::vector<void*> mux_retired;
    int num_cus = 256;
    int force_waves = 0;   // RTFHE_FORCE_WAVES=1|2|4|8: one kernel shape for every batch (the parity tests' second opinions)
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
    int backend = RTFHE_BACKEND_FFT64_MIRROR;   // the backend it was recorded on
    bool stale = false;        // recorded on an exact backend whose key form could not follow a key change (rebuild_derived_keys)
};

namespace rtfhe_host {

using rtfhe::BootstrapArgs;
using rtfhe::cplx;

constexpr int KSQ = 3;        // uint4 loads per lane per key-switch row: rows up to 768 words
constexpr int KSMM_MT = 4;    // gate tiles (of 16) per wave of k_key_switch_mm

// ---- errors (rtfhe_context.hip) ----
int fail(rtfhe_ctx* ctx, int code, const std::string& msg);
const std::string& last_error_of_thread();
#define HIPCHECK(ctx, expr)                                                                                     \
    do {                                                                                                        \
        hipError_t e__ = (expr);                                                                                \
        if (e__ != hipSuccess)                                                                                  \
            return rtfhe_host::fail(ctx, RTFHE_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));    \
    } while (0)

inline size_t bk_cplx_count(const rtfhe_params& p) { return (size_t)p.n * 2 * 2 * p.l * (p.N / 2); }
inline size_t bk_word_count(const rtfhe_params& p) { return (size_t)p.n * 2 * 2 * p.l * p.N; }
inline size_t ksk_rows(const rtfhe_params& p) { return (size_t)p.N * p.ks_t * ((1 << p.ks_basebit) - 1); }

// ---- context helpers (rtfhe_context.hip) ----
int use(rtfhe_ctx* ctx);                                                  // null check + hipSetDevice
int ensure(rtfhe_ctx* ctx, void** ptr, size_t* cap, size_t bytes);        // grow-only device buffer
int allow_lds_raw(rtfhe_ctx* ctx, const void* kernel, size_t bytes);      // raises the kernel's dynamic-LDS limit on this device once
template <typename K>
int allow_lds(rtfhe_ctx* ctx, K kernel, size_t bytes) { return allow_lds_raw(ctx, reinterpret_cast<const void*>(kernel), bytes); }
bool gpu_accessible(const rtfhe_ctx* ctx, const void* p);                 // may a kernel on ctx->device dereference p?
bool is_pinned_host(const void* p);
int copy_in(rtfhe_ctx* ctx, void* dst, const void* src, size_t bytes, int slot);      // host -> device on ctx->stream
int copy_out(rtfhe_ctx* ctx, void* dst, const void* src, size_t bytes, int slot);     // device -> host on ctx->stream, synchronous on return
void circuit_release(rtfhe_circuit* c);                                   // rtfhe_circuit.hip

// ---- twiddles (rtfhe_twiddles.hip) ----
bool unit_twiddles_ok(const HostTw& tw);
int upload_twiddles(rtfhe_ctx* ctx);

// ---- FP64 mirror backend (rtfhe_dispatch_fft.hip) ----
int prime_fft_kernels(rtfhe_ctx* ctx);                                    // grants every bootstrap kernel of the parameter set its dynamic LDS
int launch_bootstrap_fft(rtfhe_ctx* ctx, BootstrapArgs a, hipStream_t s);
// builds (outside any stream capture) the second key layouts the dispatch of a `count`-gate batch in `mode` will read; no-op when they stand
int ensure_bk_layouts(rtfhe_ctx* ctx, size_t count, int mode);
int rebuild_derived_keys(rtfhe_ctx* ctx);                               // after a key change: every derived key form that already exists, in place, now

// ---- exact-integer NTT backend (rtfhe_dispatch_ntt.hip) ----
int prime_ntt_kernels(rtfhe_ctx* ctx);
int ntt_prepare(rtfhe_ctx* ctx);                                          // tables + NTT-domain key on first use
int launch_bootstrap_ntt(rtfhe_ctx* ctx, BootstrapArgs a, hipStream_t s);
int launch_extprod_ntt(rtfhe_ctx* ctx, const int32_t* d_idx, const uint32_t* d_in, uint32_t* d_out, int32_t count, hipStream_t s);

// ---- split-FFT exact backend (rtfhe_dispatch_xfft.hip) ----
int prime_xfft_kernels(rtfhe_ctx* ctx);
int xfft_prepare(rtfhe_ctx* ctx);                                         // table + split key spectra on first use
int launch_bootstrap_xfft(rtfhe_ctx* ctx, BootstrapArgs a, hipStream_t s);
int launch_extprod_xfft(rtfhe_ctx* ctx, const int32_t* d_idx, const uint32_t* d_in, uint32_t* d_out, int32_t count, hipStream_t s);
// tables / derived keys of whichever exact backend is selected (no-op on the mirror): allocates and synchronises, so outside stream captures only
int backend_prepare(rtfhe_ctx* ctx);

// ---- stage kernels (rtfhe_stages.hip) ----
int launch_fft(rtfhe_ctx* ctx, bool forward, rtfhe::FftArgs a, hipStream_t s);
int launch_bk_permute(rtfhe_ctx* ctx, const double* src, double* dst, size_t polys, int dir, hipStream_t s);
int launch_ksk_combine(rtfhe_ctx* ctx, const uint32_t* d_raw, hipStream_t s);
int launch_ksmat_build(rtfhe_ctx* ctx, const uint32_t* d_raw, int colgroups, hipStream_t s);
// the identity key switch of a whole batch as one exact i8 contraction (second launch of the split path)
int launch_key_switch_mm(rtfhe_ctx* ctx, const BootstrapArgs& a, const uint32_t* samples, hipStream_t s);

// ---- batches (rtfhe_batch.hip) ----
int launch_bootstrap(rtfhe_ctx* ctx, int op, int mode, int steps, const void* d_in0, const void* d_in1, void* d_out,
                     size_t count, hipStream_t s, const int32_t* d_ops = nullptr, const int32_t* d_idx0 = nullptr,
                     const int32_t* d_idx1 = nullptr, const int32_t* d_idx_out = nullptr, int32_t num_wires = 0);
int ensure_tlwe1(rtfhe_ctx* ctx, rtfhe_ctx::Tlwe1& b, size_t gates);
int run_host_bootstrap_one(rtfhe_ctx* ctx, int op, int mode, int steps, const uint32_t* in0, const uint32_t* in1, uint32_t* out, size_t count, size_t out_words);
int mux_host_one(rtfhe_ctx* ctx, const uint32_t* c, const uint32_t* in0, const uint32_t* in1, uint32_t* out, size_t count);
int mux_dev_one(rtfhe_ctx* ctx, const void* d_c, const void* d_in0, const void* d_in1, void* d_out, size_t count, hipStream_t s);

// ---- several GPUs (rtfhe_multi.hip) ----
inline size_t shard_begin(size_t count, int d, int n_dev) { return count * (size_t)d / (size_t)n_dev; }     // contiguous ranges, sizes differ by at most one
int replicate(rtfhe_ctx* ctx, rtfhe_ctx* peer, const void* src, void** dst_of_peer, size_t bytes);           // the primary's copy of a key -> a peer, device to device
int sharded_host_bootstrap(rtfhe_ctx* ctx, int op, int mode, int steps, const uint32_t* in0, const uint32_t* in1, uint32_t* out, size_t count, size_t out_words);
int sharded_host_mux(rtfhe_ctx* ctx, const uint32_t* c, const uint32_t* in0, const uint32_t* in1, uint32_t* out, size_t count);
// a batch that LIVES ON THE PRIMARY DEVICE, sharded over the context's devices: op < 0 = MUX (d_c, d_in0, d_in1), else a gate batch (d_in0, d_in1)
int sharded_dev_batch(rtfhe_ctx* ctx, int op, const void* d_c, const void* d_in0, const void* d_in1, void* d_out, size_t count, hipStream_t s);

// words per gate of the output buffer, by mode (MODE_EXTRACT: the final TLWE rows; the lvl1 samples go to `ext`)
inline size_t mode_out_words(const BootstrapArgs& a, int N) { return a.mode == rtfhe::MODE_BLIND_ROTATE ? (size_t)2 * N : (size_t)a.n + 1; }

// `cnt` gates of a batch starting at gate `off` (plain batches advance the ciphertext pointers, netlist waves the index arrays)
inline BootstrapArgs batch_segment(const rtfhe_ctx* ctx, BootstrapArgs a, size_t off, size_t cnt, size_t out_words) {
    if (a.idx0) { a.ops += off; a.idx0 += off; a.idx1 += off; a.idx_out += off; }
    else { a.in0 += off * ((size_t)a.n + 1); a.in1 += off * ((size_t)a.n + 1); a.out += off * out_words; }
    if (a.ext) a.ext_first += (int32_t)off;     // one sample buffer per batch, tiled by batch-wide gate number (rtfhe::ext_slot)
    a.count = (int32_t)cnt;
    return a;
}

// The split path of a plain batch (whole rounds of the two-waves-per-gate kernels): blind rotation + sample extract of every gate
// (the bootstrap kernel in MODE_EXTRACT, launched by `blind_rotate`), then the key switch of the whole batch as one exact i8
// contraction on the matrix pipe (k_key_switch_mm) -- two launches back to back on the caller's stream, the lvl1 samples in between
// stay in HBM (4 MB per 1024 gates at N = 1024).
inline rtfhe_ctx::Tlwe1* tlwe1_of(rtfhe_ctx* ctx, hipStream_t s) {
    if (ctx->tlwe1_capture) return ctx->tlwe1_capture;
    auto it = ctx->tlwe1.find(s);
    return it == ctx->tlwe1.end() ? nullptr : &it->second;
}
inline bool split_ok(rtfhe_ctx* ctx, const BootstrapArgs& a, hipStream_t s) {
    if (!(a.mode == rtfhe::MODE_GATE && ctx->d_ksmat && ctx->ks_mm_min > 0 && (size_t)a.count >= (size_t)ctx->ks_mm_min)) return false;
    // A caller's own capture would bake THIS stream's scratch pointer into a graph the library does not own: a later, larger eager batch
    // on the stream frees and reallocates that buffer (ensure_tlwe1) and a replay then writes freed memory; a replay on another stream
    // would share the scratch with eager work on this one.  Only rtfhe_circuit_create's captures (which own their sample buffer) split.
    if (ctx->foreign_capture) return false;
    const rtfhe_ctx::Tlwe1* b = tlwe1_of(ctx, s);
    return b && (size_t)a.count <= b->cap;      // (the sample buffer is sized by ensure_tlwe1 before any launch or capture)
}
// blind_rotate(ctx, a', s) launches the bootstrap kernel(s) of the whole batch with a'.mode = MODE_EXTRACT: every gate's lvl1 sample goes to
// a'.ext in the key switch's operand order (rtfhe::ext_slot; segments of a batch advance ext_first, not the pointer) and its output row is zeroed for the key switch's atomics
template <typename F>
int launch_split(rtfhe_ctx* ctx, BootstrapArgs a, hipStream_t s, F blind_rotate) {
    uint32_t* samples = tlwe1_of(ctx, s)->d;
    a.mode = rtfhe::MODE_EXTRACT; a.ext = samples;
    if (int rc = blind_rotate(ctx, a, s)) return rc;
    return launch_key_switch_mm(ctx, a, samples, s);
}

}  // namespace rtfhe_host
