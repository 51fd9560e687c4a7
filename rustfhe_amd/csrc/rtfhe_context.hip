// rtfhe_context.hip -- contexts: creation and teardown, error plumbing, device buffers and copies, twiddle-table calls, keys (loading, the
// second layouts built on demand, the device footprint).
#include "rtfhe_host.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>

using namespace rtfhe;
using namespace rtfhe_host;

namespace {

thread_local std::string g_last_error;

// The dynamic-LDS limit of a kernel is a per-function, per-DEVICE attribute shared by every context of the process: it is only
// ever raised (a second context with a smaller mask would otherwise lower it under the first one's launches) and remembered
// process-wide, so that a launch costs no runtime call beyond the launch itself.
std::mutex g_lds_mutex;
std::map<std::pair<int, const void*>, size_t> g_lds_granted;     // (device, kernel) -> largest dynamic LDS granted so far

int ensure_pinned(rtfhe_ctx* ctx, int slot, size_t bytes) {
    if (ctx->cap_pin[slot] >= bytes && ctx->h_pin[slot]) return 0;
    if (ctx->h_pin[slot]) HIPCHECK(ctx, hipHostFree(ctx->h_pin[slot]));
    ctx->h_pin[slot] = nullptr; ctx->cap_pin[slot] = 0;
    HIPCHECK(ctx, hipHostMalloc(&ctx->h_pin[slot], bytes ? bytes : 16, hipHostMallocDefault));
    ctx->cap_pin[slot] = bytes;
    return 0;
}

// TRGSWRepF::from (trgsw.rs:68-76): ifft_torus = forward transform of the key words viewed as signed i32, from the
// device copy of the torus-form key into the device spectra
int transform_bk_from_torus(rtfhe_ctx* ctx) {
    const size_t words = bk_word_count(ctx->p);
    FftArgs a{ctx->d_tw, ctx->d_bk_torus, ctx->d_bk, (int32_t)(words / ctx->p.N), 1, 2 * ctx->p.l, 0};
    if (int rc = launch_fft(ctx, true, a, ctx->stream)) return rc;
    HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

// a new key (or new tables under a torus-form key): whatever was derived from the old spectra is stale -- and, where its buffer exists, is
// rebuilt at once by the caller (rebuild_derived_keys, rtfhe_dispatch_fft.hip)
void key_changed(rtfhe_ctx* ctx) {
    ctx->ebk_valid = false; ctx->p4bk_valid = false; ctx->ntt_ready = false; ctx->xfft_ready = false;
}

}  // namespace

namespace rtfhe_host {

int fail(rtfhe_ctx* ctx, int code, const std::string& msg) {
    g_last_error = msg;
    if (ctx) ctx->err = msg;
    return code;
}
const std::string& last_error_of_thread() { return g_last_error; }

int ensure(rtfhe_ctx* ctx, void** ptr, size_t* cap, size_t bytes) {
    if (*cap >= bytes && *ptr) return 0;
    if (*ptr) HIPCHECK(ctx, hipFree(*ptr));
    *ptr = nullptr; *cap = 0;
    HIPCHECK(ctx, hipMalloc(ptr, bytes ? bytes : 16));
    *cap = bytes;
    return 0;
}

int allow_lds_raw(rtfhe_ctx* ctx, const void* key, size_t bytes) {
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

int use(rtfhe_ctx* ctx) {
    if (!ctx) return fail(nullptr, RTFHE_ERR_INVALID, "null context");
    HIPCHECK(ctx, hipSetDevice(ctx->device));
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

}  // namespace rtfhe_host

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
        if (const char* e = std::getenv("RTFHE_PAIR_RR")) ctx->rr = std::atoi(e);
    }
    if (!rc) rc = prime_fft_kernels(ctx);
    if (!rc) rc = prime_ntt_kernels(ctx);
    if (!rc) rc = prime_xfft_kernels(ctx);
    if (!rc) rc = upload_twiddles(ctx);
    if (!rc && (hipMalloc((void**)&ctx->d_fault, 4) != hipSuccess || hipMemset(ctx->d_fault, 0, 4) != hipSuccess))
        rc = fail(ctx, RTFHE_ERR_HIP, "hipMalloc failed");
    if (!rc && hipStreamCreate(&ctx->stream) != hipSuccess) rc = fail(ctx, RTFHE_ERR_HIP, "hipStreamCreate failed");
    if (!rc && (hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess || hipEventCreate(&ctx->ev_shard) != hipSuccess ||
                hipEventCreate(&ctx->ev_sh[0]) != hipSuccess || hipEventCreate(&ctx->ev_sh[1]) != hipSuccess || hipEventCreate(&ctx->ev_sh[2]) != hipSuccess))
        rc = fail(ctx, RTFHE_ERR_HIP, "hipEventCreate failed");
    if (rc) { g_last_error = ctx->err; rtfhe_ctx_destroy(ctx); return rc; }
#ifdef RTFHE_WG_STAMPS
    // 128 words of phase sums of workgroup 0 + 4 words per workgroup (first 1024): loop start, loop end (s_memtime), hardware id
    if (hipMalloc((void**)&ctx->d_dbg, (128 + 4096) * 8) == hipSuccess) (void)hipMemset(ctx->d_dbg, 0, (128 + 4096) * 8);
#endif
    *out = ctx;
    return 0;
}

int rtfhe_ctx_create(const rtfhe_params* p, int device_id, rtfhe_ctx** out) { return create_single(p, device_id, out); }

// One context over several GPUs of the node: device_ids[0] is the primary.  Keys loaded into the context are
// transformed once on the primary and copied device-to-device to the others; batch calls shard contiguous gate ranges over the
// devices (rtfhe_multi.hip).  A device may be named more than once: every entry is a full context of its own (stream, staging, key replica).
int rtfhe_ctx_create_multi(const rtfhe_params* p, const int* device_ids, int n_dev, rtfhe_ctx** out) {
    if (!p || !out || !device_ids) return fail(nullptr, RTFHE_ERR_INVALID, "null argument");
    *out = nullptr;
    if (n_dev < 1 || n_dev > 64) return fail(nullptr, RTFHE_ERR_INVALID, "n_dev out of range");
    rtfhe_ctx* ctx = nullptr;
    if (int rc = create_single(p, device_ids[0], &ctx)) return rc;
    for (int d = 1; d < n_dev; d++) {
        rtfhe_ctx* peer = nullptr;
        if (int rc = create_single(p, device_ids[d], &peer)) { rtfhe_ctx_destroy(ctx); return rc; }
        ctx->peers.push_back(peer);
        // First contact between the two devices happens HERE, explicitly, and what the runtime answers is kept (rtfhe_ctx_peer_info): whether each
        // may address the other's memory, whether enabling that worked, and what link the runtime reports between them.  A copy between devices
        // without peer access is staged through host memory by the runtime -- correct, and several times slower than the xGMI path the design
        // expects; the context works either way and says which it got.
        rtfhe_ctx::PeerLink& k = peer->link;
        if (peer->device == ctx->device) { k.same_device = 1; continue; }
        auto enable = [&](int from, int to) {
            if (hipSetDevice(from) != hipSuccess) { (void)hipGetLastError(); return 0; }
            const hipError_t e = hipDeviceEnablePeerAccess(to, 0);
            (void)hipGetLastError();
            return (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) ? 1 : 0;
        };
        if (hipDeviceCanAccessPeer(&k.can_from, ctx->device, peer->device) != hipSuccess) { (void)hipGetLastError(); k.can_from = 0; }
        if (hipDeviceCanAccessPeer(&k.can_to, peer->device, ctx->device) != hipSuccess) { (void)hipGetLastError(); k.can_to = 0; }
        if (k.can_from) k.en_from = enable(ctx->device, peer->device);
        if (k.can_to) k.en_to = enable(peer->device, ctx->device);
        uint32_t lt = 0, hops = 0;
        if (hipExtGetLinkTypeAndHopCount(ctx->device, peer->device, &lt, &hops) == hipSuccess) { k.link_type = lt; k.hops = hops; }
        else (void)hipGetLastError();
    }
    (void)hipSetDevice(ctx->device);
    *out = ctx;
    return 0;
}

int rtfhe_device_link(int dev_a, int dev_b, int32_t* can_access, uint32_t* link_type, uint32_t* hops) {
    if (!can_access || !link_type || !hops) return fail(nullptr, RTFHE_ERR_INVALID, "null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)hipGetLastError(); return fail(nullptr, RTFHE_ERR_NO_DEVICE, "no HIP device available"); }
    if (dev_a < 0 || dev_a >= ndev || dev_b < 0 || dev_b >= ndev || dev_a == dev_b) return fail(nullptr, RTFHE_ERR_INVALID, "rtfhe_device_link: two distinct device ids of this node");
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, dev_a, dev_b) != hipSuccess) { (void)hipGetLastError(); can = 0; }
    *can_access = can; *link_type = 0xffffffffu; *hops = 0;
    uint32_t lt = 0, h = 0;
    if (hipExtGetLinkTypeAndHopCount(dev_a, dev_b, &lt, &h) == hipSuccess) { *link_type = lt; *hops = h; }
    else (void)hipGetLastError();
    return 0;
}

int rtfhe_ctx_peer_info(rtfhe_ctx* ctx, int d, rtfhe_peer_info* out) {
    if (!ctx || !out || d < 1 || d > (int)ctx->peers.size()) return fail(ctx, RTFHE_ERR_INVALID, "rtfhe_ctx_peer_info: entry out of range (1 <= d < device count)");
    rtfhe_ctx* peer = ctx->peers[d - 1];
    const rtfhe_ctx::PeerLink& k = peer->link;
    *out = rtfhe_peer_info{peer->device, k.same_device, k.can_from, k.can_to, k.en_from, k.en_to, k.link_type, k.hops, -1.f, -1.f, -1.f};
    if (peer->shard_timed && hipEventQuery(peer->ev_shard) == hipSuccess) {
        float a = -1.f, b = -1.f, c = -1.f;
        if (hipEventElapsedTime(&a, peer->ev_sh[0], peer->ev_sh[1]) == hipSuccess && hipEventElapsedTime(&b, peer->ev_sh[1], peer->ev_sh[2]) == hipSuccess &&
            hipEventElapsedTime(&c, peer->ev_sh[2], peer->ev_shard) == hipSuccess) { out->scatter_ms = a; out->compute_ms = b; out->gather_ms = c; }
    }
    (void)hipGetLastError();
    return 0;
}

int rtfhe_ctx_device_count(const rtfhe_ctx* ctx) { return ctx ? 1 + (int)ctx->peers.size() : 0; }

// device memory entry d of the context holds right now: the keys in every form built so far, staging and scratch buffers (not the twiddle tables,
// a few hundred KiB, nor the sample buffers of live circuits)
int rtfhe_ctx_memory_bytes(const rtfhe_ctx* ctx, int d, size_t* bytes) {
    if (!ctx || !bytes || d < 0 || d > (int)ctx->peers.size()) return fail(nullptr, RTFHE_ERR_INVALID, "rtfhe_ctx_memory_bytes: bad argument");
    const rtfhe_ctx* c = d == 0 ? ctx : ctx->peers[d - 1];
    const size_t spectra = bk_cplx_count(c->p) * sizeof(cplx);
    size_t b = 0;
    if (c->d_bk) b += spectra;
    if (c->d_ebk) b += spectra;
    if (c->d_p4bk) b += spectra;
    if (c->d_bk_torus) b += bk_word_count(c->p) * 4;
    if (c->d_ntt_bk) b += bk_word_count(c->p) * sizeof(double);
    if (c->d_xbk) b += 2 * spectra;
    if (c->d_ksk) b += c->ksk_bytes;
    if (c->d_ksmat) b += c->ksmat_bytes;
    for (const auto& kv : c->tlwe1) b += kv.second.cap * ((size_t)c->p.N + 1) * 4;
    b += c->cap_a + c->cap_b + c->cap_c;
    for (const auto& kv : c->mux) b += 2 * kv.second.cap;
    *bytes = b;
    return 0;
}

// pinned host memory for ciphertext buffers: host-pointer calls DMA straight from / into it (no staging copy)
void* rtfhe_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
void rtfhe_host_free(void* p) { if (p) (void)hipHostFree(p); }
#ifdef RTFHE_WG_STAMPS
extern "C" int rtfhe_debug_read_stamps(rtfhe_ctx* ctx, unsigned long long* out128) {
    if (!ctx || !ctx->d_dbg) return RTFHE_ERR_STATE;
    return hipMemcpy(out128, ctx->d_dbg, 128 * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : RTFHE_ERR_HIP;
}
extern "C" int rtfhe_debug_read_wg_times(rtfhe_ctx* ctx, unsigned long long* out4096) {
    if (!ctx || !ctx->d_dbg) return RTFHE_ERR_STATE;
    return hipMemcpy(out4096, ctx->d_dbg + 128, 4096 * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : RTFHE_ERR_HIP;
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
    if (ctx->d_etw) (void)hipFree(ctx->d_etw);
    if (ctx->d_ebk) (void)hipFree(ctx->d_ebk);
    if (ctx->d_p4bk) (void)hipFree(ctx->d_p4bk);
    if (ctx->d_bk_torus) (void)hipFree(ctx->d_bk_torus);
    if (ctx->d_ntt_bk) (void)hipFree(ctx->d_ntt_bk);
    if (ctx->d_ntt_tw) (void)hipFree(ctx->d_ntt_tw);
    if (ctx->d_xbk) (void)hipFree(ctx->d_xbk);
    if (ctx->d_xtw) (void)hipFree(ctx->d_xtw);
    if (ctx->d_ksk) (void)hipFree(ctx->d_ksk);
    if (ctx->d_ksmat) (void)hipFree(ctx->d_ksmat);
    for (auto& kv : ctx->tlwe1) if (kv.second.d) (void)hipFree(kv.second.d);
    if (ctx->d_a) (void)hipFree(ctx->d_a);
    if (ctx->d_b) (void)hipFree(ctx->d_b);
    if (ctx->d_c) (void)hipFree(ctx->d_c);
    for (void* h : ctx->h_pin) if (h) (void)hipHostFree(h);
    for (auto& kv : ctx->mux) for (void* m : kv.second.m) if (m) (void)hipFree(m);
    for (void* m : ctx->mux_retired) (void)hipFree(m);
    for (hipEvent_t e : ctx->ks_events) (void)hipEventDestroy(e);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->ev_shard) (void)hipEventDestroy(ctx->ev_shard);
    for (hipEvent_t e : ctx->ev_sh) if (e) (void)hipEventDestroy(e);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int rtfhe_set_backend(rtfhe_ctx* ctx, int backend) {
    if (int rc = use(ctx)) return rc;
    if (backend != RTFHE_BACKEND_FFT64_MIRROR && backend != RTFHE_BACKEND_NTT_EXACT && backend != RTFHE_BACKEND_FFT_SPLIT_EXACT)
        return fail(ctx, RTFHE_ERR_INVALID, "unknown backend");
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
    if (ctx->has_bk && ctx->d_bk_torus) {
        if (int rc = transform_bk_from_torus(ctx)) return rc;
        key_changed(ctx);
        if (int rc = rebuild_derived_keys(ctx)) return rc;
    }
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
    if (int rc = transform_bk_from_torus(ctx)) return rc;
    key_changed(ctx);
    ctx->has_bk = true;
    if (int rc = rebuild_derived_keys(ctx)) return rc;
    for (rtfhe_ctx* peer : ctx->peers) {      // the transformed key and its torus form, device to device
        if (int rc = replicate(ctx, peer, ctx->d_bk, (void**)&peer->d_bk, bk_cplx_count(ctx->p) * sizeof(cplx))) return rc;
        if (int rc = replicate(ctx, peer, ctx->d_bk_torus, (void**)&peer->d_bk_torus, words * 4)) return rc;
        key_changed(peer); peer->has_bk = true;
        if (int rc = rebuild_derived_keys(peer)) return fail(ctx, rc, peer->err);
    }
    HIPCHECK(ctx, hipSetDevice(ctx->device));
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
    if (int rc = launch_bk_permute(ctx, (const double*)ctx->d_a, (double*)ctx->d_bk, polys, 0, ctx->stream)) return rc;
    HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_bk_torus) { (void)hipFree(ctx->d_bk_torus); ctx->d_bk_torus = nullptr; }   // no torus form of this key
    key_changed(ctx);
    ctx->has_bk = true;
    if (int rc = rebuild_derived_keys(ctx)) return rc;
    for (rtfhe_ctx* peer : ctx->peers) {
        if (int rc = replicate(ctx, peer, ctx->d_bk, (void**)&peer->d_bk, bk_cplx_count(ctx->p) * sizeof(cplx))) return rc;
        if (peer->d_bk_torus) { (void)hipSetDevice(peer->device); (void)hipFree(peer->d_bk_torus); peer->d_bk_torus = nullptr; (void)hipSetDevice(ctx->device); }
        key_changed(peer); peer->has_bk = true;
        if (int rc = rebuild_derived_keys(peer)) return fail(ctx, rc, peer->err);
    }
    HIPCHECK(ctx, hipSetDevice(ctx->device));
    return 0;
}

int rtfhe_export_bk_fft(rtfhe_ctx* ctx, double* bk_f) {
    if (int rc = use(ctx)) return rc;
    if (!bk_f) return fail(ctx, RTFHE_ERR_INVALID, "null argument");
    if (!ctx->has_bk) return fail(ctx, RTFHE_ERR_STATE, "bootstrapping key not loaded");
    const size_t words = bk_word_count(ctx->p);
    if (int rc = ensure(ctx, &ctx->d_a, &ctx->cap_a, words * 8)) return rc;
    const size_t polys = words / ctx->p.N;
    if (int rc = launch_bk_permute(ctx, (const double*)ctx->d_bk, (double*)ctx->d_a, polys, 1, ctx->stream)) return rc;
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
    ctx->ksk_bytes = (dev_rows + 1) * ksw * 4;
    if (!ctx->d_ksk) HIPCHECK(ctx, hipMalloc((void**)&ctx->d_ksk, ctx->ksk_bytes));
    if (int rc = launch_ksk_combine(ctx, (const uint32_t*)ctx->d_a, ctx->stream)) return rc;
    HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
    // the same key as signed byte limbs in i8-MFMA operand order, for the batch key switch of the split path
    size_t ksmat_bytes = 0;
    if (ctx->ks_mm_min > 0) {
        const int colgroups = (ctx->p.n + 1 + 15) / 16;
        ksmat_bytes = (size_t)colgroups * (ctx->p.N / 2) * 4 * 64 * sizeof(uint4);
        if (!ctx->d_ksmat) HIPCHECK(ctx, hipMalloc((void**)&ctx->d_ksmat, ksmat_bytes));
        ctx->ksmat_bytes = ksmat_bytes;
        if (int rc = launch_ksmat_build(ctx, (const uint32_t*)ctx->d_a, colgroups, ctx->stream)) return rc;
        HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
        if (int rc = ensure_tlwe1(ctx, ctx->tlwe1[ctx->stream], 8192)) return rc;     // the context's own stream (host-pointer calls)
    }
    ctx->has_ksk = true;
    for (rtfhe_ctx* peer : ctx->peers) {
        if (int rc = replicate(ctx, peer, ctx->d_ksk, (void**)&peer->d_ksk, ctx->ksk_bytes)) return rc;
        peer->ksk_bytes = ctx->ksk_bytes;
        if (ksmat_bytes) {
            if (int rc = replicate(ctx, peer, ctx->d_ksmat, (void**)&peer->d_ksmat, ksmat_bytes)) return rc;
            peer->ksmat_bytes = ksmat_bytes;
        }
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

}  // extern "C"
