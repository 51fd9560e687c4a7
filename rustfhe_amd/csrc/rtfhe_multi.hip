// rtfhe_multi.hip -- one context over several GPUs of a node: independent gates are the unit of parallelism, so a batch is cut into
// contiguous ranges, one per device, and nothing is exchanged between the ranges.  Keys are replicated (device to device, once, at load time).
//
// Two forms of a batch:
//   * host buffers (rtfhe_gate_batch, rtfhe_mux_batch, ...): one host thread per device copies its range in, bootstraps it on the device's own
//     stream and copies it out -- the devices never talk to each other;
//   * a batch that LIVES ON THE PRIMARY DEVICE (rtfhe_gate_batch_dev / rtfhe_mux_batch_dev / rtfhe_bootstrap_batch_dev on a multi-device context):
//     the scatter / gather of `north_star` ("RCCL over xGMI only to scatter ciphertexts / gather results") behind the C ABI, so that a Rust
//     hom_nand_batch needs no Python and no process per GPU.  sharded_dev_batch below.
#include "rtfhe_host.hpp"

#include <thread>

using namespace rtfhe;
using namespace rtfhe_host;

namespace {

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

}  // namespace

namespace rtfhe_host {

// device 0's copy of a key -> a peer (device-to-device; xGMI between the GPUs of one node)
int replicate(rtfhe_ctx* ctx, rtfhe_ctx* peer, const void* src, void** dst_of_peer, size_t bytes) {
    if (!*dst_of_peer) {
        HIPCHECK(ctx, hipSetDevice(peer->device));
        HIPCHECK(ctx, hipMalloc(dst_of_peer, bytes));
    }
    HIPCHECK(ctx, hipMemcpyPeer(*dst_of_peer, peer->device, src, ctx->device, bytes));
    HIPCHECK(ctx, hipSetDevice(ctx->device));
    return 0;
}

// host-pointer batch: device d bootstraps the contiguous range [count d / D, count (d+1) / D)
int sharded_host_bootstrap(rtfhe_ctx* ctx, int op, int mode, int steps, const uint32_t* in0, const uint32_t* in1,
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

int sharded_host_mux(rtfhe_ctx* ctx, const uint32_t* c, const uint32_t* in0, const uint32_t* in1, uint32_t* out, size_t count) {
    if (ctx->peers.empty()) return mux_host_one(ctx, c, in0, in1, out, count);
    const int n_dev = 1 + (int)ctx->peers.size();
    const size_t w = (size_t)ctx->p.n + 1;
    return for_each_device(ctx, [&](rtfhe_ctx* cx, int d) {
        const size_t b = shard_begin(count, d, n_dev), e = shard_begin(count, d + 1, n_dev);
        return mux_host_one(cx, c + b * w, in0 + b * w, in1 + b * w, out + b * w, e - b);
    });
}

// A batch resident on the primary device, sharded over every device of the context; everything is enqueued from the calling thread and the call
// returns without synchronising (the semantics of every *_dev call).  In stream order of the caller's stream s on the primary:
//   1. an event marks "the inputs are ready";
//   2. the primary's OWN range is launched on s right behind it -- the primary never waits for anything before it computes;
//   3. every peer, on its own stream: waits for the event, pulls its range of the inputs into its staging buffers, bootstraps it, pushes the outputs
//      into the caller's output buffer on the primary, records its own event (and three more in between: rtfhe_ctx_peer_info reports the phases);
//   4. s waits for the peers' events: whatever the caller enqueues on s next sees all outputs (and may overwrite the inputs).
// The copies are hipMemcpyPeerAsync between device memory of the primary and of the peer -- EXPECTED to run on the copy engines (SDMA over xGMI)
// once peer access is in force, which rtfhe_ctx_create_multi enables explicitly and reports; UNVERIFIED: this path has only ever run with every
// entry naming one card (same-device copies).  The reason for wanting copy engines rather than copy kernels: the bootstrap kernels need every CU
// of a device at one workgroup per CU (__launch_bounds__(512, 1), LDS-full), and a copy KERNEL that still holds a CU when a bootstrap launch
// starts makes one workgroup wait a whole round for it -- the send side of an RCCL send/recv pair has exactly that hazard on the root.  Without
// peer access the runtime stages such a copy through host memory: still correct, and scatter_ms / gather_ms will show it.  Batch pointers that are
// not device memory of the primary (pinned host, managed, another GPU's memory: all admitted by gpu_accessible) are copied with
// hipMemcpyAsync(hipMemcpyDefault) instead, which lets the runtime find out where they live.
// Volumes are negligible next to the compute (config 3: 65,536 gates = 333 MB in, 167 MB out against 53 ms of bootstrapping per 8,192 gates).
// Inside a stream capture on s the whole batch stays on the primary (the staging buffers of a peer are not the capture's to bake in).
// A failure part-way leaves nothing dangling: s still waits for every peer already launched (they may be writing d_out), the current device is
// the primary's again, and the first error is what the call returns.
int sharded_dev_batch(rtfhe_ctx* ctx, int op, const void* d_c, const void* d_in0, const void* d_in1, void* d_out, size_t count, hipStream_t s) {
    if (int rc = use(ctx)) return rc;
    const bool mux = op < 0;
    auto run = [&](rtfhe_ctx* c, const void* cc, const void* i0, const void* i1, void* o, size_t cnt, hipStream_t st) {
        return mux ? mux_dev_one(c, cc, i0, i1, o, cnt, st) : launch_bootstrap(c, op, MODE_GATE, c->p.n, i0, i1, o, cnt, st);
    };
    if (count == 0) return 0;
    if (count > 0x7fffffff) return fail(ctx, RTFHE_ERR_INVALID, "count too large");
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusActive; }
    if (cs != hipStreamCaptureStatusNone) return run(ctx, d_c, d_in0, d_in1, d_out, count, s);
    const int n_dev = 1 + (int)ctx->peers.size();
    const size_t w = (size_t)ctx->p.n + 1;
    auto at = [&](const void* p, size_t gate) { return p ? (const void*)((const uint32_t*)p + gate * w) : nullptr; };
    // is p device memory of the primary?  (only then may hipMemcpyPeerAsync be told so)
    auto on_primary = [&](const void* p) {
        hipPointerAttribute_t a;
        if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
        return a.type == hipMemoryTypeDevice && a.device == ctx->device;
    };
    const bool peer_copies = on_primary(d_in0) && (!d_in1 || on_primary(d_in1)) && (!d_c || on_primary(d_c)) && on_primary(d_out);
    // staging of every peer first (it may allocate, i.e. synchronise a device), then nothing but asynchronous calls
    for (int d = 1; d < n_dev; d++) {
        rtfhe_ctx* peer = ctx->peers[d - 1];
        peer->shard_timed = false;
        const size_t bytes = (shard_begin(count, d + 1, n_dev) - shard_begin(count, d, n_dev)) * w * 4;
        if (!bytes) continue;
        int rc = use(peer);
        if (!rc) rc = ensure(peer, &peer->d_a, &peer->cap_a, bytes);
        if (!rc && (d_in1 || mux)) rc = ensure(peer, &peer->d_b, &peer->cap_b, bytes);
        if (!rc) rc = ensure(peer, &peer->d_c, &peer->cap_c, bytes);
        if (rc) { (void)hipSetDevice(ctx->device); return fail(ctx, rc, peer->err); }
    }
    if (int rc = use(ctx)) return rc;
    HIPCHECK(ctx, hipEventRecord(ctx->ev_shard, s));
    if (const size_t own = shard_begin(count, 1, n_dev))
        if (int rc = run(ctx, d_c, d_in0, d_in1, d_out, own, s)) return rc;
    // from here on an error may not return before s has been made to wait for the peers already launched
    int first_rc = 0;
    std::string first_err;
    std::vector<rtfhe_ctx*> launched;
    auto pull = [&](rtfhe_ctx* peer, void* dst, const void* src, size_t bytes) {
        return peer_copies ? hipMemcpyPeerAsync(dst, peer->device, src, ctx->device, bytes, peer->stream) : hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, peer->stream);
    };
    for (int d = 1; d < n_dev && !first_rc; d++) {
        rtfhe_ctx* peer = ctx->peers[d - 1];
        const size_t b = shard_begin(count, d, n_dev), cnt = shard_begin(count, d + 1, n_dev) - b, bytes = cnt * w * 4;
        if (!cnt) continue;
        auto hip = [&](hipError_t e, const char* what) {
            if (e == hipSuccess || first_rc) return e == hipSuccess;
            (void)hipGetLastError();
            first_rc = RTFHE_ERR_HIP;
            first_err = std::string(what) + " on device " + std::to_string(peer->device) + ": " + hipGetErrorString(e);
            return false;
        };
        if (!hip(hipSetDevice(peer->device), "hipSetDevice")) break;
        if (!hip(hipStreamWaitEvent(peer->stream, ctx->ev_shard, 0), "hipStreamWaitEvent")) break;
        bool ok = hip(hipEventRecord(peer->ev_sh[0], peer->stream), "hipEventRecord");
        // gate batch: in0 -> d_a, in1 -> d_b, out <- d_c.   MUX: c -> d_a, in1 -> d_b, in0 -> d_c, out <- d_a (as mux_host_one)
        ok = ok && hip(pull(peer, peer->d_a, at(mux ? d_c : d_in0, b), bytes), "copy of the first operand to the peer");
        if (ok && d_in1) ok = hip(pull(peer, peer->d_b, at(d_in1, b), bytes), "copy of the second operand to the peer");
        if (ok && mux) ok = hip(pull(peer, peer->d_c, at(d_in0, b), bytes), "copy of the third operand to the peer");
        ok = ok && hip(hipEventRecord(peer->ev_sh[1], peer->stream), "hipEventRecord");
        if (ok) {
            const int rc = mux ? run(peer, peer->d_a, peer->d_c, peer->d_b, peer->d_a, cnt, peer->stream)
                               : run(peer, nullptr, peer->d_a, d_in1 ? peer->d_b : nullptr, peer->d_c, cnt, peer->stream);
            if (rc) { first_rc = rc; first_err = "device " + std::to_string(peer->device) + ": " + peer->err; ok = false; }
        }
        ok = ok && hip(hipEventRecord(peer->ev_sh[2], peer->stream), "hipEventRecord");
        if (ok) {
            void* dst = (uint32_t*)d_out + b * w;
            const void* src = mux ? peer->d_a : peer->d_c;
            ok = hip(peer_copies ? hipMemcpyPeerAsync(dst, ctx->device, src, peer->device, bytes, peer->stream) : hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, peer->stream),
                     "copy of the outputs to the primary");
        }
        // whatever was enqueued on the peer's stream so far is fenced by this event, complete or not
        if (hipEventRecord(peer->ev_shard, peer->stream) == hipSuccess) { launched.push_back(peer); peer->shard_timed = ok; }
        else (void)hipGetLastError();
    }
    (void)hipSetDevice(ctx->device);
    for (rtfhe_ctx* peer : launched)
        if (hipStreamWaitEvent(s, peer->ev_shard, 0) != hipSuccess) {
            (void)hipGetLastError();
            // last resort: the caller must not see d_out before the peer is done with it
            (void)hipSetDevice(peer->device); (void)hipStreamSynchronize(peer->stream); (void)hipSetDevice(ctx->device);
        }
    if (first_rc) return fail(ctx, first_rc, first_err);
    return 0;
}

}  // namespace rtfhe_host

extern "C" {

// the contiguous gate range entry d of an n_dev-device context takes of a batch of `count` gates (pure arithmetic: what the sharded calls
// use, exported so that a caller can lay out per-device buffers; no GPU needed)
int rtfhe_shard_range(size_t count, int d, int n_dev, size_t* begin, size_t* end) {
    if (n_dev < 1 || n_dev > 64 || d < 0 || d >= n_dev || !begin || !end) return fail(nullptr, RTFHE_ERR_INVALID, "rtfhe_shard_range: bad argument");
    *begin = shard_begin(count, d, n_dev);
    *end = shard_begin(count, d + 1, n_dev);
    return 0;
}

}  // extern "C"
