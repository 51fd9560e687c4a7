// rtfhe_circuit.hip -- levelised netlists (BASELINE config 4): one dependency wave per call, or every wave of a netlist recorded once into a
// HIP graph and replayed as one submission.  Replaces eval_logic_expr over impl Logip for TFHE (nander/src/lib.rs:40-89).
#include "rtfhe_host.hpp"

#include <algorithm>

using namespace rtfhe;
using namespace rtfhe_host;

namespace rtfhe_host {

// releases the graph objects of a circuit (the handle itself stays valid for rtfhe_circuit_destroy)
void circuit_release(rtfhe_circuit* c) {
    (void)hipSetDevice(c->device);
    if (c->exec) (void)hipGraphExecDestroy(c->exec);
    if (c->graph) (void)hipGraphDestroy(c->graph);
    if (c->d_samples) (void)hipFree(c->d_samples);
    c->exec = nullptr; c->graph = nullptr; c->d_samples = nullptr;
}

}  // namespace rtfhe_host

extern "C" {

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
    if (int rc = backend_prepare(ctx)) return rc;          // nothing but kernel launches may happen inside the capture
    for (int32_t w = 0; w < num_waves; w++)                // ... so the key layouts the waves' dispatches read are built now
        if (int rc = ensure_bk_layouts(ctx, (size_t)(wave_offsets[w + 1] - wave_offsets[w]), MODE_GATE)) return rc;
    rtfhe_ctx::Tlwe1 cbuf;                                 // ... so the circuit's own sample buffer (split path) is allocated now
    if (ctx->ks_mm_min > 0 && ctx->d_ksmat) {
        size_t widest = 0;
        for (int32_t w = 0; w < num_waves; w++) widest = std::max(widest, (size_t)(wave_offsets[w + 1] - wave_offsets[w]));
        if (int rc = ensure_tlwe1(ctx, cbuf, widest)) return rc;
    }
    rtfhe_circuit* c = new (std::nothrow) rtfhe_circuit();
    if (!c) { if (cbuf.d) (void)hipFree(cbuf.d); return fail(ctx, RTFHE_ERR_NOMEM, "out of host memory"); }
    c->ctx = ctx; c->device = ctx->device; c->waves = num_waves; c->d_samples = cbuf.d; c->backend = ctx->backend;
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
    if (c->stale) return fail(ctx, RTFHE_ERR_STATE, "the circuit was recorded on an exact backend and the bootstrapping key has since been replaced by one without a "
                                                  "torus form (rtfhe_load_bk_fft): its key form could not follow; record the circuit again");
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

}  // extern "C"
