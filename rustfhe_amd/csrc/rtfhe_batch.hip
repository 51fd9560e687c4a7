// rtfhe_batch.hip -- batches of gates on one device: the backend switch, the allocation rules around stream captures, device-pointer and
// host-pointer batches, MUX, synchronisation and the timers bench.py reads.
#include "rtfhe_host.hpp"

using namespace rtfhe;
using namespace rtfhe_host;

namespace rtfhe_host {

int backend_prepare(rtfhe_ctx* ctx) {
    if (ctx->backend == RTFHE_BACKEND_NTT_EXACT) return ntt_prepare(ctx);
    if (ctx->backend == RTFHE_BACKEND_FFT_SPLIT_EXACT) return xfft_prepare(ctx);
    return 0;
}

// lvl1 sample buffer of the split path for stream s: sized outside launches (hipMalloc is not allowed inside a stream capture)
int ensure_tlwe1(rtfhe_ctx* ctx, rtfhe_ctx::Tlwe1& b, size_t gates) {
    if (b.cap >= gates) return 0;
    HIPCHECK(ctx, hipDeviceSynchronize());            // earlier launches may still read the old buffer
    if (b.d) HIPCHECK(ctx, hipFree(b.d));
    b.d = nullptr; b.cap = 0;
    const size_t cap = ((gates < 1024 ? 1024 : gates) + 15) / 16 * 16;      // whole tiles of 16 gates (rtfhe::ext_slot): 16 N + 16 words each
    HIPCHECK(ctx, hipMalloc((void**)&b.d, cap * ((size_t)ctx->p.N + 1) * 4));
    b.cap = cap;
    return 0;
}

int launch_bootstrap(rtfhe_ctx* ctx, int op, int mode, int steps, const void* d_in0, const void* d_in1, void* d_out,
                     size_t count, hipStream_t s, const int32_t* d_ops, const int32_t* d_idx0, const int32_t* d_idx1, const int32_t* d_idx_out, int32_t num_wires) {
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
    a.ext = nullptr;
    struct Unset { bool& f; ~Unset() { f = false; } } unset{ctx->foreign_capture};
    // Whatever allocates happens here, outside any stream capture: the split path's sample buffer of this stream is created / grows, and the
    // second key layout the dispatch of this batch reads is built on first use.  Inside a capture that is not rtfhe_circuit_create's own
    // (which prepared both before it began) the batch stays on the fused kernel and on the kernels that read the canonical key layout.
    if (!ctx->tlwe1_capture) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusActive; }
        if (cs != hipStreamCaptureStatusNone) {
            ctx->foreign_capture = true;
        } else {
            if (int rc = ensure_bk_layouts(ctx, count, mode)) return rc;
            if (mode == MODE_GATE && ctx->ks_mm_min > 0 && ctx->d_ksmat) {
                const rtfhe_ctx::Tlwe1* have = tlwe1_of(ctx, s);
                if (!have || count > have->cap)
                    if (int rc = ensure_tlwe1(ctx, ctx->tlwe1[s], count)) return rc;
            }
        }
    }
    if (ctx->backend == RTFHE_BACKEND_NTT_EXACT) {
        if (int rc = ntt_prepare(ctx)) return rc;
        return launch_bootstrap_ntt(ctx, a, s);
    }
    if (ctx->backend == RTFHE_BACKEND_FFT_SPLIT_EXACT) {
        if (int rc = xfft_prepare(ctx)) return rc;
        return launch_bootstrap_xfft(ctx, a, s);
    }
    return launch_bootstrap_fft(ctx, a, s);
}

int run_host_bootstrap_one(rtfhe_ctx* ctx, int op, int mode, int steps, const uint32_t* in0, const uint32_t* in1,
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

// hom_mux (tfhe.rs:27-40): i1 = AND(c, in1); i0 = AND(-c, in0); bootstrap(i1 + i0 + 1/8) -- the last line is hom_or(i1, i0).
// Three launches back to back on stream s with i1 / i0 kept in the context's own device buffers OF THAT STREAM (advisor r5: one pair shared by
// every stream let two overlapping MUX batches overwrite each other's intermediates).  Inside a caller's stream capture nothing may be
// allocated or synchronised: the call goes through only if this stream's pair already holds the batch (run one eager MUX batch of at least this
// size on the stream first) -- and that pair is then kept for as long as the context lives, because the graph owns its addresses.
int mux_dev_one(rtfhe_ctx* ctx, const void* d_c, const void* d_in0, const void* d_in1, void* d_out, size_t count, hipStream_t s) {
    if (int rc = use(ctx)) return rc;
    if (count == 0) return 0;
    const size_t bytes = count * ((size_t)ctx->p.n + 1) * 4;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusActive; }
    const bool capturing = cs != hipStreamCaptureStatusNone;
    auto it = ctx->mux.find(s);
    if (capturing) {
        if (it == ctx->mux.end() || it->second.cap < bytes)
            return fail(ctx, RTFHE_ERR_STATE, "a MUX batch inside a stream capture needs this stream's intermediate buffers to exist already: run one eager MUX batch of "
                                              "at least this many gates on the stream before capturing");
        it->second.captured = true;
    } else if (it == ctx->mux.end() || it->second.cap < bytes) {
        rtfhe_ctx::MuxBuf& mb = ctx->mux[s];
        HIPCHECK(ctx, hipDeviceSynchronize());            // earlier MUX batches of this stream may still read the old intermediates
        for (void*& m : mb.m) {
            if (m && mb.captured) ctx->mux_retired.push_back(m);      // a graph holds its address: kept until the context goes
            else if (m) HIPCHECK(ctx, hipFree(m));
            m = nullptr;
        }
        mb.cap = 0; mb.captured = false;
        for (void*& m : mb.m) HIPCHECK(ctx, hipMalloc(&m, bytes));
        mb.cap = bytes;
        it = ctx->mux.find(s);
    }
    void* const i1 = it->second.m[0];
    void* const i0 = it->second.m[1];
    const int n = ctx->p.n;
    if (int rc = launch_bootstrap(ctx, RTFHE_AND, MODE_GATE, n, d_c, d_in1, i1, count, s)) return rc;
    if (int rc = launch_bootstrap(ctx, RTFHE_ANDNY, MODE_GATE, n, d_c, d_in0, i0, count, s)) return rc;
    return launch_bootstrap(ctx, RTFHE_OR, MODE_GATE, n, i1, i0, d_out, count, s);
}

// ... with host buffers: one copy in (c, in0, in1), the three launches on the context's stream, one copy out
int mux_host_one(rtfhe_ctx* ctx, const uint32_t* c, const uint32_t* in0, const uint32_t* in1, uint32_t* out, size_t count) {
    if (int rc = use(ctx)) return rc;
    if (count == 0) return 0;
    const size_t bytes = count * ((size_t)ctx->p.n + 1) * 4;
    if (int rc = ensure(ctx, &ctx->d_a, &ctx->cap_a, bytes)) return rc;
    if (int rc = ensure(ctx, &ctx->d_b, &ctx->cap_b, bytes)) return rc;
    if (int rc = ensure(ctx, &ctx->d_c, &ctx->cap_c, bytes)) return rc;
    if (int rc = copy_in(ctx, ctx->d_a, c, bytes, 0)) return rc;
    if (int rc = copy_in(ctx, ctx->d_b, in1, bytes, 1)) return rc;
    if (int rc = copy_in(ctx, ctx->d_c, in0, bytes, 2)) return rc;
    if (int rc = mux_dev_one(ctx, ctx->d_a, ctx->d_c, ctx->d_b, ctx->d_a, count, ctx->stream)) return rc;      // (the OR writes d_a after both ANDs have read it: one stream)
    return copy_out(ctx, out, ctx->d_a, bytes, 0);
}

}  // namespace rtfhe_host

extern "C" {

// Device-pointer batches enqueue on the caller's stream and return without synchronising.  On a multi-device context the batch lives on the
// primary device and is sharded over all devices (rtfhe_multi.hip: sharded_dev_batch).
int rtfhe_gate_batch_dev(rtfhe_ctx* ctx, int op, const void* d_in0, const void* d_in1, void* d_out, size_t count, void* stream) {
    if (int rc = use(ctx)) return rc;
    if (op < RTFHE_NAND || op > RTFHE_ANDNY) return fail(ctx, RTFHE_ERR_INVALID, "unknown gate");
    if (!d_in0 || !d_out) return fail(ctx, RTFHE_ERR_INVALID, "null argument");
    if (!gpu_accessible(ctx, d_in0) || (d_in1 && !gpu_accessible(ctx, d_in1)) || !gpu_accessible(ctx, d_out))
        return fail(ctx, RTFHE_ERR_INVALID, "rtfhe_gate_batch_dev needs device pointers (got memory the GPU cannot address)");
    if (!ctx->peers.empty()) return sharded_dev_batch(ctx, op, nullptr, d_in0, d_in1, d_out, count, (hipStream_t)stream);
    return launch_bootstrap(ctx, op, MODE_GATE, ctx->p.n, d_in0, d_in1, d_out, count, (hipStream_t)stream);
}

int rtfhe_bootstrap_batch_dev(rtfhe_ctx* ctx, const void* d_tlwe, void* d_out, size_t count, void* stream) {
    return rtfhe_gate_batch_dev(ctx, RTFHE_COPY, d_tlwe, nullptr, d_out, count, stream);
}

int rtfhe_mux_batch_dev(rtfhe_ctx* ctx, const void* d_c, const void* d_in0, const void* d_in1, void* d_out, size_t count, void* stream) {
    if (int rc = use(ctx)) return rc;
    if (!d_c || !d_in0 || !d_in1 || !d_out) return fail(ctx, RTFHE_ERR_INVALID, "null argument");
    if (!gpu_accessible(ctx, d_c) || !gpu_accessible(ctx, d_in0) || !gpu_accessible(ctx, d_in1) || !gpu_accessible(ctx, d_out))
        return fail(ctx, RTFHE_ERR_INVALID, "rtfhe_mux_batch_dev needs device pointers (got memory the GPU cannot address)");
    if (!ctx->has_bk) return fail(ctx, RTFHE_ERR_STATE, "bootstrapping key not loaded");
    if (!ctx->has_ksk) return fail(ctx, RTFHE_ERR_STATE, "key-switching key not loaded");
    if (count > 0x7fffffff) return fail(ctx, RTFHE_ERR_INVALID, "count too large");
    if (!ctx->peers.empty()) return sharded_dev_batch(ctx, -1, d_c, d_in0, d_in1, d_out, count, (hipStream_t)stream);
    return mux_dev_one(ctx, d_c, d_in0, d_in1, d_out, count, (hipStream_t)stream);
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

int rtfhe_gate_batch(rtfhe_ctx* ctx, int op, const uint32_t* in0, const uint32_t* in1, uint32_t* out, size_t count) {
    if (!ctx) return fail(nullptr, RTFHE_ERR_INVALID, "null context");
    if (op < RTFHE_NAND || op > RTFHE_ANDNY) return fail(ctx, RTFHE_ERR_INVALID, "unknown gate");
    const bool unary = (op == RTFHE_NOT || op == RTFHE_COPY);
    if (!unary && !in1) return fail(ctx, RTFHE_ERR_INVALID, "binary gate needs two inputs");
    return sharded_host_bootstrap(ctx, op, MODE_GATE, ctx->p.n, in0, unary ? nullptr : in1, out, count, (size_t)ctx->p.n + 1);
}

int rtfhe_bootstrap_batch(rtfhe_ctx* ctx, const uint32_t* tlwe, uint32_t* out, size_t count) {
    return rtfhe_gate_batch(ctx, RTFHE_COPY, tlwe, nullptr, out, count);
}

int rtfhe_blind_rotate_batch(rtfhe_ctx* ctx, const uint32_t* tlwe, int32_t steps, uint32_t* acc, size_t count) {
    if (!ctx) return fail(nullptr, RTFHE_ERR_INVALID, "null context");
    if (steps < 0 || steps > ctx->p.n) return fail(ctx, RTFHE_ERR_INVALID, "steps out of range");
    return sharded_host_bootstrap(ctx, RTFHE_COPY, MODE_BLIND_ROTATE, steps, tlwe, nullptr, acc, count, (size_t)2 * ctx->p.N);
}

int rtfhe_mux_batch(rtfhe_ctx* ctx, const uint32_t* c, const uint32_t* in0, const uint32_t* in1, uint32_t* out, size_t count) {
    if (int rc = use(ctx)) return rc;
    if (!c || !in0 || !in1 || !out) return fail(ctx, RTFHE_ERR_INVALID, "null argument");
    if (!ctx->has_bk) return fail(ctx, RTFHE_ERR_STATE, "bootstrapping key not loaded");
    if (!ctx->has_ksk) return fail(ctx, RTFHE_ERR_STATE, "key-switching key not loaded");
    if (count > 0x7fffffff) return fail(ctx, RTFHE_ERR_INVALID, "count too large");
    return sharded_host_mux(ctx, c, in0, in1, out, count);
}

}  // extern "C"
