// rtfhe_stages.hip -- stage-level kernels and their entry points: the transforms, the external product, the key switch (per gate and, for a
// whole batch, as one i8 contraction), the key permutes, and the reference's transforms at any power of two (FFT plans).  The parity tests use
// these to localise a mismatch; they run the same device functions as the fused kernels.
#include "rtfhe_host.hpp"
#include <cmath>

#include "rtfhe_kernels_ksmm.hpp"
#include "rtfhe_kernels_anyn.hpp"

using namespace rtfhe;
using namespace rtfhe_host;

namespace {

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

}  // namespace

namespace rtfhe_host {

int launch_fft(rtfhe_ctx* ctx, bool forward, FftArgs a, hipStream_t s) {
    if (a.count == 0) return 0;
    return ctx->logn == 10 ? launch_fft_t<10>(ctx, forward, a, s) : launch_fft_t<11>(ctx, forward, a, s);
}

// reference order <-> device order of the key spectra (dir 0: FrrSeries -> device layout, 1: back)
int launch_bk_permute(rtfhe_ctx* ctx, const double* src, double* dst, size_t polys, int dir, hipStream_t s) {
    return ctx->logn == 10 ? launch_permute_t<10>(ctx, src, dst, polys, dir, 2 * ctx->p.l, s) : launch_permute_t<11>(ctx, src, dst, polys, dir, 2 * ctx->p.l, s);
}

// device layout of the key-switching key: the rows of two adjacent levels pre-summed (see ks_accumulate) + one all-zero row that "both digits 0" selects
int launch_ksk_combine(rtfhe_ctx* ctx, const uint32_t* d_raw, hipStream_t s) {
    KskCombineArgs a{d_raw, ctx->d_ksk, ctx->p.N, ctx->ksw};
    hipLaunchKernelGGL((k_ksk_combine<8, 2>), dim3(4096), dim3(256), 0, s, a);
    HIPCHECK(ctx, hipGetLastError());
    return 0;
}

// the same key as signed byte limbs in i8-MFMA operand order, for the batch key switch of the split path
int launch_ksmat_build(rtfhe_ctx* ctx, const uint32_t* d_raw, int colgroups, hipStream_t s) {
    KsMatArgs m{d_raw, ctx->d_ksmat, ctx->p.N, ctx->p.n, ctx->ksw, colgroups};
    hipLaunchKernelGGL((k_ksmat_build<8, 2>), dim3(4096), dim3(256), 0, s, m);
    HIPCHECK(ctx, hipGetLastError());
    return 0;
}

int launch_key_switch_mm(rtfhe_ctx* ctx, const BootstrapArgs& a, const uint32_t* samples, hipStream_t s) {
    const int per_block = 16 * KSMM_MT * KSMM_WAVES;      // gates of a workgroup: KSMM_WAVES waves x 4 tiles of 16
    const int colgroups = (ctx->p.n + 1 + 15) / 16, mgroups = (a.count + per_block - 1) / per_block;
    // K-slices (the slices of one launch add into the zeroed output; a slice is a whole number of LDS chunk pairs: four coefficients per lane
    // group).  Two workgroups fit a CU and share its matrix pipe, so what a launch costs is the LARGEST number of workgroups a CU ends up
    // with: take the fewest slices whose workgroup count fills whole rounds of the CUs (640 workgroups on 256 CUs leave half the chip idle
    // for the third of three rounds; 1,280 are five full ones).
    int splitk = 1;
    {
        double best = 0;
        for (int sk = 1; sk <= 32 && (ctx->p.N / 4) % (4 * sk) == 0; sk *= 2) {
            const double rounds = (double)mgroups * colgroups * sk / ctx->num_cus;
            const double eff = rounds / std::ceil(rounds) * std::min(1.0, rounds / 2.0);
            if (eff > best + 0.03) { best = eff; splitk = sk; }
        }
    }
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    // (no bracketing inside rtfhe_circuit_create's capture: a recorded event would become a graph node and rtfhe_timer_end would then ask
    // a never-recorded event for its time; and a timer that is never ended stops taking events at 4096 pairs)
    if (ctx->timing && !ctx->tlwe1_capture && ctx->ks_events_used + 2 <= 8192) {
        while (ctx->ks_events.size() < ctx->ks_events_used + 2) { hipEvent_t e; HIPCHECK(ctx, hipEventCreate(&e)); ctx->ks_events.push_back(e); }
        ev_a = ctx->ks_events[ctx->ks_events_used]; ev_b = ctx->ks_events[ctx->ks_events_used + 1];
        ctx->ks_events_used += 2;
        HIPCHECK(ctx, hipEventRecord(ev_a, s));
    }
    // the kernel addresses a launch's samples (tiles of 16 gates, rtfhe::ext_slot) with 32-bit byte offsets: at most 2 GiB of them per launch
    const size_t w1 = (size_t)ctx->p.N + 1, seg_max = ((size_t)1 << 31) / (w1 * 4) / per_block * per_block;
    for (size_t off = 0; off < (size_t)a.count; off += seg_max) {
        const size_t cnt = std::min(seg_max, (size_t)a.count - off);
        const int mg = (int)((cnt + per_block - 1) / per_block);
        KsMmArgs k{samples + off * w1, ctx->d_ksmat, a.idx_out ? a.out : a.out + off * ((size_t)ctx->p.n + 1), (int32_t)cnt, ctx->p.n, ctx->p.N, colgroups, splitk, mg,
                   a.ops ? a.ops + off : nullptr, a.idx0 ? a.idx0 + off : nullptr, a.idx1 ? a.idx1 + off : nullptr, a.idx_out ? a.idx_out + off : nullptr, a.num_wires};
        hipLaunchKernelGGL((k_key_switch_mm<8, 2>), dim3(mg * ((colgroups + 7) / 8 * 8) * splitk), dim3(64 * KSMM_WAVES), 0, s, k);
        HIPCHECK(ctx, hipGetLastError());
    }
    if (ev_b) HIPCHECK(ctx, hipEventRecord(ev_b, s));
    ctx->launches++;
    return 0;
}

}  // namespace rtfhe_host

extern "C" {

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
        if ((rc = launch_extprod_ntt(ctx, (const int32_t*)ctx->d_b, (const uint32_t*)ctx->d_a, (uint32_t*)ctx->d_c, (int32_t)count, ctx->stream))) return rc;
    } else if (ctx->backend == RTFHE_BACKEND_FFT_SPLIT_EXACT) {
        if ((rc = xfft_prepare(ctx))) return rc;
        if ((rc = launch_extprod_xfft(ctx, (const int32_t*)ctx->d_b, (const uint32_t*)ctx->d_a, (uint32_t*)ctx->d_c, (int32_t)count, ctx->stream))) return rc;
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

