// rtfhe_spqlios.cpp -- the reference's FFT FFI by name (include/rtfhe_spqlios.h): Spqlios_new / _destructor / _ifft / _ifft_u32 /
// _ifft_i32 / _fft / _fft_u32 / _poly_mul, each the count = 1 case of the batched C-ABI call.  Pure host glue over rtfhe.h: no
// arithmetic happens here, and nothing falls back to the CPU.
#include "../../include/rtfhe_spqlios.h"

#include <cstdio>
#include <cstdlib>
#include <new>

#include "../../include/rtfhe.h"

struct SpqliosImpl {
    rtfhe_ctx* ctx;           // N = 1024 / 2048: the gate path's context and its wave-resident transforms
    rtfhe_fft_plan* plan;     // every other power of two 16 <= N <= 2048 (the reference takes them all, spqlios.rs:40-50): rtfhe_fft_plan
    int32_t N;
};

namespace {

// the reference's require() (spqlios-fft-impl.cpp:92-97): message, then abort -- the FFI has no error channel
[[noreturn]] void die(const SpqliosImpl* si, const char* what, int rc) {
    std::fprintf(stderr, "rtfhe spqlios shim: %s failed (%d): %s\n", what, rc, rtfhe_last_error(si ? si->ctx : nullptr));
    std::abort();
}

}  // namespace

extern "C" {

SpqliosImpl* Spqlios_new(const int32_t N) {
    int device = 0;
    if (const char* e = std::getenv("RTFHE_SPQLIOS_DEVICE")) device = std::atoi(e);
    if (N != 1024 && N != 2048) {
        rtfhe_fft_plan* plan = nullptr;
        if (rtfhe_fft_plan_create(N, device, &plan) != 0) return nullptr;      // not a power of two in [16, 2048], or no device
        SpqliosImpl* si = new (std::nothrow) SpqliosImpl{nullptr, plan, N};
        if (!si) rtfhe_fft_plan_destroy(plan);
        return si;
    }
    rtfhe_params p;
    rtfhe_default_params(&p);
    p.N = N;
    p.nbit = N == 1024 ? 10 : 11;
    rtfhe_ctx* ctx = nullptr;
    if (rtfhe_ctx_create(&p, device, &ctx) != 0) return nullptr;
    SpqliosImpl* si = new (std::nothrow) SpqliosImpl{ctx, nullptr, N};
    if (!si) rtfhe_ctx_destroy(ctx);
    return si;
}

void Spqlios_destructor(SpqliosImpl* si) {
    if (!si) return;
    if (si->ctx) rtfhe_ctx_destroy(si->ctx);
    if (si->plan) rtfhe_fft_plan_destroy(si->plan);
    delete si;
}

void Spqlios_ifft(SpqliosImpl* si, double* res, const double* src) {
    if (int rc = si->plan ? rtfhe_fft_plan_ifft_f64(si->plan, src, res, 1) : rtfhe_ifft_f64_batch(si->ctx, src, res, 1)) die(si, "Spqlios_ifft", rc);
}

void Spqlios_ifft_u32(SpqliosImpl* si, double* res, const uint32_t* src) {
    // execute_reverse_torus32 reinterprets the torus words as signed (fft_processor_spqlios.cpp:100-106)
    if (int rc = si->plan ? rtfhe_fft_plan_ifft_i32(si->plan, reinterpret_cast<const int32_t*>(src), res, 1) : rtfhe_ifft_i32_batch(si->ctx, reinterpret_cast<const int32_t*>(src), res, 1)) die(si, "Spqlios_ifft_u32", rc);
}

void Spqlios_ifft_i32(SpqliosImpl* si, double* res, const int32_t* src) {
    if (int rc = si->plan ? rtfhe_fft_plan_ifft_i32(si->plan, src, res, 1) : rtfhe_ifft_i32_batch(si->ctx, src, res, 1)) die(si, "Spqlios_ifft_i32", rc);
}

void Spqlios_fft(SpqliosImpl* si, double* res, const double* src) {
    if (int rc = si->plan ? rtfhe_fft_plan_fft_f64(si->plan, src, res, 1) : rtfhe_fft_f64_batch(si->ctx, src, res, 1)) die(si, "Spqlios_fft", rc);
}

void Spqlios_fft_u32(SpqliosImpl* si, uint32_t* res, const double* src) {
    if (int rc = si->plan ? rtfhe_fft_plan_fft_u32(si->plan, src, res, 1) : rtfhe_fft_u32_batch(si->ctx, src, res, 1)) die(si, "Spqlios_fft_u32", rc);
}

void Spqlios_poly_mul(SpqliosImpl* si, uint32_t* res, const uint32_t* src_a, const uint32_t* src_b) {
    if (int rc = si->plan ? rtfhe_fft_plan_poly_mul(si->plan, src_a, src_b, res, 1) : rtfhe_poly_mul_batch(si->ctx, src_a, src_b, res, 1)) die(si, "Spqlios_poly_mul", rc);
}

}  // extern "C"
