/*
 * rtfhe_spqlios.h -- the reference's ONLY existing FFI, by name: the eight Spqlios_* symbols its `utils` crate binds
 * (utils/src/spqlios.rs:18-32) and utils/src/spqlios/spqlios-wrapper.cpp:9-53 defines.  librtfhe_hip.so exports them with the
 * same names, argument order and buffer ownership, so the reference's crate links against the engine with no source change
 * (swap `cargo:rustc-link-lib=static=spqlios` of utils/build.rs for `dylib=rtfhe_hip`).  Each call is the count = 1 case of the
 * batched entry point named beside it (rtfhe.h) and runs on the GPU; there is no CPU fallback.
 *
 * Contract, as in the reference:
 *   - the handle is opaque and owned by the caller's `Spqlios` wrapper (spqlios.rs:34-45); one handle per thread, not thread-safe
 *     (math.rs:349-351: thread_local FFT_MAP);
 *   - res / src are caller-allocated arrays of N elements (FrrSeries: Re[0..N/2) then Im[0..N/2), spqlios.rs:147,205-208);
 *   - no error returns: the reference's require() aborts (spqlios-fft-impl.cpp:92-97), and so does a failed call here, after
 *     printing the engine's error to stderr.
 * Sizes: every power of two 16 <= N <= 2048, as the reference (Spqlios::new asserts N >= 16 and a power of two, spqlios.rs:40-50; its
 * unit test runs at 16, :243-276).  N = 1024 / 2048 run on the gate path's wave-resident transforms, the others on rtfhe_fft_plan
 * (rtfhe.h): one workgroup per polynomial, the same butterfly networks, the same bytes.
 * Differences, all in the direction of safety: Spqlios_new returns NULL for any other N (the reference aborts in require()) or when no
 * MI355X is usable; Spqlios_destructor also releases the handle's memory (the reference runs the C++ destructor only and leaks the
 * object, spqlios-wrapper.cpp:14-16 -- its Rust side never touches the pointer again, spqlios.rs:139-145).  Device:
 * RTFHE_SPQLIOS_DEVICE (default 0).
 */
#ifndef RTFHE_SPQLIOS_H
#define RTFHE_SPQLIOS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct SpqliosImpl SpqliosImpl;

SpqliosImpl *Spqlios_new(const int32_t N);                                          /* spqlios-wrapper.cpp:9-12  ; rtfhe_ctx_create     */
void Spqlios_destructor(SpqliosImpl *si);                                           /* :14-16                    ; rtfhe_ctx_destroy    */
void Spqlios_ifft(SpqliosImpl *si, double *res, const double *src);                 /* :18-20 execute_reverse        ; rtfhe_ifft_f64_batch */
void Spqlios_ifft_u32(SpqliosImpl *si, double *res, const uint32_t *src);           /* :22-24 execute_reverse_torus32; rtfhe_ifft_i32_batch */
void Spqlios_ifft_i32(SpqliosImpl *si, double *res, const int32_t *src);            /* :26-28 execute_reverse_int    ; rtfhe_ifft_i32_batch */
void Spqlios_fft(SpqliosImpl *si, double *res, const double *src);                  /* :30-32 execute_direct         ; rtfhe_fft_f64_batch  */
void Spqlios_fft_u32(SpqliosImpl *si, uint32_t *res, const double *src);            /* :34-36 execute_direct_torus32 ; rtfhe_fft_u32_batch  */
void Spqlios_poly_mul(SpqliosImpl *si, uint32_t *res, const uint32_t *src_a, const uint32_t *src_b);   /* :38-53 ; rtfhe_poly_mul_batch */

#ifdef __cplusplus
}
#endif
#endif
