//! Raw bindings; one declaration per entry point of `include/rtfhe.h` (same order).
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rtfhe_params {
    pub n: i32,          // TLWE lvl0 dimension    hom_nand/src/tlwe.rs:175
    pub N: i32,          // TRLWE degree           hom_nand/src/trlwe.rs:76
    pub nbit: i32,       // log2(N)                hom_nand/src/tfhe.rs:16
    pub l: i32,          // gadget levels          hom_nand/src/trgsw.rs:115
    pub bgbit: i32,      // gadget base bits       hom_nand/src/trgsw.rs:112
    pub ks_t: i32,       // key-switch levels      hom_nand/src/tlwe.rs:178
    pub ks_basebit: i32, // key-switch base bits   hom_nand/src/tlwe.rs:179
}
/// What the runtime reported about entry d of a multi-device context against the primary (include/rtfhe.h: rtfhe_peer_info).
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rtfhe_peer_info {
    pub device: i32,
    pub same_device: i32,
    pub can_access_from_primary: i32,
    pub can_access_to_primary: i32,
    pub enabled_from_primary: i32,
    pub enabled_to_primary: i32,
    pub link_type: u32,
    pub hops: u32,
    pub scatter_ms: f32,
    pub compute_ms: f32,
    pub gather_ms: f32,
}
pub enum rtfhe_ctx {}
pub enum rtfhe_circuit {}
pub enum rtfhe_fft_plan {}

pub const RTFHE_NAND: c_int = 0;
pub const RTFHE_AND: c_int = 1;
pub const RTFHE_OR: c_int = 2;
pub const RTFHE_XOR: c_int = 3;
pub const RTFHE_NOT: c_int = 4;
pub const RTFHE_COPY: c_int = 5;
pub const RTFHE_ANDNY: c_int = 6;

pub const RTFHE_BACKEND_FFT64_MIRROR: c_int = 0;
pub const RTFHE_BACKEND_NTT_EXACT: c_int = 1;
pub const RTFHE_BACKEND_FFT_SPLIT_EXACT: c_int = 2;

pub const RTFHE_OK: c_int = 0;
pub const RTFHE_ERR_INVALID: c_int = -1;
pub const RTFHE_ERR_NO_DEVICE: c_int = -2;
pub const RTFHE_ERR_HIP: c_int = -3;
pub const RTFHE_ERR_STATE: c_int = -4;
pub const RTFHE_ERR_NOMEM: c_int = -5;

extern "C" {
    pub fn rtfhe_default_params(p: *mut rtfhe_params);
    pub fn rtfhe_ctx_create(p: *const rtfhe_params, device_id: c_int, out: *mut *mut rtfhe_ctx) -> c_int;
    pub fn rtfhe_ctx_create_multi(p: *const rtfhe_params, device_ids: *const c_int, n_dev: c_int, out: *mut *mut rtfhe_ctx) -> c_int;
    pub fn rtfhe_ctx_device_count(ctx: *const rtfhe_ctx) -> c_int;
    pub fn rtfhe_ctx_memory_bytes(ctx: *const rtfhe_ctx, d: c_int, bytes: *mut usize) -> c_int;
    pub fn rtfhe_ctx_peer_info(ctx: *mut rtfhe_ctx, d: c_int, out: *mut rtfhe_peer_info) -> c_int;
    pub fn rtfhe_device_link(dev_a: c_int, dev_b: c_int, can_access: *mut i32, link_type: *mut u32, hops: *mut u32) -> c_int;
    pub fn rtfhe_shard_range(count: usize, d: c_int, n_dev: c_int, begin: *mut usize, end: *mut usize) -> c_int;
    pub fn rtfhe_host_alloc(bytes: usize) -> *mut c_void;
    pub fn rtfhe_host_free(p: *mut c_void);
    pub fn rtfhe_ctx_destroy(ctx: *mut rtfhe_ctx);
    pub fn rtfhe_last_error(ctx: *const rtfhe_ctx) -> *const c_char;
    pub fn rtfhe_version() -> *const c_char;
    pub fn rtfhe_device_count() -> c_int;
    pub fn rtfhe_set_backend(ctx: *mut rtfhe_ctx, backend: c_int) -> c_int;
    pub fn rtfhe_get_backend(ctx: *const rtfhe_ctx) -> c_int;
    pub fn rtfhe_get_twiddles(ctx: *const rtfhe_ctx, ifft_table: *mut f64, fft_table: *mut f64) -> c_int;
    pub fn rtfhe_set_twiddles(ctx: *mut rtfhe_ctx, ifft_table: *const f64, fft_table: *const f64) -> c_int;
    pub fn rtfhe_ctx_params(ctx: *const rtfhe_ctx, p: *mut rtfhe_params) -> c_int;
    pub fn rtfhe_twiddles_load(ctx: *mut rtfhe_ctx, path: *const c_char, entries_changed: *mut i32) -> c_int;
    pub fn rtfhe_twiddles_write(ctx: *const rtfhe_ctx, path: *const c_char) -> c_int;
    pub fn rtfhe_twiddles_file_write(path: *const c_char, n: i32, ifft_table: *const f64, fft_table: *const f64) -> c_int;
    pub fn rtfhe_twiddles_file_read(path: *const c_char, n: i32, ifft_table: *mut f64, fft_table: *mut f64) -> c_int;

    pub fn rtfhe_load_bk_torus(ctx: *mut rtfhe_ctx, bk: *const u32) -> c_int;
    pub fn rtfhe_load_bk_fft(ctx: *mut rtfhe_ctx, bk_f: *const f64) -> c_int;
    pub fn rtfhe_export_bk_fft(ctx: *mut rtfhe_ctx, bk_f: *mut f64) -> c_int;
    pub fn rtfhe_load_ksk(ctx: *mut rtfhe_ctx, ksk: *const u32) -> c_int;
    /// `KeySwitchingKey(Vec<[[TLWERep<M>; IKS_T]; IKS_L]>)` flattened as it stands: `u32[N][8][4][n+1]` (tlwe.rs:243-245)
    pub fn rtfhe_load_ksk_ref(ctx: *mut rtfhe_ctx, ksk_ref: *const u32) -> c_int;

    pub fn rtfhe_gate_batch(ctx: *mut rtfhe_ctx, op: c_int, in0: *const u32, in1: *const u32, out: *mut u32, count: usize) -> c_int;
    pub fn rtfhe_mux_batch(ctx: *mut rtfhe_ctx, c: *const u32, in0: *const u32, in1: *const u32, out: *mut u32, count: usize) -> c_int;
    pub fn rtfhe_bootstrap_batch(ctx: *mut rtfhe_ctx, tlwe: *const u32, out: *mut u32, count: usize) -> c_int;

    pub fn rtfhe_gate_batch_dev(ctx: *mut rtfhe_ctx, op: c_int, d_in0: *const c_void, d_in1: *const c_void, d_out: *mut c_void,
                                count: usize, stream: *mut c_void) -> c_int;
    pub fn rtfhe_mux_batch_dev(ctx: *mut rtfhe_ctx, d_c: *const c_void, d_in0: *const c_void, d_in1: *const c_void, d_out: *mut c_void,
                               count: usize, stream: *mut c_void) -> c_int;
    pub fn rtfhe_bootstrap_batch_dev(ctx: *mut rtfhe_ctx, d_tlwe: *const c_void, d_out: *mut c_void, count: usize, stream: *mut c_void) -> c_int;
    pub fn rtfhe_circuit_wave_dev(ctx: *mut rtfhe_ctx, d_ops: *const c_void, d_idx0: *const c_void, d_idx1: *const c_void,
                                  d_idx_out: *const c_void, d_wires: *mut c_void, num_wires: usize, count: usize, stream: *mut c_void) -> c_int;
    pub fn rtfhe_circuit_create(ctx: *mut rtfhe_ctx, d_ops: *const c_void, d_idx0: *const c_void, d_idx1: *const c_void, d_idx_out: *const c_void,
                                wave_offsets: *const i32, num_waves: i32, d_wires: *mut c_void, num_wires: usize, out: *mut *mut rtfhe_circuit) -> c_int;
    pub fn rtfhe_circuit_launch(c: *mut rtfhe_circuit, stream: *mut c_void) -> c_int;
    pub fn rtfhe_circuit_destroy(c: *mut rtfhe_circuit);
    pub fn rtfhe_sync(ctx: *mut rtfhe_ctx, stream: *mut c_void) -> c_int;
    pub fn rtfhe_timer_begin(ctx: *mut rtfhe_ctx, stream: *mut c_void) -> c_int;
    pub fn rtfhe_timer_end(ctx: *mut rtfhe_ctx, stream: *mut c_void, ms: *mut f64, launches: *mut i64) -> c_int;
    pub fn rtfhe_timer_end_detail(ctx: *mut rtfhe_ctx, stream: *mut c_void, ms: *mut f64, key_switch_ms: *mut f64, launches: *mut i64) -> c_int;

    pub fn rtfhe_blind_rotate_batch(ctx: *mut rtfhe_ctx, tlwe: *const u32, steps: i32, acc: *mut u32, count: usize) -> c_int;
    pub fn rtfhe_external_product_batch(ctx: *mut rtfhe_ctx, bk_index: *const i32, trlwe: *const u32, out: *mut u32, count: usize) -> c_int;
    pub fn rtfhe_key_switch_batch(ctx: *mut rtfhe_ctx, tlwe1: *const u32, out: *mut u32, count: usize) -> c_int;
    pub fn rtfhe_ifft_i32_batch(ctx: *mut rtfhe_ctx, src: *const i32, res: *mut f64, count: usize) -> c_int;
    pub fn rtfhe_fft_u32_batch(ctx: *mut rtfhe_ctx, src: *const f64, res: *mut u32, count: usize) -> c_int;
    pub fn rtfhe_ifft_f64_batch(ctx: *mut rtfhe_ctx, src: *const f64, res: *mut f64, count: usize) -> c_int;
    pub fn rtfhe_fft_f64_batch(ctx: *mut rtfhe_ctx, src: *const f64, res: *mut f64, count: usize) -> c_int;
    pub fn rtfhe_poly_mul_batch(ctx: *mut rtfhe_ctx, a: *const u32, b: *const u32, res: *mut u32, count: usize) -> c_int;

    // the reference's transforms at any power of two 16 <= N <= 2048 (FFT_Processor_Spqlios(N); Spqlios::new takes them all)
    pub fn rtfhe_fft_plan_create(n: i32, device_id: c_int, out: *mut *mut rtfhe_fft_plan) -> c_int;
    pub fn rtfhe_fft_plan_destroy(plan: *mut rtfhe_fft_plan);
    pub fn rtfhe_fft_plan_degree(plan: *const rtfhe_fft_plan) -> i32;
    pub fn rtfhe_fft_plan_get_twiddles(plan: *const rtfhe_fft_plan, ifft_table: *mut f64, fft_table: *mut f64) -> c_int;
    pub fn rtfhe_fft_plan_set_twiddles(plan: *mut rtfhe_fft_plan, ifft_table: *const f64, fft_table: *const f64) -> c_int;
    pub fn rtfhe_fft_plan_ifft_i32(plan: *mut rtfhe_fft_plan, src: *const i32, res: *mut f64, count: usize) -> c_int;
    pub fn rtfhe_fft_plan_ifft_f64(plan: *mut rtfhe_fft_plan, src: *const f64, res: *mut f64, count: usize) -> c_int;
    pub fn rtfhe_fft_plan_fft_u32(plan: *mut rtfhe_fft_plan, src: *const f64, res: *mut u32, count: usize) -> c_int;
    pub fn rtfhe_fft_plan_fft_f64(plan: *mut rtfhe_fft_plan, src: *const f64, res: *mut f64, count: usize) -> c_int;
    pub fn rtfhe_fft_plan_poly_mul(plan: *mut rtfhe_fft_plan, a: *const u32, b: *const u32, res: *mut u32, count: usize) -> c_int;

    // production: randomness from the OS CSPRNG (like the reference's thread_rng)
    pub fn rtfhe_keygen(p: *const rtfhe_params, key0: *mut i32, key1: *mut i32, bk: *mut u32, ksk: *mut u32) -> c_int;
    pub fn rtfhe_keygen_with_keys(p: *const rtfhe_params, key0: *const i32, key1: *const i32, bk: *mut u32, ksk: *mut u32) -> c_int;
    pub fn rtfhe_tlwe_encrypt_bits(p: *const rtfhe_params, key0: *const i32, bits: *const u8, out: *mut u32, count: usize) -> c_int;
    // TEST ONLY (seeded xoshiro256**, not secure)
    pub fn rtfhe_ksk_expand_ref(p: *const rtfhe_params, key0: *const i32, key1: *const i32, ksk: *const u32, ksk_ref: *mut u32) -> c_int;
    pub fn rtfhe_ksk_expand_ref_deterministic(p: *const rtfhe_params, seed: u64, key0: *const i32, key1: *const i32, ksk: *const u32, ksk_ref: *mut u32) -> c_int;
    pub fn rtfhe_keygen_deterministic(p: *const rtfhe_params, seed: u64, key0: *mut i32, key1: *mut i32, bk: *mut u32, ksk: *mut u32) -> c_int;
    pub fn rtfhe_keygen_with_keys_deterministic(p: *const rtfhe_params, seed: u64, key0: *const i32, key1: *const i32, bk: *mut u32, ksk: *mut u32) -> c_int;
    pub fn rtfhe_tlwe_encrypt_bits_deterministic(p: *const rtfhe_params, key0: *const i32, seed: u64, bits: *const u8, out: *mut u32, count: usize) -> c_int;
    pub fn rtfhe_tlwe_decrypt_bits(p: *const rtfhe_params, key0: *const i32, input: *const u32, bits: *mut u8, count: usize) -> c_int;
    pub fn rtfhe_keys_write(path: *const c_char, p: *const rtfhe_params, key0: *const i32, key1: *const i32, bk: *const u32, ksk: *const u32) -> c_int;
    pub fn rtfhe_keys_read_header(path: *const c_char, p: *mut rtfhe_params, flags: *mut u32) -> c_int;
    pub fn rtfhe_keys_read(path: *const c_char, key0: *mut i32, key1: *mut i32, bk: *mut u32, ksk: *mut u32) -> c_int;
    pub fn rtfhe_tlwe_write(path: *const c_char, n: i32, cts: *const u32, count: usize) -> c_int;
    pub fn rtfhe_tlwe_read(path: *const c_char, n: *mut i32, count: *mut u64, cts: *mut u32, capacity: usize) -> c_int;
    pub fn rtfhe_tlwe_phase(p: *const rtfhe_params, key0: *const i32, input: *const u32, phase: *mut u32, count: usize) -> c_int;
}
