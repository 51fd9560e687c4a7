/*
 * rtfhe.h -- C ABI of librtfhe_hip.so, the MI355X (gfx950) engine for the HomNAND hot path of
 * hideki1217/rusTfhe.  Plain pointers and sizes only; no C++ or torch types cross this boundary.
 *
 * What each entry point replaces in the reference (paths relative to the reference repo root):
 *
 *   rtfhe_ctx_create / _destroy      <- thread_local FFT_MAP + Spqlios_new / Spqlios_destructor
 *                                       (utils/src/math.rs:349-360, utils/src/spqlios.rs:18-20,139-145)
 *   rtfhe_ctx_create_multi           <- (no reference counterpart: it is single-threaded, tlwe.rs:264 TODO) one context over the
 *                                       GPUs of a node; the same batch calls then shard gates over them
 *   rtfhe_load_bk_torus              <- BootstrappingKey::new's TRGSWRepF::from (hom_nand/src/tfhe.rs:119-126,
 *                                       hom_nand/src/trgsw.rs:68-76)
 *   rtfhe_load_bk_fft                <- BootstrappingKey(Vec<TRGSWRepF>) as the reference holds it (tfhe.rs:116)
 *   rtfhe_load_ksk_ref               <- KeySwitchingKey(Vec<[[TLWERep<M>; IKS_T]; IKS_L]>) = [[TLWERep; 4]; 8] per coefficient, exactly as
 *                                       the reference holds it (hom_nand/src/tlwe.rs:178-180, 243-245): entry [i][l][t-1] = get(i, l, t),
 *                                       t = 1 .. 4 (:252-283)
 *   rtfhe_load_ksk                   <- the same key without the entry t = 4 of every level, which identity_key_switch never reads
 *                                       (its digits are basebit = 2 bits wide, tlwe.rs:43-73): [[TLWERep; 3]; 8]
 *   rtfhe_gate_batch[_dev]           <- TFHE::hom_nand/and/or/xor/not (hom_nand/src/tfhe.rs:41-71), count gates at once
 *   rtfhe_mux_batch[_dev]            <- TFHE::hom_mux (tfhe.rs:27-40)
 *   rtfhe_circuit_wave_dev           <- eval_logic_expr over impl Logip for TFHE (nander/src/lib.rs:40-89), one level at a time
 *   rtfhe_circuit_create / _launch   <- the same evaluation, all levels of a netlist recorded once and replayed as one submission
 *   rtfhe_bootstrap_batch[_dev]      <- TFHE::bootstrap (tfhe.rs:73-80)
 *   rtfhe_blind_rotate_batch         <- TFHE::blind_rotate with the gate test vector (tfhe.rs:81-113)
 *   rtfhe_external_product_batch     <- Cross for TRGSWRepF (hom_nand/src/trgsw.rs:264-306)
 *   rtfhe_key_switch_batch           <- TLWERep::identity_key_switch (hom_nand/src/tlwe.rs:43-73)
 *   rtfhe_ifft_i32_batch             <- Spqlios_ifft_i32 / _u32 (utils/src/spqlios.rs:22-23, spqlios-wrapper.cpp:22-28)
 *   rtfhe_fft_u32_batch              <- Spqlios_fft_u32 (utils/src/spqlios.rs:25, spqlios-wrapper.cpp:34-36)
 *   rtfhe_ifft_f64_batch / _fft_f64_batch / _poly_mul_batch
 *                                    <- Spqlios_ifft / Spqlios_fft / Spqlios_poly_mul (spqlios-wrapper.cpp:18-20,30-32,38-53)
 *   rtfhe_keys_* / rtfhe_tlwe_write/read  (no reference counterpart: it has no serialization; fixes App. A's layouts into files)
 *   rtfhe_keygen / rtfhe_tlwe_*      <- TFHE::new, Cryptor::encrypto/decrypto(TLWE, ..) (tfhe.rs:21-25, tlwe.rs:213-241);
 *                                       randomness from the OS CSPRNG like the reference's thread_rng; *_deterministic = seeded, TEST ONLY
 *
 * Conventions: every call returns 0 on success or a negative rtfhe_status; nothing aborts or throws
 * across the ABI; rtfhe_last_error() gives the message of the last failure on that context.  The caller
 * owns every buffer.  Host-pointer calls copy in/out and are synchronous; *_dev calls take device
 * pointers, enqueue on the given hipStream_t (passed as void*) and return without synchronising.
 * A context is bound to one device (rtfhe_ctx_create) or to a set of devices (rtfhe_ctx_create_multi) and is not thread-safe (the reference's handle is not either:
 * one per thread, utils/src/math.rs:349-351).  There is NO CPU fallback: without a usable HIP device
 * rtfhe_ctx_create fails with RTFHE_ERR_NO_DEVICE.
 *
 * Flat little-endian layouts (the reference defines none):
 *   TLWE lvl0   u32[n+1]           a[0..n), b
 *   TLWE lvl1   u32[N+1]           a'[0..N), b'
 *   TRLWE       u32[2][N]          b(X), a(X)
 *   BK torus    u32[n][2][2l][N]   comp 0 = TRGSWRep.cipher rows, comp 1 = TRGSWRep.p_key rows
 *   BK fft      f64[n][2][2l][N]   same order; each poly an FrrSeries: Re[0..N/2) then Im[0..N/2),
 *                                  in the transform's native order (utils/src/spqlios.rs:147,205-208)
 *   KSK         u32[N][t][base-1][n+1]   rows d = 1 .. base-1 of level l of coefficient i   (rtfhe_load_ksk, rtfhe_keygen*)
 *   KSK (ref)   u32[N][t][base][n+1]     the reference's [[TLWERep; IKS_T]; IKS_L]: one more row (t = base) per level, never read
 *                                        (rtfhe_load_ksk_ref drops it)
 */
#ifndef RTFHE_H
#define RTFHE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rtfhe_ctx rtfhe_ctx;

typedef struct {
    int32_t n;          /* TLWE lvl0 dimension    (635)   hom_nand/src/tlwe.rs:175  */
    int32_t N;          /* TRLWE degree           (1024)  hom_nand/src/trlwe.rs:76  */
    int32_t nbit;       /* log2(N)                (10)    hom_nand/src/tfhe.rs:16   */
    int32_t l;          /* gadget levels          (3)     hom_nand/src/trgsw.rs:115 */
    int32_t bgbit;      /* gadget base bits       (6)     hom_nand/src/trgsw.rs:112 */
    int32_t ks_t;       /* key-switch levels      (8)     hom_nand/src/tlwe.rs:178  */
    int32_t ks_basebit; /* key-switch base bits   (2)     hom_nand/src/tlwe.rs:179  */
} rtfhe_params;

typedef enum {
    RTFHE_NAND = 0, RTFHE_AND = 1, RTFHE_OR = 2, RTFHE_XOR = 3, RTFHE_NOT = 4, RTFHE_COPY = 5,
    RTFHE_ANDNY = 6   /* hom_and(-in0, in1): the second AND of hom_mux, hom_nand/src/tfhe.rs:34 */
} rtfhe_gate;

typedef enum {
    RTFHE_OK = 0,
    RTFHE_ERR_INVALID = -1,     /* bad argument / unsupported parameter set */
    RTFHE_ERR_NO_DEVICE = -2,   /* no usable HIP device */
    RTFHE_ERR_HIP = -3,         /* a HIP runtime call failed */
    RTFHE_ERR_STATE = -4,       /* keys not loaded yet */
    RTFHE_ERR_NOMEM = -5
} rtfhe_status;

/* polynomial-multiply backend of the external product */
typedef enum {
    RTFHE_BACKEND_FFT64_MIRROR = 0,  /* default: FP64 transform mirroring the reference's spqlios operation for operation;
                                        outputs bit-identical to the reference CPU path */
    RTFHE_BACKEND_NTT_EXACT = 1,     /* exact negacyclic NTT mod P = 2^50 - 16383 (N = 1024 and 2048): reference semantics of the exact
                                        Polynomial::cross (utils/src/math.rs:238-257); bit-identical to an exact-integer
                                        evaluation, decrypt-level parity with the reference's FFT path */
    RTFHE_BACKEND_FFT_SPLIT_EXACT = 2 /* the same exact products (bit-identical to RTFHE_BACKEND_NTT_EXACT) through an FMA-contracted FP64 FFT:
                                        the key split into signed 16-bit halves, each half-product rounded to the nearest integer (distance from
                                        an integer proven < 2^-8 at N = 1024, < 2^-6 at N = 2048, for every input) and recombined mod 2^32;
                                        N = 1024 and 2048; needs the key in torus form; 1.4 x / 1.7 x the NTT backend's rate */
} rtfhe_backend;

/* ---- context ---- */
void rtfhe_default_params(rtfhe_params *p);
int rtfhe_ctx_create(const rtfhe_params *p, int device_id, rtfhe_ctx **out);
/* One context over n_dev GPUs of the node (what a Rust `hom_nand_batch` binds for a whole-node batch).
 * device_ids[0] is the primary.  Keys loaded into the context are transformed once on the primary and copied
 * device-to-device to the others.  Every batch call shards contiguous gate ranges over the devices -- device d gets
 * [count d / n_dev, count (d+1) / n_dev) -- and outputs land at the same indices as on one device, bit-identical:
 *   - host-pointer calls (rtfhe_gate_batch, rtfhe_mux_batch, rtfhe_bootstrap_batch, rtfhe_blind_rotate_batch): one host thread and one
 *     stream per device, direct host<->device copies per device;
 *   - device-pointer calls (rtfhe_gate_batch_dev, rtfhe_mux_batch_dev, rtfhe_bootstrap_batch_dev): the batch lives on the PRIMARY device;
 *     every other device pulls its range over xGMI (hipMemcpyPeerAsync on its own stream), bootstraps it and pushes the outputs back,
 *     while the primary computes its own range; the caller's stream then waits for the other devices' events, so stream order holds as
 *     on one device and the call stays asynchronous.  Inside a stream capture the whole batch stays on the primary.
 * Netlist waves / circuits and stage-level calls run on the primary device only.  A device may be named more than once: every entry is
 * a full context of its own (stream, staging buffers, key replica). */
int rtfhe_ctx_create_multi(const rtfhe_params *p, const int *device_ids, int n_dev, rtfhe_ctx **out);
int rtfhe_ctx_device_count(const rtfhe_ctx *ctx);      /* devices behind this context (1 for rtfhe_ctx_create) */
/* device memory entry d (0 = primary) of the context holds right now, in bytes: the keys in every form built so far (second layouts of
 * the bootstrapping key are built by the first batch whose kernel shape reads them), staging and scratch buffers.  Not counted: the twiddle
 * tables (a few hundred KiB) and the sample buffers of live circuits. */
int rtfhe_ctx_memory_bytes(const rtfhe_ctx *ctx, int d, size_t *bytes);
/* What the runtime reported about entry d (1 <= d < rtfhe_ctx_device_count) of a multi-device context against entry 0, the primary, when
 * rtfhe_ctx_create_multi set it up -- peer access is queried in both directions and enabled explicitly there, never left to a first copy -- and
 * how long that entry's share of the LAST device-resident sharded batch (rtfhe_gate_batch_dev / rtfhe_mux_batch_dev / rtfhe_bootstrap_batch_dev)
 * took on its own stream.  Nothing here is assumed: a field says what a HIP call returned on this machine. */
typedef struct rtfhe_peer_info {
    int32_t device;                   /* HIP device id of entry d */
    int32_t same_device;              /* 1: entry d names the primary's own device (a rehearsal on one card: no peer access involved) */
    int32_t can_access_from_primary;  /* hipDeviceCanAccessPeer(primary -> entry d) */
    int32_t can_access_to_primary;    /* hipDeviceCanAccessPeer(entry d -> primary) */
    int32_t enabled_from_primary;     /* hipDeviceEnablePeerAccess on the primary for entry d's device succeeded (or was already in force) */
    int32_t enabled_to_primary;       /* ... on entry d's device for the primary */
    uint32_t link_type;               /* hipExtGetLinkTypeAndHopCount(primary, entry d): 1 HyperTransport, 2 QPI, 3 PCIe, 4 InfiniBand, 5 xGMI;
                                       * 0xffffffff when the query failed or same_device */
    uint32_t hops;
    float scatter_ms;                 /* last sharded device-resident batch: entry d pulling its range of the inputs from the primary, */
    float compute_ms;                 /* bootstrapping it, */
    float gather_ms;                  /* pushing its outputs into the caller's buffer on the primary; -1 when there was no such batch yet,
                                       * the entry had no gates in it, or it has not completed (call rtfhe_sync first) */
} rtfhe_peer_info;
int rtfhe_ctx_peer_info(rtfhe_ctx *ctx, int d, rtfhe_peer_info *out);
/* the same questions about any two devices of the node, without a context (a one-process-per-GPU job prints this per rank): *can_access =
 * hipDeviceCanAccessPeer(dev_a -> dev_b), *link_type / *hops = hipExtGetLinkTypeAndHopCount (0xffffffff / 0 when the query fails).  Queries
 * only: nothing is enabled. */
int rtfhe_device_link(int dev_a, int dev_b, int32_t *can_access, uint32_t *link_type, uint32_t *hops);
/* the range [*begin, *end) of a `count`-gate host batch that entry d of an n_dev-device context bootstraps (no GPU needed) */
int rtfhe_shard_range(size_t count, int d, int n_dev, size_t *begin, size_t *end);
void rtfhe_ctx_destroy(rtfhe_ctx *ctx);
/* pinned host memory for ciphertext buffers: host-pointer calls DMA straight from / into such a buffer.  Any other host
 * pointer is handed to the runtime's own pageable-copy path (measured faster than staging it here); RTFHE_STAGING=1 in the
 * environment stages pageable buffers through the context's pinned buffers instead (one extra host copy). */
void *rtfhe_host_alloc(size_t bytes);
void rtfhe_host_free(void *p);
const char *rtfhe_last_error(const rtfhe_ctx *ctx);   /* ctx may be NULL: last ctx-less error */
const char *rtfhe_version(void);
int rtfhe_device_count(void);
int rtfhe_set_backend(rtfhe_ctx *ctx, int backend);   /* takes effect for subsequent calls; the NTT-domain key is derived from
                                                         the torus-form key (rtfhe_load_bk_torus) on first use */
int rtfhe_get_backend(const rtfhe_ctx *ctx);
/* twiddle tables in the reference's memory layout (2N doubles each direction; blocks 4 cos | 4 sin) */
int rtfhe_get_twiddles(const rtfhe_ctx *ctx, double *ifft_table, double *fft_table);
int rtfhe_set_twiddles(rtfhe_ctx *ctx, const double *ifft_table, const double *fft_table);
int rtfhe_ctx_params(const rtfhe_ctx *ctx, rtfhe_params *p);     /* the parameter set the context was created with */
/* The tables are DATA: the reference builds them with libm's cos / sin of a double-rounded angle (accurate_cos / accurate_sin,
 * utils/src/spqlios/spqlios-fft-impl.cpp:99-113), two hosts' libms may differ by an ulp in a few entries, and one differing entry changes
 * torus words.  rtfhe_ctx_create builds them with THIS host's libm -- what a reference built on this host would hold.  To
 * reproduce another build's bits, ship its tables as a file ("RTFHETW1" | i32 N | i32 0 | f64 ifft_table[2N] | f64 fft_table[2N] | u64 fnv1a):
 * rtfhe_twiddles_load installs the file's tables only if they differ from the context's (*entries_changed = differing entries, may be
 * NULL; a key loaded in torus form is re-transformed); rtfhe_twiddles_write saves the context's.  rustfhe_amd/assets/twiddles_N*.bin are
 * the tables of the reference build the golden vectors under tests/golden/ were made with. */
int rtfhe_twiddles_load(rtfhe_ctx *ctx, const char *path, int32_t *entries_changed);
int rtfhe_twiddles_write(const rtfhe_ctx *ctx, const char *path);
/* the same file format without a context (pure file I/O, checksum verified on read): N = ring degree, 2N doubles per table */
int rtfhe_twiddles_file_write(const char *path, int32_t N, const double *ifft_table, const double *fft_table);
int rtfhe_twiddles_file_read(const char *path, int32_t N, double *ifft_table, double *fft_table);

/* ---- keys ----
 * Loading a bootstrapping key (or new twiddle tables under a torus-form key) into a context that already holds one replaces it IN PLACE: the
 * spectra and every further form of the key that has been built (second kernel layouts, the exact backends' forms) are rebuilt in their
 * existing buffers before the call returns, on every device of the context.  A recorded circuit, or a capture the caller took around a *_dev
 * call, therefore replays against the new key.  One exception: a key given as spectra (rtfhe_load_bk_fft) has no torus form, the exact
 * backends' forms cannot follow it, and circuits recorded on those backends fail with RTFHE_ERR_STATE from then on (record them again after
 * loading a torus-form key); a caller's own capture of an exact-backend batch must be retaken in that case. */
int rtfhe_load_bk_torus(rtfhe_ctx *ctx, const uint32_t *bk /* [n][2][2l][N] */);
int rtfhe_load_bk_fft(rtfhe_ctx *ctx, const double *bk_f /* [n][2][2l][N] */);
int rtfhe_export_bk_fft(rtfhe_ctx *ctx, double *bk_f /* [n][2][2l][N] */);
int rtfhe_load_ksk(rtfhe_ctx *ctx, const uint32_t *ksk /* [N][t][base-1][n+1] */);
/* the reference's container flattened as it stands: [N][IKS_L = t][IKS_T = base][n+1], each TLWERep as a[0..n) then b */
int rtfhe_load_ksk_ref(rtfhe_ctx *ctx, const uint32_t *ksk_ref /* [N][t][base][n+1] */);

/* ---- the hot path: host buffers ---- */
int rtfhe_gate_batch(rtfhe_ctx *ctx, int op, const uint32_t *in0, const uint32_t *in1,
                     uint32_t *out, size_t count);            /* [count][n+1] each; in1 ignored for NOT/COPY */
int rtfhe_mux_batch(rtfhe_ctx *ctx, const uint32_t *c, const uint32_t *in0, const uint32_t *in1,
                    uint32_t *out, size_t count);
int rtfhe_bootstrap_batch(rtfhe_ctx *ctx, const uint32_t *tlwe, uint32_t *out, size_t count);

/* ---- the hot path: device buffers, asynchronous on `stream` (a hipStream_t, may be NULL) ---- */
int rtfhe_gate_batch_dev(rtfhe_ctx *ctx, int op, const void *d_in0, const void *d_in1, void *d_out,
                         size_t count, void *stream);
int rtfhe_mux_batch_dev(rtfhe_ctx *ctx, const void *d_c, const void *d_in0, const void *d_in1, void *d_out,
                        size_t count, void *stream);          /* d_out may alias none of the inputs.  The two intermediate batches live
                                                                 * in buffers of the context that belong to `stream` (MUX batches on different
                                                                 * streams may overlap).  Inside a caller's stream capture the call succeeds only
                                                                 * if an eager MUX batch of at least `count` gates ran on that stream before
                                                                 * (nothing may be allocated inside a capture): RTFHE_ERR_STATE otherwise; the
                                                                 * buffers a capture used are then kept until the context is destroyed. */
int rtfhe_bootstrap_batch_dev(rtfhe_ctx *ctx, const void *d_tlwe, void *d_out, size_t count, void *stream);
/* one dependency wave of a gate netlist (the build-side counterpart of nander's eager tree walk, nander/src/lib.rs:72-89):
 * gate g reads rows idx0[g] and idx1[g] of the wire table d_wires (u32[num_wires][n+1]), applies ops[g] and writes row
 * idx_out[g]; all four arrays are int32[count] in device memory.  Gates of one call must be independent.  Indices and
 * opcodes are validated on the device against num_wires: an offending gate is skipped (nothing is read or written through
 * it) and the next rtfhe_sync returns RTFHE_ERR_INVALID. */
int rtfhe_circuit_wave_dev(rtfhe_ctx *ctx, const void *d_ops, const void *d_idx0, const void *d_idx1,
                           const void *d_idx_out, void *d_wires, size_t num_wires, size_t count, void *stream);
/* A whole levelised netlist as ONE submission (BASELINE config 4; the reference walks its expression tree gate by gate,
 * nander/src/lib.rs:72-89): the waves wave_offsets[w] .. wave_offsets[w+1] (host array, num_waves + 1 entries) of the same four
 * device arrays are captured once into a HIP graph; rtfhe_circuit_launch replays it on `stream` (asynchronous, one runtime
 * call per evaluation).  The device arrays and the wire table must stay alive and in place while the circuit exists.  A circuit
 * is normally destroyed before its context; if the context goes first, the circuit's graph is released with it, a later
 * rtfhe_circuit_launch fails with RTFHE_ERR_STATE and rtfhe_circuit_destroy only frees the handle. */
typedef struct rtfhe_circuit rtfhe_circuit;
int rtfhe_circuit_create(rtfhe_ctx *ctx, const void *d_ops, const void *d_idx0, const void *d_idx1, const void *d_idx_out,
                         const int32_t *wave_offsets, int32_t num_waves, void *d_wires, size_t num_wires, rtfhe_circuit **out);
int rtfhe_circuit_launch(rtfhe_circuit *c, void *stream);
void rtfhe_circuit_destroy(rtfhe_circuit *c);
/* waits for `stream`; also reports (once) a netlist gate skipped since the previous call */
int rtfhe_sync(rtfhe_ctx *ctx, void *stream);
/* device-side timing of the launches enqueued by the *_dev calls between begin and end (HIP events on
 * `stream`); end returns total milliseconds and the number of kernel launches */
int rtfhe_timer_begin(rtfhe_ctx *ctx, void *stream);
int rtfhe_timer_end(rtfhe_ctx *ctx, void *stream, double *ms, int64_t *launches);
/* the same, and of the total the device time spent in the batch key switches of the split path.  Every batch of at least
 * RTFHE_KS_MM_MIN gates (environment, default 1: every batch -- any size, netlist waves, N = 1024 and 2048, both backends) runs as two
 * launches: blind rotation + sample extract (which also zeroes the gates' output rows), then the key switch of the whole batch as one
 * exact i8 contraction (k_key_switch_mm).  RTFHE_KS_MM_MIN=0 keeps the key switch fused into the bootstrap kernel; so does a batch
 * enqueued inside a caller's own stream capture (the split path's scratch buffer belongs to the stream, not to the caller's graph) --
 * key_switch_ms is 0 when all were fused, and the launches of an rtfhe_circuit are never bracketed. */
int rtfhe_timer_end_detail(rtfhe_ctx *ctx, void *stream, double *ms, double *key_switch_ms, int64_t *launches);

/* ---- stage-level entry points (parity tests; same kernels' building blocks) ---- */
int rtfhe_blind_rotate_batch(rtfhe_ctx *ctx, const uint32_t *tlwe /* [count][n+1] */, int32_t steps,
                             uint32_t *acc /* [count][2][N] */, size_t count);
int rtfhe_external_product_batch(rtfhe_ctx *ctx, const int32_t *bk_index /* [count] */,
                                 const uint32_t *trlwe /* [count][2][N] */, uint32_t *out, size_t count);
int rtfhe_key_switch_batch(rtfhe_ctx *ctx, const uint32_t *tlwe1 /* [count][N+1] */,
                           uint32_t *out /* [count][n+1] */, size_t count);
int rtfhe_ifft_i32_batch(rtfhe_ctx *ctx, const int32_t *src /* [count][N] */, double *res /* [count][N] */, size_t count);
int rtfhe_fft_u32_batch(rtfhe_ctx *ctx, const double *src /* [count][N] */, uint32_t *res /* [count][N] */, size_t count);
/* the rest of the reference's FFT FFI (off the gate path; utils/src/spqlios.rs:18-32, only its N = 16 unit test uses poly_mul) */
int rtfhe_ifft_f64_batch(rtfhe_ctx *ctx, const double *src /* [count][N] */, double *res /* [count][N] */, size_t count);
int rtfhe_fft_f64_batch(rtfhe_ctx *ctx, const double *src /* [count][N] */, double *res /* [count][N], no truncation */, size_t count);
int rtfhe_poly_mul_batch(rtfhe_ctx *ctx, const uint32_t *a, const uint32_t *b, uint32_t *res /* [count][N] each */, size_t count);

/* ---- the reference's transforms at ANY power of two 16 <= N <= 2048 ----
 * The reference's FFT FFI takes every such N (Spqlios::new, utils/src/spqlios.rs:40-50; FFT_Processor_Spqlios, fft_processor_spqlios.cpp:7-14)
 * and its own unit test runs at N = 16 (spqlios.rs:243-276); a context exists for the gate path's N = 1024 / 2048 only.  A plan holds
 * the twiddle tables of one N on one device and runs the same butterfly networks (ifft_model / fft_model, spqlios-fft-impl.cpp:469-641 /
 * 204-397) on the GPU, one workgroup per polynomial: the same bytes as the reference for every N, with no claim of speed.  Not
 * thread-safe (one plan per host thread, as the reference's thread_local FFT_MAP, math.rs:349-351); host pointers in and out; errors
 * through rtfhe_last_error(NULL).  The Spqlios_* symbols of rtfhe_spqlios.h use a plan for every N other than 1024 / 2048. */
typedef struct rtfhe_fft_plan rtfhe_fft_plan;
int rtfhe_fft_plan_create(int32_t N, int device_id, rtfhe_fft_plan **out);      /* FFT_Processor_Spqlios(N), fft_processor_spqlios.cpp:7-14 */
void rtfhe_fft_plan_destroy(rtfhe_fft_plan *plan);
int32_t rtfhe_fft_plan_degree(const rtfhe_fft_plan *plan);
int rtfhe_fft_plan_get_twiddles(const rtfhe_fft_plan *plan, double *ifft_table /* [2N] */, double *fft_table /* [2N] */);   /* reference layout */
int rtfhe_fft_plan_set_twiddles(rtfhe_fft_plan *plan, const double *ifft_table, const double *fft_table);
int rtfhe_fft_plan_ifft_i32(rtfhe_fft_plan *plan, const int32_t *src, double *res, size_t count);    /* execute_reverse_int / _torus32 */
int rtfhe_fft_plan_ifft_f64(rtfhe_fft_plan *plan, const double *src, double *res, size_t count);     /* execute_reverse */
int rtfhe_fft_plan_fft_u32(rtfhe_fft_plan *plan, const double *src, uint32_t *res, size_t count);    /* execute_direct_torus32 */
int rtfhe_fft_plan_fft_f64(rtfhe_fft_plan *plan, const double *src, double *res, size_t count);      /* execute_direct */
int rtfhe_fft_plan_poly_mul(rtfhe_fft_plan *plan, const uint32_t *a, const uint32_t *b, uint32_t *res, size_t count);   /* spqlios-wrapper.cpp:38-53 */

/* ---- key generation / encryption (host side) ----
 * Production entry points draw every key bit, mask and noise sample from the OS CSPRNG (getrandom(2), expanded with ChaCha20),
 * as the reference draws from rand::thread_rng (utils/src/math.rs:417-479).  They fail with RTFHE_ERR_STATE if the OS gives
 * no entropy.  rtfhe_keygen_with_keys is TFHE::new for caller-supplied secret keys (hom_nand/src/tfhe.rs:21-25). */
int rtfhe_keygen(const rtfhe_params *p, int32_t *key0 /* [n] */, int32_t *key1 /* [N] */,
                 uint32_t *bk /* [n][2][2l][N] */, uint32_t *ksk /* [N][t][base-1][n+1] */);
int rtfhe_keygen_with_keys(const rtfhe_params *p, const int32_t *key0, const int32_t *key1, uint32_t *bk, uint32_t *ksk);
int rtfhe_tlwe_encrypt_bits(const rtfhe_params *p, const int32_t *key0, const uint8_t *bits, uint32_t *out /* [count][n+1] */,
                            size_t count);
/* the reference's KeySwitchingKey shape from the compact one: entries t = 1 .. base-1 of every level copied, entry t = base =
 * TLWE(base * s_i / 2^(basebit (l+1))) freshly encrypted as KeySwitchingKey::new fills it (hom_nand/src/tlwe.rs:252-274) */
int rtfhe_ksk_expand_ref(const rtfhe_params *p, const int32_t *key0, const int32_t *key1,
                         const uint32_t *ksk /* [N][t][base-1][n+1] */, uint32_t *ksk_ref /* [N][t][base][n+1] */);
/* TEST ONLY -- NOT SECURE: the same, reproducible from a 64-bit seed expanded through xoshiro256** (not a CSPRNG: whoever knows
 * the seed regenerates every mask and noise sample and recovers the secret key from the key-switching key; a reused seed
 * reuses mask and noise).  For fixtures, parity tests and benchmarks only; never for keys or ciphertexts that protect data. */
int rtfhe_keygen_deterministic(const rtfhe_params *p, uint64_t seed, int32_t *key0, int32_t *key1, uint32_t *bk, uint32_t *ksk);
int rtfhe_keygen_with_keys_deterministic(const rtfhe_params *p, uint64_t seed, const int32_t *key0, const int32_t *key1,
                                         uint32_t *bk, uint32_t *ksk);
int rtfhe_ksk_expand_ref_deterministic(const rtfhe_params *p, uint64_t seed, const int32_t *key0, const int32_t *key1,
                                       const uint32_t *ksk, uint32_t *ksk_ref);
int rtfhe_tlwe_encrypt_bits_deterministic(const rtfhe_params *p, const int32_t *key0, uint64_t seed,
                                          const uint8_t *bits, uint32_t *out /* [count][n+1] */, size_t count);
int rtfhe_tlwe_decrypt_bits(const rtfhe_params *p, const int32_t *key0, const uint32_t *in,
                            uint8_t *bits, size_t count);
int rtfhe_tlwe_phase(const rtfhe_params *p, const int32_t *key0, const uint32_t *in, uint32_t *phase, size_t count);

/* ---- wire format (flat little-endian files with an FNV-1a checksum; the reference has no serialization) ----
 * key file   "RTFHEKY1" | rtfhe_params | u32 flags (1 bk, 2 ksk, 4 secret keys) | u32 0 | [key0 | key1] | [bk] | [ksk] | u64 fnv
 * batch file "RTFHECT1" | i32 n | i32 0 | u64 count | u32[count][n+1] | u64 fnv */
int rtfhe_keys_write(const char *path, const rtfhe_params *p, const int32_t *key0, const int32_t *key1,
                     const uint32_t *bk, const uint32_t *ksk);          /* null sections are left out */
int rtfhe_keys_read_header(const char *path, rtfhe_params *p, uint32_t *flags);
/* Buffers are sized for the parameter set rtfhe_keys_read_header returns (call it first); null = skip that section; asking for a
 * section the file does not hold (see flags) fails. */
int rtfhe_keys_read(const char *path, int32_t *key0, int32_t *key1, uint32_t *bk, uint32_t *ksk);
int rtfhe_tlwe_write(const char *path, int32_t n, const uint32_t *cts, size_t count);
int rtfhe_tlwe_read(const char *path, int32_t *n, uint64_t *count, uint32_t *cts /* NULL: header only */, size_t capacity);

#ifdef __cplusplus
}
#endif
#endif
