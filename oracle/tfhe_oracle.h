/*
 * tfhe_oracle.h -- CPU ORACLE for the HomNAND hot path of hideki1217/rusTfhe.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (rustfhe_amd/, the
 * C-ABI library librtfhe_hip.so) never links, imports or calls anything in oracle/.
 *
 * It is a plain-C restatement of the reference's algorithm; every function cites the
 * reference file:line it follows (paths relative to the reference repo root).
 *
 * Parity status: PINNED.
 *   - the FP64 transforms are checked bit-for-bit against the reference's own compiled
 *     spqlios (oracle/_ref/libspqlios_ref.so, built by oracle/Makefile from the sources
 *     where they lie under /root/reference) -- tests/test_oracle_vs_ref.py;
 *   - the integer glue is checked against every deterministic known-answer test the
 *     reference holds for this path (utils/src/math.rs:75-84,761-903,1207-1273,
 *     utils/src/spqlios.rs:243-276, hom_nand/src/tlwe.rs:302-326) -- tests/test_oracle_kat.py;
 *   - whole-gate vectors generated here with the reference FFT plugged in are committed
 *     under tests/golden/ (scripts/gen_golden.py).
 *
 * Flat little-endian layouts (the reference has none):
 *   TLWE  lvl0 : u32[n+1]      = a[0..n), b
 *   TLWE  lvl1 : u32[N+1]      = a'[0..N), b'
 *   TRLWE      : u32[2][N]     = b(X), a(X)
 *   BK torus   : u32[n][2][2l][N]   comp 0 = `cipher` (b rows), comp 1 = `p_key` (a rows)
 *   BK fft     : f64[n][2][2l][N]   same order, each poly an FrrSeries (Re[0..N/2) | Im[0..N/2))
 *   KSK        : u32[N][t][base-1][n+1]
 */
#ifndef TFHE_ORACLE_H
#define TFHE_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int32_t n;          /* TLWE lvl0 dimension          hom_nand/src/tlwe.rs:175  (635)  */
    int32_t N;          /* TRLWE degree                 hom_nand/src/trlwe.rs:76  (1024) */
    int32_t nbit;       /* log2(N)                      hom_nand/src/tfhe.rs:16   (10)   */
    int32_t l;          /* gadget levels                hom_nand/src/trgsw.rs:115 (3)    */
    int32_t bgbit;      /* gadget base bits             hom_nand/src/trgsw.rs:112 (6)    */
    int32_t ks_t;       /* key-switch levels IKS_L      hom_nand/src/tlwe.rs:178  (8)    */
    int32_t ks_basebit; /* key-switch base bits         hom_nand/src/tlwe.rs:179  (2)    */
} orc_params;

enum { ORC_NAND = 0, ORC_AND = 1, ORC_OR = 2, ORC_XOR = 3, ORC_NOT = 4, ORC_COPY = 5, ORC_ANDNY = 6 };
enum { ORC_BACKEND_FFT64_MIRROR = 0, ORC_BACKEND_EXACT_INT = 1, ORC_BACKEND_HOOK = 2 };

void orc_default_params(orc_params *p);

/* ---------------------------------------------------------------- transforms */
typedef struct orc_plan orc_plan;

/* hooks with the signature of the reference FFI (utils/src/spqlios.rs:18-32):
 *   fwd = Spqlios_ifft_i32(handle, res, src), inv = Spqlios_fft_u32(handle, res, src) */
typedef void (*orc_fwd_hook)(void *handle, double *res, const int32_t *src);
typedef void (*orc_inv_hook)(void *handle, uint32_t *res, const double *src);

orc_plan *orc_plan_new(int32_t N);                 /* tables from libm, spqlios-fft-impl.cpp:158-193,400-437 */
void orc_plan_free(orc_plan *pl);
void orc_plan_set_backend(orc_plan *pl, int backend);
void orc_plan_set_hooks(orc_plan *pl, void *handle, orc_fwd_hook fwd, orc_inv_hook inv);
/* tables in the reference's memory layout (blocks of 4 cos | 4 sin), 2N doubles each;
 * only the first 2N-8 entries are ever written by the reference */
void orc_plan_export_tables(const orc_plan *pl, double *ifft_table, double *fft_table);
void orc_plan_import_tables(orc_plan *pl, const double *ifft_table, const double *fft_table);

/* execute_reverse_int / execute_reverse_torus32, fft_processor_spqlios.cpp:58-106 */
void orc_ifft_i32(orc_plan *pl, double *res, const int32_t *src);
/* execute_reverse, fft_processor_spqlios.cpp:16-55 */
void orc_ifft_f64(orc_plan *pl, double *res, const double *src);
/* execute_direct_torus32, fft_processor_spqlios.cpp:156-183 */
void orc_fft_u32(orc_plan *pl, uint32_t *res, const double *src);
/* execute_direct, fft_processor_spqlios.cpp:108-153 */
void orc_fft_f64(orc_plan *pl, double *res, const double *src);
/* Spqlios_poly_mul, spqlios-wrapper.cpp:38-53 */
void orc_poly_mul(orc_plan *pl, uint32_t *res, const uint32_t *a, const uint32_t *b);
/* FrrSeries::hadamard, utils/src/spqlios.rs:204-222 */
void orc_hadamard(int32_t N, double *res, const double *l, const double *r);

/* ---------------------------------------------------------------- integer glue */
uint32_t orc_torus_from_f32(float v);                               /* math.rs:691-696 */
uint32_t orc_make_decomp_mask(uint32_t l, uint32_t bits);           /* math.rs:542-560 */
uint32_t orc_inline_decomp_mask(uint32_t l, uint32_t bits);         /* math.rs:581-593 */
void orc_decomp_scalar(uint32_t x, uint32_t bits, uint32_t mask, int32_t l, int32_t *out);   /* math.rs:561-577 */
void orc_decomp_u32_scalar(uint32_t x, uint32_t bits, int32_t l, uint32_t *out);              /* math.rs:598-616 */
void orc_decomp_poly(int32_t N, const uint32_t *p, uint32_t bits, uint32_t mask, int32_t l,
                     int32_t *out /* [l][N] */);                    /* math.rs:300-326 */
void orc_rotate_u32(int32_t N, const uint32_t *p, int32_t n, uint32_t *out);   /* math.rs:85-132 */
void orc_rotate_i32(int32_t N, const int32_t *p, int32_t n, int32_t *out);
/* exact negacyclic product, Polynomial::cross math.rs:238-257 + convolution :713-723;
 * res = a (*) b  mod X^N+1 mod 2^32 */
void orc_negacyclic_mul_u32(int32_t N, const uint32_t *a, const int32_t *b, uint32_t *res);

/* ---------------------------------------------------------------- scheme */
/* TRGSWRepF::from, trgsw.rs:68-76: torus rows -> FrrSeries rows; count polys */
void orc_trgsw_to_fft(orc_plan *pl, const uint32_t *rows, double *rows_f, size_t count);
/* Cross for TRGSWRepF, trgsw.rs:264-306 */
void orc_external_product(const orc_params *p, orc_plan *pl, const double *trgsw_f /* [2][2l][N] */,
                          const uint32_t *trgsw_t /* torus form, exact backend only, may be NULL */,
                          const uint32_t *trlwe /* [2][N] */, uint32_t *out /* [2][N] */);
/* TRGSWRepF::cmux, trgsw.rs:319-321 */
void orc_cmux(const orc_params *p, orc_plan *pl, const double *trgsw_f, const uint32_t *trgsw_t,
              const uint32_t *rep1, const uint32_t *rep0, uint32_t *out);
/* TFHE::blind_rotate with the gate test vector, tfhe.rs:81-113; steps <= n allows prefixes */
void orc_blind_rotate(const orc_params *p, orc_plan *pl, const double *bk_f, const uint32_t *bk_t,
                      const uint32_t *tlwe /* [n+1] */, int32_t steps, uint32_t *acc /* [2][N] */);
/* TRLWERep::sample_extract_index, trlwe.rs:110-121 */
void orc_sample_extract(int32_t N, const uint32_t *trlwe, int32_t index, uint32_t *tlwe1 /* [N+1] */);
/* TLWERep::identity_key_switch, tlwe.rs:43-73 */
void orc_key_switch(const orc_params *p, const uint32_t *ksk, const uint32_t *tlwe1, uint32_t *out);
/* the same, key in the reference's container shape u32[N][t][base][n+1] (tlwe.rs:243-245, get(i,l,t) = [i][l][t-1]) */
void orc_key_switch_ref(const orc_params *p, const uint32_t *ksk_ref, const uint32_t *tlwe1, uint32_t *out);
/* gate pre-steps, tfhe.rs:27-71 */
void orc_gate_linear(const orc_params *p, int op, const uint32_t *in0, const uint32_t *in1, uint32_t *t);
/* TFHE::bootstrap, tfhe.rs:73-88 */
void orc_bootstrap(const orc_params *p, orc_plan *pl, const double *bk_f, const uint32_t *bk_t,
                   const uint32_t *ksk, const uint32_t *t, uint32_t *out);
void orc_gate(const orc_params *p, orc_plan *pl, int op, const double *bk_f, const uint32_t *bk_t,
              const uint32_t *ksk, const uint32_t *in0, const uint32_t *in1, uint32_t *out);
/* hom_mux, tfhe.rs:27-40 */
void orc_mux(const orc_params *p, orc_plan *pl, const double *bk_f, const uint32_t *bk_t,
             const uint32_t *ksk, const uint32_t *c, const uint32_t *in0, const uint32_t *in1, uint32_t *out);
/* nthreads independent gate streams, each with its own plan (the reference's thread model
 * would be thread_local! FFT_MAP, math.rs:349-351).  Returns wall seconds. */
void orc_set_mt_hooks(void *(*new_fn)(int32_t), orc_fwd_hook fwd, orc_inv_hook inv);   /* backend ORC_BACKEND_HOOK in the call below */
double orc_gate_batch_mt(const orc_params *p, int backend, int op, const double *bk_f, const uint32_t *bk_t,
                         const uint32_t *ksk, const uint32_t *in0, const uint32_t *in1, uint32_t *out,
                         size_t count, int nthreads);
/* bench.py's all-core baseline: `count` gates over `in_count` distinct inputs (gate g takes input g % in_count, writes out[g]), threads
 * optionally pinned (cpus[t] >= 0), keys replicated per memory node and first-touched by a thread of that node (node_of_thread[t] < nnodes).
 * FP64-mirror / hook backends only (bk_f).  Returns the timed seconds, -1.0 on failure. */
double orc_gate_batch_mt_numa(const orc_params *p, int backend, int op, const double *bk_f, const uint32_t *ksk,
                              const uint32_t *in0, const uint32_t *in1, size_t in_count, uint32_t *out, size_t count,
                              int nthreads, const int *cpus, const int *node_of_thread, int nnodes);

/* ---------------------------------------------------------------- keys / encryption (own seeded RNG) */
typedef struct { uint64_t s[4]; } orc_rng;
void orc_rng_seed(orc_rng *r, uint64_t seed);
uint64_t orc_rng_next(orc_rng *r);
uint32_t orc_rng_uniform_torus(orc_rng *r);          /* math.rs:425-432: torus!(Uniform f32 [0,1)) */
uint32_t orc_rng_gaussian_torus(orc_rng *r, float alpha);   /* math.rs:417-424: torus!(Normal f32) */
void orc_gen_binary_key(orc_rng *r, int32_t len, int32_t *key);     /* math.rs:471-479 */
/* Crypto<Torus32> for TLWE, tlwe.rs:213-241 */
void orc_tlwe_encrypt(orc_rng *r, int32_t n, const int32_t *key, uint32_t msg, float alpha, uint32_t *ct);
uint32_t orc_tlwe_phase(int32_t n, const int32_t *key, const uint32_t *ct);
int orc_torus2binary(uint32_t t);                    /* tlwe.rs:187-194 */
uint32_t orc_binary2torus(int bit);                  /* tlwe.rs:181-186 */
/* Crypto<Polynomial<Torus32>> for TRLWE, trlwe.rs:127-147 */
void orc_trlwe_encrypt(orc_rng *r, orc_plan *pl, int32_t N, const int32_t *key, const uint32_t *msg,
                       float alpha, uint32_t *ct /* [2][N] */);
void orc_trlwe_phase(orc_plan *pl, int32_t N, const int32_t *key, const uint32_t *ct, uint32_t *phase);
/* Crypto<i32> for TRGSW, trgsw.rs:217-229 (+ :118-138) */
void orc_trgsw_encrypt(orc_rng *r, orc_plan *pl, const orc_params *p, const int32_t *key, int32_t mu,
                       float alpha, uint32_t *ct /* [2][2l][N] */);
/* BootstrappingKey::new, tfhe.rs:119-126 (torus form; convert with orc_trgsw_to_fft) */
void orc_bk_gen(orc_rng *r, orc_plan *pl, const orc_params *p, const int32_t *key0, const int32_t *key1,
                float alpha, uint32_t *bk_t /* [n][2][2l][N] */);
/* KeySwitchingKey::new, tlwe.rs:247-277 */
void orc_ksk_gen(orc_rng *r, const orc_params *p, const int32_t *key1, const int32_t *key0, float alpha,
                 uint32_t *ksk /* [N][t][base-1][n+1] */);

void orc_ksk_expand_ref(orc_rng *r, const orc_params *p, const int32_t *key1, const int32_t *key0, float alpha,
                        const uint32_t *ksk /* [N][t][base-1][n+1] */, uint32_t *ksk_ref /* [N][t][base][n+1] */);
uint64_t orc_fnv64(const void *data, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif
