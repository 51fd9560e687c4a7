/*
 * tfhe_oracle.c -- CPU ORACLE (test infrastructure, NOT product code; see tfhe_oracle.h).
 *
 * Plain-C restatement of the reference HomNAND path.  Compile with -ffp-contract=off:
 * the reference's AVX code uses separate vmulpd/vaddpd/vsubpd (no FMA) and every
 * product and sum below must be rounded individually to reproduce its bits
 * (utils/src/spqlios/spqlios-fft-avx.s:203-222, spqlios-ifft-avx.s:130-149).
 */
#define _GNU_SOURCE
#include "tfhe_oracle.h"

#include <math.h>
#include <pthread.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

void orc_default_params(orc_params *p) {
    p->n = 635; p->N = 1024; p->nbit = 10; p->l = 3; p->bgbit = 6; p->ks_t = 8; p->ks_basebit = 2;
}

/* ================================================================= transforms */

struct orc_plan {
    int32_t N;        /* polynomial degree; the transform works on N/2 complex points */
    int32_t backend;
    /* per-stage twiddles, natural index k (cos[k], sin[k]) */
    double *tw_twist_c, *tw_twist_s;     /* forward twist,  angle +2*pi*j/(2N), j < N/2  */
    double *tw_fwd_c, *tw_fwd_s;         /* forward stages, concatenated halfnn = N/4 ... 4 */
    double *tw_inv_c, *tw_inv_s;         /* inverse stages, concatenated halfnn = 4 ... N/4 */
    double *tw_untw_c, *tw_untw_s;       /* inverse untwist, angle -2*pi*j/(2N) */
    double *re, *im;                     /* scratch N/2 each */
    double *scratch;                     /* 4N doubles for poly_mul / external product */
    void *hook_handle; orc_fwd_hook hook_fwd; orc_inv_hook hook_inv;
};

/* accurate_cos / accurate_sin, spqlios-fft-impl.cpp:99-113 */
static double accurate_cos(int32_t i, int32_t n) {
    i = ((i % n) + n) % n;
    if (i >= 3 * n / 4) return cos(2. * M_PI * (n - i) / (double)n);
    if (i >= 2 * n / 4) return -cos(2. * M_PI * (i - n / 2) / (double)n);
    if (i >= 1 * n / 4) return -cos(2. * M_PI * (n / 2 - i) / (double)n);
    return cos(2. * M_PI * (i) / (double)n);
}
static double accurate_sin(int32_t i, int32_t n) {
    i = ((i % n) + n) % n;
    if (i >= 3 * n / 4) return -sin(2. * M_PI * (n - i) / (double)n);
    if (i >= 2 * n / 4) return -sin(2. * M_PI * (i - n / 2) / (double)n);
    if (i >= 1 * n / 4) return sin(2. * M_PI * (n / 2 - i) / (double)n);
    return sin(2. * M_PI * (i) / (double)n);
}

static double *dalloc(size_t cnt) {
    void *p = NULL;
    if (posix_memalign(&p, 64, (cnt ? cnt : 1) * sizeof(double)) != 0) abort();
    memset(p, 0, (cnt ? cnt : 1) * sizeof(double));
    return (double *)p;
}

/* Same values as new_ifft_table / new_fft_table (spqlios-fft-impl.cpp:400-437, 158-193),
 * stored per stage in natural order instead of 4-cos/4-sin blocks. */
orc_plan *orc_plan_new(int32_t N) {
    if (N < 16 || (N & (N - 1))) return NULL;
    orc_plan *pl = (orc_plan *)calloc(1, sizeof(orc_plan));
    const int32_t n = 2 * N, ns4 = N / 2;
    pl->N = N;
    pl->backend = ORC_BACKEND_FFT64_MIRROR;
    pl->tw_twist_c = dalloc(ns4); pl->tw_twist_s = dalloc(ns4);
    pl->tw_untw_c = dalloc(ns4);  pl->tw_untw_s = dalloc(ns4);
    pl->tw_fwd_c = dalloc(ns4);   pl->tw_fwd_s = dalloc(ns4);
    pl->tw_inv_c = dalloc(ns4);   pl->tw_inv_s = dalloc(ns4);
    pl->re = dalloc(ns4); pl->im = dalloc(ns4);
    pl->scratch = dalloc(4 * (size_t)N);
    for (int32_t j = 0; j < ns4; j++) {
        pl->tw_twist_c[j] = accurate_cos(j, n);
        pl->tw_twist_s[j] = accurate_sin(j, n);
        pl->tw_untw_c[j] = accurate_cos(-j, n);
        pl->tw_untw_s[j] = accurate_sin(-j, n);
    }
    size_t o = 0;
    for (int32_t nn = ns4; nn >= 8; nn /= 2) {          /* forward: new_ifft_table :424-435 */
        int32_t halfnn = nn / 2, j = n / nn;
        for (int32_t k = 0; k < halfnn; k++) {
            pl->tw_fwd_c[o + k] = accurate_cos(j * k, n);
            pl->tw_fwd_s[o + k] = accurate_sin(j * k, n);
        }
        o += halfnn;
    }
    o = 0;
    for (int32_t halfnn = 4; halfnn < ns4; halfnn *= 2) {   /* inverse: new_fft_table :173-184 */
        int32_t nn = 2 * halfnn, j = n / nn;
        for (int32_t k = 0; k < halfnn; k++) {
            pl->tw_inv_c[o + k] = accurate_cos(-j * k, n);
            pl->tw_inv_s[o + k] = accurate_sin(-j * k, n);
        }
        o += halfnn;
    }
    return pl;
}

void orc_plan_free(orc_plan *pl) {
    if (!pl) return;
    free(pl->tw_twist_c); free(pl->tw_twist_s); free(pl->tw_untw_c); free(pl->tw_untw_s);
    free(pl->tw_fwd_c); free(pl->tw_fwd_s); free(pl->tw_inv_c); free(pl->tw_inv_s);
    free(pl->re); free(pl->im); free(pl->scratch);
    free(pl);
}

void orc_plan_set_backend(orc_plan *pl, int backend) { pl->backend = backend; }
void orc_plan_set_hooks(orc_plan *pl, void *handle, orc_fwd_hook fwd, orc_inv_hook inv) {
    pl->hook_handle = handle; pl->hook_fwd = fwd; pl->hook_inv = inv;
    pl->backend = ORC_BACKEND_HOOK;
}

/* reference memory layout: for each stage, blocks |c0 c1 c2 c3|s0 s1 s2 s3| */
static size_t put_blocks(double *dst, const double *c, const double *s, int32_t cnt) {
    size_t w = 0;
    for (int32_t i = 0; i < cnt; i += 4) {
        for (int k = 0; k < 4; k++) dst[w++] = c[i + k];
        for (int k = 0; k < 4; k++) dst[w++] = s[i + k];
    }
    return w;
}
static size_t get_blocks(const double *src, double *c, double *s, int32_t cnt) {
    size_t r = 0;
    for (int32_t i = 0; i < cnt; i += 4) {
        for (int k = 0; k < 4; k++) c[i + k] = src[r++];
        for (int k = 0; k < 4; k++) s[i + k] = src[r++];
    }
    return r;
}
void orc_plan_export_tables(const orc_plan *pl, double *ifft_table, double *fft_table) {
    const int32_t ns4 = pl->N / 2;
    memset(ifft_table, 0, sizeof(double) * 2 * pl->N);
    memset(fft_table, 0, sizeof(double) * 2 * pl->N);
    size_t w = put_blocks(ifft_table, pl->tw_twist_c, pl->tw_twist_s, ns4), o = 0;
    for (int32_t nn = ns4; nn >= 8; nn /= 2) { w += put_blocks(ifft_table + w, pl->tw_fwd_c + o, pl->tw_fwd_s + o, nn / 2); o += nn / 2; }
    w = 0; o = 0;
    for (int32_t h = 4; h < ns4; h *= 2) { w += put_blocks(fft_table + w, pl->tw_inv_c + o, pl->tw_inv_s + o, h); o += h; }
    put_blocks(fft_table + w, pl->tw_untw_c, pl->tw_untw_s, ns4);
}
void orc_plan_import_tables(orc_plan *pl, const double *ifft_table, const double *fft_table) {
    const int32_t ns4 = pl->N / 2;
    size_t r = get_blocks(ifft_table, pl->tw_twist_c, pl->tw_twist_s, ns4), o = 0;
    for (int32_t nn = ns4; nn >= 8; nn /= 2) { r += get_blocks(ifft_table + r, pl->tw_fwd_c + o, pl->tw_fwd_s + o, nn / 2); o += nn / 2; }
    r = 0; o = 0;
    for (int32_t h = 4; h < ns4; h *= 2) { r += get_blocks(fft_table + r, pl->tw_inv_c + o, pl->tw_inv_s + o, h); o += h; }
    get_blocks(fft_table + r, pl->tw_untw_c, pl->tw_untw_s, ns4);
}

/* Forward transform on pl->re / pl->im in place.  Follows ifft_model,
 * spqlios-fft-impl.cpp:469-641 (= asm `ifft`, spqlios-ifft-avx.s:64-272). */
static void mirror_forward(orc_plan *pl) {
    const int32_t ns4 = pl->N / 2;
    double *restrict re = pl->re, *restrict im = pl->im;
    {   /* multiply by omega^j  (:496-518) */
        const double *restrict c = pl->tw_twist_c, *restrict s = pl->tw_twist_s;
        for (int32_t j = 0; j < ns4; j++) {
            double rc = re[j] * c[j], ic = im[j] * c[j], rs = re[j] * s[j], is = im[j] * s[j];
            re[j] = rc - is;
            im[j] = ic + rs;
        }
    }
    size_t o = 0;
    for (int32_t nn = ns4; nn >= 8; nn /= 2) {          /* (:526-572) */
        const int32_t halfnn = nn / 2;
        const double *restrict c = pl->tw_fwd_c + o, *restrict s = pl->tw_fwd_s + o;
        for (int32_t block = 0; block < ns4; block += nn) {
            double *restrict r0 = re + block, *restrict i0 = im + block;
            double *restrict r1 = re + block + halfnn, *restrict i1 = im + block + halfnn;
            for (int32_t k = 0; k < halfnn; k++) {
                double sr = r0[k] + r1[k], si = i0[k] + i1[k];
                double dr = r0[k] - r1[k], di = i0[k] - i1[k];
                r0[k] = sr; i0[k] = si;
                double a = dr * c[k], b = di * s[k];
                r1[k] = a - b;
                a = dr * s[k]; b = di * c[k];
                i1[k] = a + b;
            }
        }
        o += halfnn;
    }
    for (int32_t b = 0; b < ns4; b += 4) {              /* size 4 (:575-603) */
        double r0 = re[b], r1 = re[b + 1], r2 = re[b + 2], r3 = re[b + 3];
        double j0 = im[b], j1 = im[b + 1], j2 = im[b + 2], j3 = im[b + 3];
        re[b] = r0 + r2;  re[b + 1] = r1 + r3;  re[b + 2] = r0 + (-r2);  re[b + 3] = (-j1) + j3;
        im[b] = j0 + j2;  im[b + 1] = j1 + j3;  im[b + 2] = j0 + (-j2);  im[b + 3] = r1 + (-r3);
    }
    for (int32_t b = 0; b < ns4; b += 2) {              /* size 2 (:606-634) */
        double r0 = re[b], r1 = re[b + 1], j0 = im[b], j1 = im[b + 1];
        re[b] = r0 + r1; re[b + 1] = r0 + (-r1);
        im[b] = j0 + j1; im[b + 1] = j0 + (-j1);
    }
}

/* Inverse transform on pl->re / pl->im in place.  Follows fft_model,
 * spqlios-fft-impl.cpp:204-397 (= asm `fft`, spqlios-fft-avx.s:79-280). */
static void mirror_inverse(orc_plan *pl) {
    const int32_t ns4 = pl->N / 2;
    double *restrict re = pl->re, *restrict im = pl->im;
    for (int32_t b = 0; b < ns4; b += 2) {              /* size 2 (:248-269) */
        double r0 = re[b], r1 = re[b + 1], j0 = im[b], j1 = im[b + 1];
        re[b] = r0 + r1; re[b + 1] = r0 + (-r1);
        im[b] = j0 + j1; im[b + 1] = j0 + (-j1);
    }
    for (int32_t b = 0; b < ns4; b += 4) {              /* size 4 (:289-310) */
        double r0 = re[b], r1 = re[b + 1], r2 = re[b + 2], r3 = re[b + 3];
        double j0 = im[b], j1 = im[b + 1], j2 = im[b + 2], j3 = im[b + 3];
        re[b] = r0 + r2;  re[b + 1] = r1 + j3;     re[b + 2] = r0 + (-r2);  re[b + 3] = r1 + (-j3);
        im[b] = j0 + j2;  im[b + 1] = j1 + (-r3);  im[b + 2] = j0 + (-j2);  im[b + 3] = j1 + r3;
    }
    size_t o = 0;
    for (int32_t halfnn = 4; halfnn < ns4; halfnn *= 2) {   /* (:315-363) */
        const int32_t nn = 2 * halfnn;
        const double *restrict c = pl->tw_inv_c + o, *restrict s = pl->tw_inv_s + o;
        for (int32_t block = 0; block < ns4; block += nn) {
            double *restrict r0 = re + block, *restrict i0 = im + block;
            double *restrict r1 = re + block + halfnn, *restrict i1 = im + block + halfnn;
            for (int32_t k = 0; k < halfnn; k++) {
                double t0 = r1[k] * c[k], t1 = r1[k] * s[k], t2 = i1[k] * c[k], t3 = i1[k] * s[k];
                double tr = t0 - t3, ti = t1 + t2;
                double ar = r0[k], ai = i0[k];
                r0[k] = ar + tr; i0[k] = ai + ti;
                r1[k] = ar - tr; i1[k] = ai - ti;
            }
        }
        o += halfnn;
    }
    {   /* multiply by omb^j (:374-396) */
        const double *restrict c = pl->tw_untw_c, *restrict s = pl->tw_untw_s;
        for (int32_t j = 0; j < ns4; j++) {
            double rc = re[j] * c[j], ic = im[j] * c[j], rs = re[j] * s[j], is = im[j] * s[j];
            re[j] = rc - is;
            im[j] = ic + rs;
        }
    }
}

void orc_ifft_i32(orc_plan *pl, double *res, const int32_t *src) {
    const int32_t N = pl->N, ns4 = N / 2;
    if (pl->backend == ORC_BACKEND_HOOK) { pl->hook_fwd(pl->hook_handle, res, src); return; }
    for (int32_t i = 0; i < ns4; i++) { pl->re[i] = (double)src[i]; pl->im[i] = (double)src[i + ns4]; }
    mirror_forward(pl);
    memcpy(res, pl->re, sizeof(double) * ns4);
    memcpy(res + ns4, pl->im, sizeof(double) * ns4);
}

void orc_ifft_f64(orc_plan *pl, double *res, const double *src) {
    const int32_t ns4 = pl->N / 2;
    memcpy(pl->re, src, sizeof(double) * ns4);
    memcpy(pl->im, src + ns4, sizeof(double) * ns4);
    mirror_forward(pl);
    memcpy(res, pl->re, sizeof(double) * ns4);
    memcpy(res + ns4, pl->im, sizeof(double) * ns4);
}

static void load_scaled(orc_plan *pl, const double *src) {
    const int32_t N = pl->N, ns4 = N / 2;
    const double _2sN = (double)2 / (double)N;            /* fft_processor_spqlios.cpp:158 */
    for (int32_t i = 0; i < ns4; i++) { pl->re[i] = src[i] * _2sN; pl->im[i] = src[i + ns4] * _2sN; }
}

void orc_fft_u32(orc_plan *pl, uint32_t *res, const double *src) {
    const int32_t ns4 = pl->N / 2;
    if (pl->backend == ORC_BACKEND_HOOK) { pl->hook_inv(pl->hook_handle, res, src); return; }
    load_scaled(pl, src);
    mirror_inverse(pl);
    /* Torus32(int64_t(x)): truncation toward zero, then wrap to 32 bits (:182) */
    for (int32_t i = 0; i < ns4; i++) {
        res[i] = (uint32_t)(int64_t)pl->re[i];
        res[i + ns4] = (uint32_t)(int64_t)pl->im[i];
    }
}

void orc_fft_f64(orc_plan *pl, double *res, const double *src) {
    const int32_t ns4 = pl->N / 2;
    load_scaled(pl, src);
    mirror_inverse(pl);
    memcpy(res, pl->re, sizeof(double) * ns4);
    memcpy(res + ns4, pl->im, sizeof(double) * ns4);
}

void orc_hadamard(int32_t N, double *res, const double *l, const double *r) {
    const int32_t h = N / 2;
    for (int32_t i = 0; i < h; i++) {
        double ii = l[i + h] * r[i + h];
        double rr = l[i] * r[i];
        double ri = l[i] * r[i + h];
        double ir = l[i + h] * r[i];
        res[i] = rr - ii;
        res[i + h] = ir + ri;
    }
}

void orc_poly_mul(orc_plan *pl, uint32_t *res, const uint32_t *a, const uint32_t *b) {
    const int32_t N = pl->N, h = N / 2;
    double *ta = pl->scratch, *tb = pl->scratch + N;
    orc_ifft_i32(pl, ta, (const int32_t *)a);
    orc_ifft_i32(pl, tb, (const int32_t *)b);
    for (int32_t i = 0; i < h; i++) {                     /* spqlios-wrapper.cpp:45-50 */
        double aimbim = ta[i + h] * tb[i + h];
        double arebim = ta[i] * tb[i + h];
        double p = ta[i] * tb[i];
        double q = ta[i + h] * tb[i];
        ta[i] = p - aimbim;
        ta[i + h] = q + arebim;
    }
    orc_fft_u32(pl, res, ta);
}

/* ================================================================= integer glue */

uint32_t orc_torus_from_f32(float v) {
    const float X = 4294967296.0f;                        /* u32::MAX as f32 rounds to 2^32 */
    volatile float w = v - floorf(v);
    volatile float fr = w - truncf(w);                    /* f32::fract */
    volatile float x = fr * X;
    if (!(x > 0.0f)) return 0u;                           /* Rust `as u32` saturates, NaN -> 0 */
    if (x >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)x;
}

uint32_t orc_make_decomp_mask(uint32_t l, uint32_t bits) {
    const uint32_t total = 32;
    uint32_t u = 0;
    if ((total - l * bits) != 0) {
        u = u + (1u << (total - l * bits - 1));
        for (uint32_t i = l; i >= 1; i--) u += 1u << (total - i * bits - 1);
    } else {
        for (uint32_t i = l - 1; i >= 1; i--) u += 1u << (total - i * bits - 1);
    }
    return u;
}

uint32_t orc_inline_decomp_mask(uint32_t l, uint32_t bits) {
    const uint32_t total = 32;
    uint32_t u = 0;
    if ((total - l * bits) != 0) { for (uint32_t i = 1; i <= l; i++) u |= 1u << (total - i * bits - 1); }
    else { for (uint32_t i = 1; i < l; i++) u |= 1u << (total - i * bits - 1); }
    return u;
}

static inline int32_t decomp_digit(uint32_t u, uint32_t bits, int32_t i) {
    const uint32_t mask = (1u << bits) - 1;
    uint32_t v = (u >> (32 - bits * (uint32_t)(i + 1))) & mask;
    return (int32_t)((v & (1u << (bits - 1))) * 0xfffffffeu + v);
}

void orc_decomp_scalar(uint32_t x, uint32_t bits, uint32_t mask, int32_t l, int32_t *out) {
    uint32_t u = (x + mask) ^ mask;
    for (int32_t i = 0; i < l; i++) out[i] = decomp_digit(u, bits, i);
}

void orc_decomp_u32_scalar(uint32_t x, uint32_t bits, int32_t l, uint32_t *out) {
    const uint32_t total = 32;
    uint32_t u = x + (((total - (uint32_t)l * bits) != 0) ? (1u << (total - (uint32_t)l * bits - 1)) : 0u);
    const uint32_t mask = (1u << bits) - 1;
    for (int32_t i = 0; i < l; i++) out[i] = (u >> (total - bits * (uint32_t)(i + 1))) & mask;
}

void orc_decomp_poly(int32_t N, const uint32_t *p, uint32_t bits, uint32_t mask, int32_t l, int32_t *out) {
    for (int32_t i = 0; i < l; i++)
        for (int32_t k = 0; k < N; k++) {
            uint32_t u = (p[k] + mask) ^ mask;
            out[(size_t)i * N + k] = decomp_digit(u, bits, i);
        }
}

static inline int32_t mod_floor(int32_t a, int32_t m) { int32_t r = a % m; return (r < 0) ? r + m : r; }

void orc_rotate_u32(int32_t N, const uint32_t *p, int32_t n, uint32_t *out) {
    int32_t r = mod_floor(n, 2 * N);
    if (r <= N) {
        for (int32_t i = 0; i < r; i++) out[i] = 0u - p[N - r + i];
        for (int32_t i = r; i < N; i++) out[i] = p[i - r];
    } else {
        int32_t q = r - N;
        for (int32_t i = 0; i < q; i++) out[i] = p[N - q + i];
        for (int32_t i = q; i < N; i++) out[i] = 0u - p[i - q];
    }
}
void orc_rotate_i32(int32_t N, const int32_t *p, int32_t n, int32_t *out) {
    orc_rotate_u32(N, (const uint32_t *)p, n, (uint32_t *)out);
}

void orc_negacyclic_mul_u32(int32_t N, const uint32_t *a, const int32_t *b, uint32_t *res) {
    for (int32_t k = 0; k < N; k++) {
        uint32_t acc = 0;
        for (int32_t j = 0; j <= k; j++) acc += a[k - j] * (uint32_t)b[j];
        for (int32_t j = k + 1; j < N; j++) acc -= a[N + k - j] * (uint32_t)b[j];
        res[k] = acc;
    }
}

/* ================================================================= scheme */

void orc_trgsw_to_fft(orc_plan *pl, const uint32_t *rows, double *rows_f, size_t count) {
    const int32_t N = pl->N;
    for (size_t i = 0; i < count; i++)
        orc_ifft_i32(pl, rows_f + i * N, (const int32_t *)(rows + i * N));   /* ifft_torus: u32 viewed as i32 */
}

void orc_external_product(const orc_params *p, orc_plan *pl, const double *trgsw_f, const uint32_t *trgsw_t,
                          const uint32_t *trlwe, uint32_t *out) {
    const int32_t N = p->N, l = p->l, rows = 2 * l;
    const uint32_t mask = orc_make_decomp_mask((uint32_t)l, (uint32_t)p->bgbit);
    int32_t *dec = (int32_t *)malloc(sizeof(int32_t) * (size_t)rows * N);
    orc_decomp_poly(N, trlwe, (uint32_t)p->bgbit, mask, l, dec);                       /* b digits */
    orc_decomp_poly(N, trlwe + N, (uint32_t)p->bgbit, mask, l, dec + (size_t)l * N);   /* a digits */
    if (pl->backend == ORC_BACKEND_EXACT_INT) {
        uint32_t *tmp = (uint32_t *)malloc(sizeof(uint32_t) * N);
        for (int comp = 0; comp < 2; comp++) {
            uint32_t *o = out + (size_t)comp * N;
            memset(o, 0, sizeof(uint32_t) * N);
            for (int32_t j = 0; j < rows; j++) {
                orc_negacyclic_mul_u32(N, trgsw_t + ((size_t)comp * rows + j) * N, dec + (size_t)j * N, tmp);
                for (int32_t k = 0; k < N; k++) o[k] += tmp[k];
            }
        }
        free(tmp);
    } else {
        double *dec_f = (double *)malloc(sizeof(double) * (size_t)rows * N);
        double *sum = (double *)malloc(sizeof(double) * N);
        double *had = (double *)malloc(sizeof(double) * N);
        for (int32_t j = 0; j < rows; j++) orc_ifft_i32(pl, dec_f + (size_t)j * N, dec + (size_t)j * N);
        for (int comp = 0; comp < 2; comp++) {
            for (int32_t k = 0; k < N; k++) sum[k] = 0.0;                     /* fold from FrrSeries::zero() */
            for (int32_t j = 0; j < rows; j++) {
                orc_hadamard(N, had, trgsw_f + ((size_t)comp * rows + j) * N, dec_f + (size_t)j * N);
                for (int32_t k = 0; k < N; k++) sum[k] = sum[k] + had[k];
            }
            orc_fft_u32(pl, out + (size_t)comp * N, sum);
        }
        free(dec_f); free(sum); free(had);
    }
    free(dec);
}

void orc_cmux(const orc_params *p, orc_plan *pl, const double *trgsw_f, const uint32_t *trgsw_t,
              const uint32_t *rep1, const uint32_t *rep0, uint32_t *out) {
    const int32_t N2 = 2 * p->N;
    uint32_t *d = (uint32_t *)calloc((size_t)N2, sizeof(uint32_t));
    uint32_t *x = (uint32_t *)malloc(sizeof(uint32_t) * N2);
    for (int32_t k = 0; k < N2; k++) d[k] = rep1[k] - rep0[k];
    orc_external_product(p, pl, trgsw_f, trgsw_t, d, x);
    for (int32_t k = 0; k < N2; k++) out[k] = x[k] + rep0[k];
    free(d); free(x);
}

void orc_blind_rotate(const orc_params *p, orc_plan *pl, const double *bk_f, const uint32_t *bk_t,
                      const uint32_t *tlwe, int32_t steps, uint32_t *acc) {
    const int32_t N = p->N, n = p->n, rows = 2 * p->l;
    const uint32_t sh = 32u - (uint32_t)p->nbit - 1u;
    const size_t trgsw_sz = (size_t)2 * rows * N;
    uint32_t *testvec = (uint32_t *)malloc(sizeof(uint32_t) * 2 * N);
    uint32_t *rot = (uint32_t *)malloc(sizeof(uint32_t) * 2 * N);
    for (int32_t k = 0; k < N; k++) { testvec[k] = orc_torus_from_f32(1.0f / 8.0f); testvec[N + k] = 0; }
    const int32_t bbar = (int32_t)(tlwe[n] >> sh);                       /* tfhe.rs:97, floor */
    orc_rotate_u32(N, testvec, -bbar, acc);
    orc_rotate_u32(N, testvec + N, -bbar, acc + N);
    if (steps > n) steps = n;
    for (int32_t i = 0; i < steps; i++) {
        const int32_t abar = (int32_t)((tlwe[i] + (1u << (sh - 1))) >> sh);   /* tfhe.rs:107-108, round */
        orc_rotate_u32(N, acc, abar, rot);
        orc_rotate_u32(N, acc + N, abar, rot + N);
        orc_cmux(p, pl, bk_f ? bk_f + (size_t)i * trgsw_sz : NULL, bk_t ? bk_t + (size_t)i * trgsw_sz : NULL,
                 rot, acc, acc);
    }
    free(testvec); free(rot);
}

void orc_sample_extract(int32_t N, const uint32_t *trlwe, int32_t index, uint32_t *tlwe1) {
    const uint32_t *b = trlwe, *a = trlwe + N;
    for (int32_t i = 0; i < N; i++) tlwe1[i] = (i <= index) ? a[index - i] : 0u - a[N + index - i];
    tlwe1[N] = b[index];
}

void orc_key_switch(const orc_params *p, const uint32_t *ksk, const uint32_t *tlwe1, uint32_t *out) {
    const int32_t N = p->N, n = p->n, t = p->ks_t, bb = p->ks_basebit;
    const int32_t base1 = (1 << bb) - 1;
    const uint32_t total = 32;
    const uint32_t round = ((total - (uint32_t)(t * bb)) != 0) ? (1u << (total - (uint32_t)(t * bb) - 1)) : 0u;
    const uint32_t mask = (1u << bb) - 1;
    for (int32_t k = 0; k < n; k++) out[k] = 0;
    out[n] = tlwe1[N];
    for (int32_t i = 0; i < N; i++) {
        const uint32_t u = tlwe1[i] + round;
        for (int32_t l = 0; l < t; l++) {
            const uint32_t d = (u >> (total - (uint32_t)bb * (uint32_t)(l + 1))) & mask;
            if (d != 0) {
                const uint32_t *row = ksk + (((size_t)i * t + l) * base1 + (d - 1)) * (size_t)(n + 1);
                for (int32_t k = 0; k <= n; k++) out[k] -= row[k];
            }
        }
    }
}

/* The same key switch reading the key in the reference's OWN container shape, KeySwitchingKey(Vec<[[TLWERep<M>; IKS_T]; IKS_L]>)
 * with IKS_T = 2^IKS_BASEBIT entries per level (hom_nand/src/tlwe.rs:178-180, 243-245): get(i, l, t) = [i][l][t - 1]
 * (tlwe.rs:281-283), called with t = digit in 1 .. base-1 (tlwe.rs:43-73) -- entry [base - 1] of a level is never read. */
void orc_key_switch_ref(const orc_params *p, const uint32_t *ksk_ref, const uint32_t *tlwe1, uint32_t *out) {
    const int32_t N = p->N, n = p->n, t = p->ks_t, bb = p->ks_basebit;
    const int32_t iks_t = 1 << bb;
    const uint32_t total = 32;
    const uint32_t round = ((total - (uint32_t)(t * bb)) != 0) ? (1u << (total - (uint32_t)(t * bb) - 1)) : 0u;
    const uint32_t mask = (1u << bb) - 1;
    for (int32_t k = 0; k < n; k++) out[k] = 0;
    out[n] = tlwe1[N];
    for (int32_t i = 0; i < N; i++) {
        const uint32_t u = tlwe1[i] + round;
        for (int32_t l = 0; l < t; l++) {
            const uint32_t d = (u >> (total - (uint32_t)bb * (uint32_t)(l + 1))) & mask;
            if (d != 0) {
                const uint32_t *row = ksk_ref + (((size_t)i * t + l) * iks_t + (d - 1)) * (size_t)(n + 1);
                for (int32_t k = 0; k <= n; k++) out[k] -= row[k];
            }
        }
    }
}

void orc_gate_linear(const orc_params *p, int op, const uint32_t *in0, const uint32_t *in1, uint32_t *t) {
    const int32_t n = p->n;
    const uint32_t c8 = orc_torus_from_f32(1.0f / 8.0f), c4 = orc_torus_from_f32(2.0f * (1.0f / 8.0f));
    switch (op) {
    case ORC_NAND: for (int32_t k = 0; k <= n; k++) t[k] = 0u - (in0[k] + in1[k]); t[n] += c8; break;
    case ORC_AND:  for (int32_t k = 0; k <= n; k++) t[k] = in0[k] + in1[k]; t[n] -= c8; break;
    case ORC_OR:   for (int32_t k = 0; k <= n; k++) t[k] = in0[k] + in1[k]; t[n] += c8; break;
    case ORC_XOR:  for (int32_t k = 0; k <= n; k++) t[k] = (in0[k] + in1[k]) * 2u; t[n] += c4; break;
    case ORC_NOT:  for (int32_t k = 0; k <= n; k++) t[k] = 0u - in0[k]; break;
    case ORC_ANDNY: for (int32_t k = 0; k <= n; k++) t[k] = in1[k] - in0[k]; t[n] -= c8; break;   /* hom_and(-c, in0), tfhe.rs:34 */
    default:       for (int32_t k = 0; k <= n; k++) t[k] = in0[k]; break;
    }
}

void orc_bootstrap(const orc_params *p, orc_plan *pl, const double *bk_f, const uint32_t *bk_t,
                   const uint32_t *ksk, const uint32_t *t, uint32_t *out) {
    const int32_t N = p->N;
    uint32_t *acc = (uint32_t *)malloc(sizeof(uint32_t) * 2 * N);
    uint32_t *t1 = (uint32_t *)malloc(sizeof(uint32_t) * (N + 1));
    orc_blind_rotate(p, pl, bk_f, bk_t, t, p->n, acc);
    orc_sample_extract(N, acc, 0, t1);
    orc_key_switch(p, ksk, t1, out);
    free(acc); free(t1);
}

void orc_gate(const orc_params *p, orc_plan *pl, int op, const double *bk_f, const uint32_t *bk_t,
              const uint32_t *ksk, const uint32_t *in0, const uint32_t *in1, uint32_t *out) {
    uint32_t *t = (uint32_t *)malloc(sizeof(uint32_t) * (p->n + 1));
    orc_gate_linear(p, op, in0, in1, t);
    orc_bootstrap(p, pl, bk_f, bk_t, ksk, t, out);
    free(t);
}

void orc_mux(const orc_params *p, orc_plan *pl, const double *bk_f, const uint32_t *bk_t,
             const uint32_t *ksk, const uint32_t *c, const uint32_t *in0, const uint32_t *in1, uint32_t *out) {
    const int32_t n = p->n;
    uint32_t *i1 = (uint32_t *)malloc(sizeof(uint32_t) * (n + 1));
    uint32_t *i0 = (uint32_t *)malloc(sizeof(uint32_t) * (n + 1));
    uint32_t *nc = (uint32_t *)malloc(sizeof(uint32_t) * (n + 1));
    orc_gate(p, pl, ORC_AND, bk_f, bk_t, ksk, c, in1, i1);
    for (int32_t k = 0; k <= n; k++) nc[k] = 0u - c[k];
    orc_gate(p, pl, ORC_AND, bk_f, bk_t, ksk, nc, in0, i0);
    for (int32_t k = 0; k <= n; k++) nc[k] = i1[k] + i0[k];
    nc[n] += orc_torus_from_f32(1.0f / 8.0f);
    orc_bootstrap(p, pl, bk_f, bk_t, ksk, nc, out);
    free(i1); free(i0); free(nc);
}

typedef struct {
    const orc_params *p; int backend; int op; const double *bk_f; const uint32_t *bk_t; const uint32_t *ksk;
    const uint32_t *in0, *in1; uint32_t *out; size_t begin, end;
    pthread_barrier_t *ready;   /* every worker + the timing thread: plans exist, nothing has been computed yet */
} mt_job;

/* optional: every worker thread gets its own handle of an external transform (the reference's Spqlios_new /
 * Spqlios_ifft_i32 / Spqlios_fft_u32 from oracle/_ref), as the reference's thread_local FFT_MAP would (math.rs:349-360) */
static void *(*g_mt_new)(int32_t) = NULL;
static orc_fwd_hook g_mt_fwd = NULL;
static orc_inv_hook g_mt_inv = NULL;
void orc_set_mt_hooks(void *(*new_fn)(int32_t), orc_fwd_hook fwd, orc_inv_hook inv) { g_mt_new = new_fn; g_mt_fwd = fwd; g_mt_inv = inv; }

static void *mt_worker(void *arg) {
    mt_job *j = (mt_job *)arg;
    orc_plan *pl = orc_plan_new(j->p->N);
    orc_plan_set_backend(pl, j->backend);
    if (j->backend == ORC_BACKEND_HOOK && g_mt_new) orc_plan_set_hooks(pl, g_mt_new(j->p->N), g_mt_fwd, g_mt_inv);
    const size_t w = (size_t)j->p->n + 1;
    pthread_barrier_wait(j->ready);
    for (size_t g = j->begin; g < j->end; g++)
        orc_gate(j->p, pl, j->op, j->bk_f, j->bk_t, j->ksk, j->in0 + g * w, j->in1 ? j->in1 + g * w : NULL, j->out + g * w);
    orc_plan_free(pl);
    return NULL;
}

/* count gates on nthreads threads, one contiguous range and one plan (the reference: one Spqlios handle) per thread.
 * Returns the seconds from the moment every thread holds its plan to the last thread's last gate: thread and plan
 * creation (table building, libm) stay outside the timed region, as they are outside the reference's own timeit!. */
double orc_gate_batch_mt(const orc_params *p, int backend, int op, const double *bk_f, const uint32_t *bk_t,
                         const uint32_t *ksk, const uint32_t *in0, const uint32_t *in1, uint32_t *out,
                         size_t count, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    mt_job *jobs = (mt_job *)malloc(sizeof(mt_job) * (size_t)nthreads);
    pthread_barrier_t ready;
    pthread_barrier_init(&ready, NULL, (unsigned)nthreads + 1u);
    struct timespec t0, t1;
    for (int t = 0; t < nthreads; t++) {
        jobs[t] = (mt_job){p, backend, op, bk_f, bk_t, ksk, in0, in1, out,
                           count * (size_t)t / (size_t)nthreads, count * (size_t)(t + 1) / (size_t)nthreads, &ready};
        pthread_create(&th[t], NULL, mt_worker, &jobs[t]);
    }
    pthread_barrier_wait(&ready);
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    pthread_barrier_destroy(&ready);
    free(th); free(jobs);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* The all-core CPU baseline of bench.py (round 4).  What differs from orc_gate_batch_mt:
 *  - `count` gates are run over `in_count` distinct inputs (gate g takes input g % in_count and writes out[g]): a thread gets tens of gates,
 *    not a handful, so ramp-up and stragglers stop dominating the figure;
 *  - every thread may be pinned to one CPU (cpus[t] >= 0), and the bootstrapping key spectra and the key-switching key (62 MB each, streamed
 *    once per gate by every thread) are replicated per memory node: node_of_thread[t] names the replica thread t reads, and the FIRST thread
 *    of a node allocates and fills that replica itself, i.e. first-touches it on its own node (Linux places a page on the node of the thread
 *    that first writes it).  One copy first-touched by the main thread -- round 3 -- made every core of the other sockets stream it across
 *    the inter-socket links.
 * The reference's own threading model would be exactly this: `&self` keys shared read-only, one FFT plan per thread (thread_local! FFT_MAP,
 * utils/src/math.rs:349-351).  Returns the seconds from the moment every thread holds its plan and every replica is filled to the last
 * thread's last gate; -1.0 on allocation failure. */
typedef struct {
    const orc_params *p; int backend; int op; const double *bk_f; size_t bk_doubles; const uint32_t *ksk; size_t ksk_words;
    const uint32_t *in0, *in1; uint32_t *out; size_t in_count, begin, end;
    int cpu, node, leader;
    double **rep_bk; uint32_t **rep_ksk;        /* per node, filled by the node's leader */
    pthread_barrier_t *filled, *ready;
    int *fail;
    struct start_gate *gate;
} numa_job;

/* Every worker waits here until ALL threads exist: the barriers below are sized for nthreads, so a worker that ran ahead into one while a
 * later pthread_create failed (thread / process limits of the host) would wait for ever.  state: 0 = wait, 1 = go, 2 = give up. */
typedef struct start_gate { pthread_mutex_t m; pthread_cond_t c; int state; } start_gate;
static int gate_wait(start_gate *g) {
    pthread_mutex_lock(&g->m);
    while (g->state == 0) pthread_cond_wait(&g->c, &g->m);
    const int st = g->state;
    pthread_mutex_unlock(&g->m);
    return st;
}
static void gate_open(start_gate *g, int state) {
    pthread_mutex_lock(&g->m);
    g->state = state;
    pthread_cond_broadcast(&g->c);
    pthread_mutex_unlock(&g->m);
}

static void *numa_worker(void *arg) {
    numa_job *j = (numa_job *)arg;
    if (gate_wait(j->gate) != 1) return NULL;
    if (j->cpu >= 0) {
        cpu_set_t set;
        CPU_ZERO(&set);
        CPU_SET(j->cpu, &set);
        pthread_setaffinity_np(pthread_self(), sizeof(set), &set);      /* best effort: a refused mask leaves the thread where it is */
    }
    if (j->leader) {
        double *b = (double *)malloc(sizeof(double) * j->bk_doubles);
        uint32_t *k = (uint32_t *)malloc(sizeof(uint32_t) * j->ksk_words);
        if (b && k) {
            memcpy(b, j->bk_f, sizeof(double) * j->bk_doubles);          /* first touch: on this thread's node */
            memcpy(k, j->ksk, sizeof(uint32_t) * j->ksk_words);
        } else {
            *j->fail = 1;
        }
        j->rep_bk[j->node] = b;
        j->rep_ksk[j->node] = k;
    }
    orc_plan *pl = orc_plan_new(j->p->N);
    orc_plan_set_backend(pl, j->backend);
    if (j->backend == ORC_BACKEND_HOOK && g_mt_new) orc_plan_set_hooks(pl, g_mt_new(j->p->N), g_mt_fwd, g_mt_inv);
    pthread_barrier_wait(j->filled);
    const double *bk = j->rep_bk[j->node];
    const uint32_t *ks = j->rep_ksk[j->node];
    const size_t w = (size_t)j->p->n + 1;
    pthread_barrier_wait(j->ready);
    if (!*j->fail)
        for (size_t g = j->begin; g < j->end; g++) {
            const size_t s = g % j->in_count;
            orc_gate(j->p, pl, j->op, bk, NULL, ks, j->in0 + s * w, j->in1 ? j->in1 + s * w : NULL, j->out + g * w);
        }
    orc_plan_free(pl);
    return NULL;
}

double orc_gate_batch_mt_numa(const orc_params *p, int backend, int op, const double *bk_f, const uint32_t *ksk,
                              const uint32_t *in0, const uint32_t *in1, size_t in_count, uint32_t *out, size_t count,
                              int nthreads, const int *cpus, const int *node_of_thread, int nnodes) {
    if (nthreads < 1 || nnodes < 1 || in_count < 1) return -1.0;
    const size_t bk_doubles = (size_t)p->n * 2 * 2 * (size_t)p->l * (size_t)p->N;
    const size_t ksk_words = (size_t)p->N * (size_t)p->ks_t * (((size_t)1 << p->ks_basebit) - 1) * ((size_t)p->n + 1);
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    numa_job *jobs = (numa_job *)malloc(sizeof(numa_job) * (size_t)nthreads);
    double **rep_bk = (double **)calloc((size_t)nnodes, sizeof(double *));
    uint32_t **rep_ksk = (uint32_t **)calloc((size_t)nnodes, sizeof(uint32_t *));
    int *seen = (int *)calloc((size_t)nnodes, sizeof(int));
    int fail = 0;
    pthread_barrier_t filled, ready;
    start_gate gate = {PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, 0};
    if (!th || !jobs || !rep_bk || !rep_ksk || !seen) { free(th); free(jobs); free(rep_bk); free(rep_ksk); free(seen); return -1.0; }
    if (pthread_barrier_init(&filled, NULL, (unsigned)nthreads) != 0) { free(th); free(jobs); free(rep_bk); free(rep_ksk); free(seen); return -1.0; }
    if (pthread_barrier_init(&ready, NULL, (unsigned)nthreads + 1u) != 0) {
        pthread_barrier_destroy(&filled);
        free(th); free(jobs); free(rep_bk); free(rep_ksk); free(seen);
        return -1.0;
    }
    struct timespec t0, t1;
    int created = 0;
    for (int t = 0; t < nthreads; t++) {
        int node = node_of_thread ? node_of_thread[t] : 0;
        if (node < 0 || node >= nnodes) node = 0;
        const int leader = !seen[node];
        seen[node] = 1;
        jobs[t] = (numa_job){p, backend, op, bk_f, bk_doubles, ksk, ksk_words, in0, in1, out, in_count,
                             count * (size_t)t / (size_t)nthreads, count * (size_t)(t + 1) / (size_t)nthreads,
                             cpus ? cpus[t] : -1, node, leader, rep_bk, rep_ksk, &filled, &ready, &fail, &gate};
        if (pthread_create(&th[t], NULL, numa_worker, &jobs[t]) != 0) break;
        created++;
    }
    if (created < nthreads) {              /* the threads that exist never reach a barrier: release them, join them, report failure */
        gate_open(&gate, 2);
        for (int t = 0; t < created; t++) pthread_join(th[t], NULL);
        pthread_barrier_destroy(&filled);
        pthread_barrier_destroy(&ready);
        free(rep_bk); free(rep_ksk); free(seen); free(th); free(jobs);
        return -1.0;
    }
    gate_open(&gate, 1);
    pthread_barrier_wait(&ready);
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    pthread_barrier_destroy(&filled);
    pthread_barrier_destroy(&ready);
    for (int k = 0; k < nnodes; k++) { free(rep_bk[k]); free(rep_ksk[k]); }
    free(rep_bk); free(rep_ksk); free(seen); free(th); free(jobs);
    if (fail) return -1.0;
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* ================================================================= keys / encryption */

static uint64_t splitmix64(uint64_t *x) {
    uint64_t z = (*x += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
void orc_rng_seed(orc_rng *r, uint64_t seed) { for (int i = 0; i < 4; i++) r->s[i] = splitmix64(&seed); }
static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
uint64_t orc_rng_next(orc_rng *r) {                       /* xoshiro256** */
    uint64_t *s = r->s;
    const uint64_t result = rotl64(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl64(s[3], 45);
    return result;
}
static inline float rng_unit_f32(orc_rng *r) { return (float)(orc_rng_next(r) >> 40) * (1.0f / 16777216.0f); }

uint32_t orc_rng_uniform_torus(orc_rng *r) { return orc_torus_from_f32(rng_unit_f32(r)); }

/* Normal(-0.0, alpha) in f32 then torus!().  The reference draws from rand_distr (unseedable,
 * math.rs:417-424); the oracle uses a libm-free Irwin-Hall(12) variate so that keys regenerate
 * bit-identically from a seed on any host (distribution parity, not bit parity -- SURVEY 8c). */
uint32_t orc_rng_gaussian_torus(orc_rng *r, float alpha) {
    double z = -6.0;
    for (int i = 0; i < 12; i++) z += (double)(orc_rng_next(r) >> 11) * (1.0 / 9007199254740992.0);
    return orc_torus_from_f32((float)z * alpha);
}

void orc_gen_binary_key(orc_rng *r, int32_t len, int32_t *key) {
    for (int32_t i = 0; i < len; i++) key[i] = (int32_t)(orc_rng_next(r) >> 63);
}

uint32_t orc_binary2torus(int bit) { return orc_torus_from_f32(bit ? 1.0f / 8.0f : -1.0f / 8.0f); }
int orc_torus2binary(uint32_t t) {
    const float X = 1.0f / 4294967296.0f;                 /* 1.0 / (u32::MAX as f32) */
    float f = (float)t * X;
    return (f < 0.5f) ? 1 : 0;
}

void orc_tlwe_encrypt(orc_rng *r, int32_t n, const int32_t *key, uint32_t msg, float alpha, uint32_t *ct) {
    uint32_t b = 0;
    for (int32_t i = 0; i < n; i++) ct[i] = orc_rng_uniform_torus(r);
    const uint32_t e = orc_rng_gaussian_torus(r, alpha);
    for (int32_t i = 0; i < n; i++) if (key[i]) b += ct[i];
    ct[n] = b + e + msg;
}
uint32_t orc_tlwe_phase(int32_t n, const int32_t *key, const uint32_t *ct) {
    uint32_t s = 0;
    for (int32_t i = 0; i < n; i++) if (key[i]) s += ct[i];
    return ct[n] - s;
}

/* a.fft_cross(key), math.rs:337-347: ifft_torus(a), ifft_int(key), hadamard, fft_torus */
static void fft_cross_key(orc_plan *pl, int32_t N, const uint32_t *a, const int32_t *key, uint32_t *res) {
    if (pl->backend == ORC_BACKEND_EXACT_INT) { orc_negacyclic_mul_u32(N, a, key, res); return; }
    double *fa = pl->scratch + 2 * (size_t)N, *fk = pl->scratch + 3 * (size_t)N, *h = pl->scratch;
    orc_ifft_i32(pl, fa, (const int32_t *)a);
    orc_ifft_i32(pl, fk, key);
    orc_hadamard(N, h, fa, fk);
    orc_fft_u32(pl, res, h);
}

void orc_trlwe_encrypt(orc_rng *r, orc_plan *pl, int32_t N, const int32_t *key, const uint32_t *msg,
                       float alpha, uint32_t *ct) {
    uint32_t *b = ct, *a = ct + N;
    uint32_t *e = (uint32_t *)malloc(sizeof(uint32_t) * N);
    for (int32_t k = 0; k < N; k++) a[k] = orc_rng_uniform_torus(r);
    for (int32_t k = 0; k < N; k++) e[k] = orc_rng_gaussian_torus(r, alpha);
    fft_cross_key(pl, N, a, key, b);
    for (int32_t k = 0; k < N; k++) b[k] = b[k] + (msg ? msg[k] : 0u) + e[k];
    free(e);
}
void orc_trlwe_phase(orc_plan *pl, int32_t N, const int32_t *key, const uint32_t *ct, uint32_t *phase) {
    fft_cross_key(pl, N, ct + N, key, phase);
    for (int32_t k = 0; k < N; k++) phase[k] = ct[k] - phase[k];
}

void orc_trgsw_encrypt(orc_rng *r, orc_plan *pl, const orc_params *p, const int32_t *key, int32_t mu,
                       float alpha, uint32_t *ct) {
    const int32_t N = p->N, l = p->l, rows = 2 * l;
    uint32_t *pair = (uint32_t *)malloc(sizeof(uint32_t) * 2 * N);
    for (int32_t j = 0; j < rows; j++) {                  /* create_zero_encrypted_pols, trgsw.rs:118-138 */
        orc_trlwe_encrypt(r, pl, N, key, NULL, alpha, pair);
        memcpy(ct + (size_t)j * N, pair, sizeof(uint32_t) * N);                     /* cipher[j] */
        memcpy(ct + ((size_t)rows + j) * N, pair + N, sizeof(uint32_t) * N);        /* p_key[j]  */
    }
    const float bg_inv = 1.0f / (float)(1 << p->bgbit);
    for (int32_t i = 0; i < l; i++) {                     /* trgsw.rs:221-227 */
        float pw = 1.0f;
        for (int32_t e = 0; e < 1 + i; e++) pw *= bg_inv;
        const uint32_t t = orc_torus_from_f32((float)mu * pw);
        ct[(size_t)i * N] += t;                            /* cipher[i].add_constant */
        ct[((size_t)rows + i + l) * N] += t;               /* p_key[i+L].add_constant */
    }
    free(pair);
}

void orc_bk_gen(orc_rng *r, orc_plan *pl, const orc_params *p, const int32_t *key0, const int32_t *key1,
                float alpha, uint32_t *bk_t) {
    const size_t sz = (size_t)2 * 2 * p->l * p->N;
    for (int32_t i = 0; i < p->n; i++) orc_trgsw_encrypt(r, pl, p, key1, key0[i], alpha, bk_t + (size_t)i * sz);
}

void orc_ksk_gen(orc_rng *r, const orc_params *p, const int32_t *key1, const int32_t *key0, float alpha,
                 uint32_t *ksk) {
    const int32_t N = p->N, n = p->n, t = p->ks_t, bb = p->ks_basebit, base1 = (1 << bb) - 1;
    for (int32_t i = 0; i < N; i++)
        for (int32_t l = 0; l < t; l++)
            for (int32_t d = 0; d < base1; d++) {
                /* torus!(s_i * 0.5^(basebit*(l+1)) * (d+1)), tlwe.rs:252-256 */
                float pw = 1.0f;
                for (int32_t e = 0; e < bb * (l + 1); e++) pw *= 0.5f;
                const uint32_t item = orc_torus_from_f32((float)key1[i] * pw * (float)(d + 1));
                orc_tlwe_encrypt(r, n, key0, item, alpha, ksk + (((size_t)i * t + l) * base1 + d) * (size_t)(n + 1));
            }
}

/* KeySwitchingKey::new in the reference's shape (tlwe.rs:247-277): IKS_T = base entries per level, entry t - 1 =
 * TLWE(t * s_i / 2^(basebit (l+1))), t = 1 .. base.  Entries 1 .. base-1 are taken from `ksk` (orc_ksk_gen, so that fixtures made
 * with the compact key stay valid); entry `base` is encrypted here. */
void orc_ksk_expand_ref(orc_rng *r, const orc_params *p, const int32_t *key1, const int32_t *key0, float alpha,
                        const uint32_t *ksk, uint32_t *ksk_ref) {
    const int32_t N = p->N, n = p->n, t = p->ks_t, bb = p->ks_basebit, base = 1 << bb;
    const size_t w = (size_t)n + 1;
    for (int32_t i = 0; i < N; i++)
        for (int32_t l = 0; l < t; l++) {
            const size_t il = (size_t)i * t + l;
            memcpy(ksk_ref + il * base * w, ksk + il * (base - 1) * w, (size_t)(base - 1) * w * sizeof(uint32_t));
            float pw = 1.0f;
            for (int32_t e = 0; e < bb * (l + 1); e++) pw *= 0.5f;
            const uint32_t item = orc_torus_from_f32((float)key1[i] * pw * (float)base);
            orc_tlwe_encrypt(r, n, key0, item, alpha, ksk_ref + (il * base + (base - 1)) * w);
        }
}

uint64_t orc_fnv64(const void *data, size_t bytes) {
    const unsigned char *p = (const unsigned char *)data;
    uint64_t h = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < bytes; i++) { h ^= p[i]; h *= 0x100000001b3ull; }
    return h;
}
