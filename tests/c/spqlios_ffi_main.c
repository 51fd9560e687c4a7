/* The reference's FFT FFI (utils/src/spqlios.rs:18-32, spqlios-wrapper.cpp:9-53) called BY NAME through two libraries in one
 * process -- the reference's own build (oracle/_ref/libspqlios_ref.so, the checker) and the engine (librtfhe_hip.so, the thing
 * tested) -- with the same inputs; every output must be the same bytes (Spqlios_poly_mul: +-1 LSB, the reference's -Ofast build
 * contracts that loop to FMA).  One N per process (the reference caches 2/N in a function-local static).
 *   usage: spqlios_ffi <librtfhe_hip.so> <libspqlios_ref.so> <N> */
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef void *(*new_fn)(int32_t);
typedef void (*del_fn)(void *);
typedef void (*f_dd)(void *, double *, const double *);
typedef void (*f_du)(void *, double *, const uint32_t *);
typedef void (*f_di)(void *, double *, const int32_t *);
typedef void (*f_ud)(void *, uint32_t *, const double *);
typedef void (*f_uuu)(void *, uint32_t *, const uint32_t *, const uint32_t *);

typedef struct { void *so; new_fn nw; del_fn dl; f_dd ifft, fft; f_du ifft_u32; f_di ifft_i32; f_ud fft_u32; f_uuu poly_mul; } api;

static int load(api *a, const char *path) {
    a->so = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!a->so) { fprintf(stderr, "dlopen %s: %s\n", path, dlerror()); return 1; }
#define SYM(field, type, name) do { a->field = (type)dlsym(a->so, name); if (!a->field) { fprintf(stderr, "%s lacks %s\n", path, name); return 1; } } while (0)
    SYM(nw, new_fn, "Spqlios_new"); SYM(dl, del_fn, "Spqlios_destructor"); SYM(ifft, f_dd, "Spqlios_ifft"); SYM(fft, f_dd, "Spqlios_fft");
    SYM(ifft_u32, f_du, "Spqlios_ifft_u32"); SYM(ifft_i32, f_di, "Spqlios_ifft_i32"); SYM(fft_u32, f_ud, "Spqlios_fft_u32");
    SYM(poly_mul, f_uuu, "Spqlios_poly_mul");
    return 0;
}

static uint64_t rng_state = 0x243f6a8885a308d3ull;
static uint64_t rnd(void) { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

int main(int argc, char **argv) {
    if (argc != 4) { fprintf(stderr, "usage: %s librtfhe_hip.so libspqlios_ref.so N\n", argv[0]); return 2; }
    const int N = atoi(argv[3]);
    api mine, ref;
    if (load(&mine, argv[1]) || load(&ref, argv[2])) return 2;
    if (mine.nw(48) != NULL) { fprintf(stderr, "Spqlios_new(48) must return NULL\n"); return 1; }
    void *hm = mine.nw(N), *hr = ref.nw(N);
    if (!hm || !hr) { fprintf(stderr, "Spqlios_new(%d): engine %p reference %p\n", N, hm, hr); return 1; }
    int32_t *si = malloc(sizeof(int32_t) * N);
    uint32_t *su = malloc(4 * N), *sb = malloc(4 * N), *um = malloc(4 * N), *ur = malloc(4 * N);
    double *sd = malloc(8 * N), *dm = malloc(8 * N), *dr = malloc(8 * N);
    int bad = 0;
    for (int t = 0; t < 24 && !bad; t++) {
        for (int k = 0; k < N; k++) {
            const uint64_t r = rnd();
            si[k] = t % 3 == 0 ? (int32_t)(r % 64) - 32 : t % 3 == 1 ? (int32_t)(uint32_t)r : (int32_t)(r & 1);
            su[k] = (uint32_t)(r >> 32);
            sb[k] = (uint32_t)((r >> 8) % 64);
        }
        mine.ifft_i32(hm, dm, si); ref.ifft_i32(hr, dr, si);
        if (memcmp(dm, dr, 8 * N)) { fprintf(stderr, "Spqlios_ifft_i32 differs (trial %d)\n", t); bad = 1; break; }
        mine.ifft_u32(hm, dm, su); ref.ifft_u32(hr, dr, su);
        if (memcmp(dm, dr, 8 * N)) { fprintf(stderr, "Spqlios_ifft_u32 differs (trial %d)\n", t); bad = 1; break; }
        for (int k = 0; k < N; k++) sd[k] = dr[k] * (double)(1 + rnd() % 1000);     /* a spectrum with large entries */
        mine.fft_u32(hm, um, sd); ref.fft_u32(hr, ur, sd);
        if (memcmp(um, ur, 4 * N)) { fprintf(stderr, "Spqlios_fft_u32 differs (trial %d)\n", t); bad = 1; break; }
        mine.fft(hm, dm, sd); ref.fft(hr, dr, sd);
        if (memcmp(dm, dr, 8 * N)) { fprintf(stderr, "Spqlios_fft differs (trial %d)\n", t); bad = 1; break; }
        for (int k = 0; k < N; k++) sd[k] = (double)si[k] * 0.5;
        mine.ifft(hm, dm, sd); ref.ifft(hr, dr, sd);
        if (memcmp(dm, dr, 8 * N)) { fprintf(stderr, "Spqlios_ifft differs (trial %d)\n", t); bad = 1; break; }
        mine.poly_mul(hm, um, su, sb); ref.poly_mul(hr, ur, su, sb);
        for (int k = 0; k < N; k++) {
            const int32_t d = (int32_t)(um[k] - ur[k]);
            if (d > 1 || d < -1) { fprintf(stderr, "Spqlios_poly_mul differs by %d at %d (trial %d)\n", d, k, t); bad = 1; break; }
        }
    }
    mine.dl(hm); ref.dl(hr);
    printf(bad ? "FAILED\n" : "spqlios ffi ok N=%d\n", N);
    return bad;
}
