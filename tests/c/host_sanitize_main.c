/* Walks the product's host-only sources (key generation / encryption, wire format: rustfhe_amd/csrc/rtfhe_keygen.cpp, rtfhe_wire.cpp)
 * under AddressSanitizer + UndefinedBehaviorSanitizer on a small parameter set: secure and deterministic key generation, encrypt ->
 * decrypt, key file and ciphertext file round trips, a truncated and a corrupted file. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "rtfhe.h"

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "CHECK failed: %s (line %d)\n", #c, __LINE__); exit(1); } } while (0)

int main(int argc, char **argv) {
    const char *dir = argc > 1 ? argv[1] : "/tmp";
    rtfhe_params p = {0};
    p.n = 20; p.N = 64; p.nbit = 6; p.l = 3; p.bgbit = 6; p.ks_t = 8; p.ks_basebit = 2;
    const size_t bkw = (size_t)p.n * 2 * 2 * p.l * p.N, ksw = (size_t)p.N * p.ks_t * 3 * (p.n + 1);
    int32_t *key0 = malloc(sizeof(int32_t) * p.n), *key1 = malloc(sizeof(int32_t) * p.N);
    int32_t *k0b = malloc(sizeof(int32_t) * p.n), *k1b = malloc(sizeof(int32_t) * p.N);
    uint32_t *bk = malloc(4 * bkw), *ksk = malloc(4 * ksw), *bk2 = malloc(4 * bkw), *ksk2 = malloc(4 * ksw);
    CHECK(key0 && key1 && k0b && k1b && bk && ksk && bk2 && ksk2);

    CHECK(rtfhe_keygen_deterministic(&p, 7, key0, key1, bk, ksk) == 0);
    CHECK(rtfhe_keygen_deterministic(&p, 7, k0b, k1b, bk2, ksk2) == 0);
    CHECK(!memcmp(key0, k0b, sizeof(int32_t) * p.n) && !memcmp(bk, bk2, 4 * bkw) && !memcmp(ksk, ksk2, 4 * ksw));
    CHECK(rtfhe_keygen_with_keys_deterministic(&p, 9, key0, key1, bk2, ksk2) == 0);
    CHECK(rtfhe_keygen(&p, k0b, k1b, bk2, ksk2) == 0);                      /* OS CSPRNG */
    CHECK(rtfhe_keygen_with_keys(&p, key0, key1, bk2, ksk2) == 0);
    for (int i = 0; i < p.n; i++) CHECK(key0[i] == 0 || key0[i] == 1);

    enum { CNT = 37 };
    uint8_t bits[CNT], back[CNT];
    uint32_t *ct = malloc(4 * (size_t)CNT * (p.n + 1)), *ct2 = malloc(4 * (size_t)CNT * (p.n + 1)), phase[CNT];
    CHECK(ct && ct2);
    for (int i = 0; i < CNT; i++) bits[i] = (uint8_t)((i * 7 + 3) & 1);
    CHECK(rtfhe_tlwe_encrypt_bits_deterministic(&p, key0, 11, bits, ct, CNT) == 0);
    CHECK(rtfhe_tlwe_decrypt_bits(&p, key0, ct, back, CNT) == 0 && !memcmp(bits, back, CNT));
    CHECK(rtfhe_tlwe_encrypt_bits(&p, key0, bits, ct2, CNT) == 0);
    CHECK(rtfhe_tlwe_decrypt_bits(&p, key0, ct2, back, CNT) == 0 && !memcmp(bits, back, CNT));
    CHECK(rtfhe_tlwe_phase(&p, key0, ct, phase, CNT) == 0);
    CHECK(rtfhe_tlwe_encrypt_bits(&p, key0, bits, ct2, 0) == 0);             /* empty batch */

    char path[512], path2[512];
    snprintf(path, sizeof path, "%s/keys.bin", dir);
    snprintf(path2, sizeof path2, "%s/cts.bin", dir);
    CHECK(rtfhe_keys_write(path, &p, key0, key1, bk, ksk) == 0);
    rtfhe_params q; uint32_t flags = 0;
    CHECK(rtfhe_keys_read_header(path, &q, &flags) == 0 && q.n == p.n && q.N == p.N && flags == 7);
    memset(bk2, 0, 4 * bkw); memset(ksk2, 0, 4 * ksw);
    CHECK(rtfhe_keys_read(path, k0b, k1b, bk2, ksk2) == 0);
    CHECK(!memcmp(key0, k0b, sizeof(int32_t) * p.n) && !memcmp(key1, k1b, sizeof(int32_t) * p.N) && !memcmp(bk, bk2, 4 * bkw) && !memcmp(ksk, ksk2, 4 * ksw));
    CHECK(rtfhe_keys_read(path, NULL, NULL, NULL, ksk2) == 0);              /* skip sections */
    CHECK(rtfhe_keys_write(path, &p, NULL, NULL, bk, NULL) == 0);            /* evaluation-key-only file */
    CHECK(rtfhe_keys_read_header(path, &q, &flags) == 0 && flags == 1);
    CHECK(rtfhe_keys_read(path, k0b, NULL, NULL, NULL) != 0);                /* a section the file does not hold */

    CHECK(rtfhe_tlwe_write(path2, p.n, ct, CNT) == 0);
    int32_t n = 0; uint64_t count = 0;
    CHECK(rtfhe_tlwe_read(path2, &n, &count, NULL, 0) == 0 && n == p.n && count == CNT);
    CHECK(rtfhe_tlwe_read(path2, &n, &count, ct2, CNT - 1) != 0);            /* capacity too small */
    CHECK(rtfhe_tlwe_read(path2, &n, &count, ct2, CNT) == 0 && !memcmp(ct, ct2, 4 * (size_t)CNT * (p.n + 1)));
    {   /* corruption and truncation are detected */
        FILE *f = fopen(path2, "r+b"); CHECK(f);
        fseek(f, 40, SEEK_SET); int c = fgetc(f); fseek(f, 40, SEEK_SET); fputc(c ^ 1, f); fclose(f);
        CHECK(rtfhe_tlwe_read(path2, &n, &count, ct2, CNT) != 0);
        f = fopen(path2, "wb"); CHECK(f); fwrite("RTFHECT1", 1, 8, f); fclose(f);
        CHECK(rtfhe_tlwe_read(path2, &n, &count, ct2, CNT) != 0);
        CHECK(rtfhe_tlwe_read("/nonexistent/dir/x.bin", &n, &count, NULL, 0) != 0);
    }
    {   /* twiddle table files (the table-file format of rtfhe_wire.cpp): round trip, wrong degree, corruption, truncation */
        char path3[512];
        snprintf(path3, sizeof path3, "%s/tw.bin", dir);
        const int32_t TN = 64;
        double *ta = malloc(sizeof(double) * 2 * TN), *tb = malloc(sizeof(double) * 2 * TN), *ra = malloc(sizeof(double) * 2 * TN), *rb = malloc(sizeof(double) * 2 * TN);
        for (int k = 0; k < 2 * TN; k++) { ta[k] = 0.25 * k - 3.0; tb[k] = -0.0 * k + 1.0 / (k + 1); }
        CHECK(rtfhe_twiddles_file_write(path3, TN, ta, tb) == 0);
        CHECK(rtfhe_twiddles_file_read(path3, TN, ra, rb) == 0 && !memcmp(ta, ra, sizeof(double) * 2 * TN) && !memcmp(tb, rb, sizeof(double) * 2 * TN));
        CHECK(rtfhe_twiddles_file_read(path3, 2 * TN, ra, rb) != 0);
        CHECK(rtfhe_twiddles_file_write(path3, 8, ta, tb) != 0 && rtfhe_twiddles_file_read(NULL, TN, ra, rb) != 0);
        FILE *f = fopen(path3, "r+b"); CHECK(f);
        fseek(f, 100, SEEK_SET); int c = fgetc(f); fseek(f, 100, SEEK_SET); fputc(c ^ 4, f); fclose(f);
        CHECK(rtfhe_twiddles_file_read(path3, TN, ra, rb) != 0);
        f = fopen(path3, "wb"); CHECK(f); fwrite("RTFHETW1", 1, 8, f); fclose(f);
        CHECK(rtfhe_twiddles_file_read(path3, TN, ra, rb) != 0);
        free(ta); free(tb); free(ra); free(rb);
    }
    free(key0); free(key1); free(k0b); free(k1b); free(bk); free(ksk); free(bk2); free(ksk2); free(ct); free(ct2);
    printf("host sanitizer walk ok\n");
    return 0;
}
