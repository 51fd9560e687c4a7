/* Sanitizer driver for the CPU oracle ("ASan/UBSan on the CPU oracle").  Compiled together with
 * oracle/tfhe_oracle.c under -fsanitize=address,undefined by tests/test_oracle_sanitizers.py; walks every part of the
 * restatement once on a small TLWE dimension (n = 6, N = 1024): key generation, all gates + MUX through both multiply
 * backends, the threaded batch entry, rotate / decomposition edge cases.  Exit code 0 = every gate decrypted correctly
 * and the sanitizers stayed silent (they abort the process otherwise). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "tfhe_oracle.h"

int main(void) {
    orc_params p;
    orc_default_params(&p);
    p.n = 6;
    const int32_t N = p.N, n = p.n, w = n + 1;
    orc_plan *pl = orc_plan_new(N);
    orc_rng r;
    orc_rng_seed(&r, 99);
    int32_t *key0 = malloc(sizeof(int32_t) * (size_t)n), *key1 = malloc(sizeof(int32_t) * (size_t)N);
    orc_gen_binary_key(&r, n, key0);
    orc_gen_binary_key(&r, N, key1);
    const size_t bk_words = (size_t)n * 2 * 2 * p.l * N, ksk_words = (size_t)N * p.ks_t * 3 * w;
    uint32_t *bk_t = malloc(4 * bk_words), *ksk = malloc(4 * ksk_words);
    double *bk_f = malloc(8 * bk_words);
    orc_bk_gen(&r, pl, &p, key0, key1, 1.0f / 33554432.0f, bk_t);
    orc_ksk_gen(&r, &p, key1, key0, 1.0f / 32768.0f, ksk);
    orc_trgsw_to_fft(pl, bk_t, bk_f, bk_words / (size_t)N);

    int bad = 0;
    uint32_t *c[2], *out = malloc(4 * (size_t)w);
    for (int b = 0; b < 2; b++) {
        c[b] = malloc(4 * (size_t)w);
        orc_tlwe_encrypt(&r, n, key0, orc_binary2torus(b), 1.0f / 32768.0f, c[b]);
    }
    static const int tt[4][4] = {{1, 1, 1, 0}, {0, 0, 0, 1}, {0, 1, 1, 1}, {0, 1, 1, 0}};     /* NAND AND OR XOR */
    for (int backend = 0; backend < 2; backend++) {
        orc_plan_set_backend(pl, backend ? ORC_BACKEND_EXACT_INT : ORC_BACKEND_FFT64_MIRROR);
        for (int op = ORC_NAND; op <= ORC_XOR; op++)
            for (int i = 0; i < 4; i++) {
                orc_gate(&p, pl, op, bk_f, bk_t, ksk, c[i & 1], c[i >> 1], out);
                if (orc_torus2binary(orc_tlwe_phase(n, key0, out)) != tt[op][i]) { printf("gate %d input %d backend %d WRONG\n", op, i, backend); bad++; }
            }
        orc_gate(&p, pl, ORC_NOT, bk_f, bk_t, ksk, c[1], NULL, out);
        if (orc_torus2binary(orc_tlwe_phase(n, key0, out)) != 0) { printf("NOT wrong\n"); bad++; }
        orc_mux(&p, pl, bk_f, bk_t, ksk, c[1], c[0], c[1], out);
        if (orc_torus2binary(orc_tlwe_phase(n, key0, out)) != 1) { printf("MUX wrong\n"); bad++; }
    }
    orc_plan_set_backend(pl, ORC_BACKEND_FFT64_MIRROR);
    /* threaded entry: 5 gates on 3 threads == the same gates one by one */
    uint32_t *in0 = malloc(4 * (size_t)w * 5), *in1 = malloc(4 * (size_t)w * 5), *o5 = malloc(4 * (size_t)w * 5);
    for (int g = 0; g < 5; g++) { memcpy(in0 + g * w, c[g & 1], 4 * (size_t)w); memcpy(in1 + g * w, c[(g >> 1) & 1], 4 * (size_t)w); }
    orc_gate_batch_mt(&p, ORC_BACKEND_FFT64_MIRROR, ORC_NAND, bk_f, NULL, ksk, in0, in1, o5, 5, 3);
    for (int g = 0; g < 5; g++) {
        orc_gate(&p, pl, ORC_NAND, bk_f, NULL, ksk, in0 + g * w, in1 + g * w, out);
        if (memcmp(out, o5 + g * w, 4 * (size_t)w)) { printf("mt gate %d differs\n", g); bad++; }
    }
    /* rotate by every class of amount, decomposition of extreme words */
    uint32_t q[8] = {1, 2, 3, 4, 5, 6, 7, 8}, qr[8];
    const int32_t amounts[] = {0, 1, 7, 8, 9, 15, 16, 17, -1, -8, -9, -16, -17, 2147483647, -2147483647 - 1};
    for (size_t k = 0; k < sizeof(amounts) / sizeof(amounts[0]); k++) orc_rotate_u32(8, q, amounts[k], qr);
    int32_t dig[3];
    const uint32_t words[] = {0u, 0xffffffffu, 0x80000000u, 0x7fffffffu, 0x02084000u, 0xfdf7bfffu};
    for (size_t k = 0; k < sizeof(words) / sizeof(words[0]); k++) orc_decomp_scalar(words[k], 6, orc_make_decomp_mask(3, 6), 3, dig);
    printf(bad ? "FAILED %d\n" : "oracle sanitizer walk ok\n", bad);
    free(in0); free(in1); free(o5); free(c[0]); free(c[1]); free(out); free(bk_t); free(bk_f); free(ksk); free(key0); free(key1);
    orc_plan_free(pl);
    return bad ? 1 : 0;
}
