// C++ counterpart of the reference's hom_nand/examples/homnand-bench.rs:7-137 (BASELINE config 1): key generation,
// then for NAND/AND/OR/XOR (4 input pairs each) and NOT (2) encrypt, time the gate, decrypt, check the truth table.
// Built and run by tests/test_cpp_host.py on the GPU box:  g++ -std=c++17 homnand_bench.cpp -L rustfhe_amd -lrtfhe_hip
#include <chrono>
#include <cstdio>
#include <random>

#include "../../rustfhe_amd/host/hom_nand_ring.hpp"

using namespace hom_nand;

int main() {
    constexpr int TLWE_N = TLWEHelper::N, TRLWE_N = 1 << TFHEHelper::NBIT;
    std::mt19937_64 rng(2021);
    std::array<Binary, TLWE_N> s0;
    std::array<Binary, TRLWE_N> s1;
    for (auto& b : s0) b = (rng() & 1) ? Binary::One : Binary::Zero;
    for (auto& b : s1) b = (rng() & 1) ? Binary::One : Binary::Zero;
    try {
        TFHE<TLWE_N, TRLWE_N> tfhe(s0, s1);
        // production forms: key material and ciphertext randomness from the OS CSPRNG, like the reference's thread_rng
        auto enc = [&](Binary b) { return Cryptor::encrypto<TLWE_N>(TLWE{}, s0, b); };
        {   // two encryptions of the same bit never share mask or noise; the seeded TEST-ONLY form is reproducible
            auto e1 = enc(Binary::One), e2 = enc(Binary::One);
            if (e1 == e2) { std::printf("secure encryption repeated itself\n"); return 1; }
            auto d1 = Cryptor::encrypto_deterministic<TLWE_N>(TLWE{}, s0, Binary::One, 100), d2 = Cryptor::encrypto_deterministic<TLWE_N>(TLWE{}, s0, Binary::One, 100);
            if (!(d1 == d2)) { std::printf("deterministic encryption not reproducible\n"); return 1; }
        }
        struct G { const char* title; int op; int tt[4]; };
        const G gates[] = {{"nand", 0, {1, 1, 1, 0}}, {"and", 1, {0, 0, 0, 1}}, {"or", 2, {0, 1, 1, 1}}, {"xor", 3, {0, 1, 1, 0}}};
        int bad = 0;
        for (const G& g : gates)
            for (int i = 0; i < 4; i++) {
                const Binary in0 = (i & 1) ? Binary::One : Binary::Zero, in1 = (i & 2) ? Binary::One : Binary::Zero;
                auto a = enc(in0), b = enc(in1);
                auto t0 = std::chrono::steady_clock::now();
                TLWERep<TLWE_N> r = g.op == 0 ? tfhe.hom_nand(a, b) : g.op == 1 ? tfhe.hom_and(a, b) : g.op == 2 ? tfhe.hom_or(a, b) : tfhe.hom_xor(a, b);
                auto us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
                std::printf("%s %d %d: %lld micro-seconds\n", g.title, (int)in0, (int)in1, (long long)us);
                if ((int)Cryptor::decrypto<TLWE_N>(TLWE{}, s0, r) != g.tt[i]) { std::printf("  WRONG\n"); bad++; }
            }
        for (int i = 0; i < 2; i++) {
            auto r = tfhe.hom_not(enc(i ? Binary::One : Binary::Zero));
            if ((int)Cryptor::decrypto<TLWE_N>(TLWE{}, s0, r) != 1 - i) { std::printf("not %d WRONG\n", i); bad++; }
        }
        // Logip's derived gates through NAND only (nander/src/lib.rs:19-38) and a mux
        const Logip<TLWERep<TLWE_N>>& lp = tfhe;
        auto one = enc(Binary::One), zero = enc(Binary::Zero);
        if ((int)Cryptor::decrypto<TLWE_N>(TLWE{}, s0, lp.Logip<TLWERep<TLWE_N>>::xor_(one, zero)) != 1) { std::printf("logip xor WRONG\n"); bad++; }
        if ((int)Cryptor::decrypto<TLWE_N>(TLWE{}, s0, tfhe.hom_mux(one, zero, one)) != 1) { std::printf("mux WRONG\n"); bad++; }
        // a batch: same answers as one by one
        std::vector<TLWERep<TLWE_N>> va, vb;
        for (int i = 0; i < 6; i++) { va.push_back(enc((i & 1) ? Binary::One : Binary::Zero)); vb.push_back(enc((i & 2) ? Binary::One : Binary::Zero)); }
        auto vr = tfhe.hom_nand_batch(va, vb);
        for (int i = 0; i < 6; i++) if (!(vr[i] == tfhe.hom_nand(va[i], vb[i]))) { std::printf("batch != single at %d\n", i); bad++; }
        // ring-level surface: rotate KATs (utils/src/math.rs:75-84 on N = 8 analogue), TRLWE -> sample extract, TRGSW cmux
        {
            Polynomial<8> q; for (int i = 0; i < 8; i++) q[i] = (uint32_t)(i + 1);
            const Polynomial<8> r1 = q.rotate(1), rm1 = q.rotate(-1), r8 = q.rotate(8), r16 = q.rotate(16);
            if (r1[0] != (uint32_t)-8 || r1[1] != 1 || rm1[7] != (uint32_t)-1 || rm1[0] != 2 || r8[3] != (uint32_t)-4 || !(r16 == q)) { std::printf("rotate WRONG\n"); bad++; }
            Polynomial<TRLWE_N> m1, m0;
            for (int k = 0; k < TRLWE_N; k++) { m1[k] = 0x20000000u; m0[k] = 0xE0000000u; }
            auto rep1 = trlwe_encrypto<TRLWE_N>(s1, m1, 501), rep0 = trlwe_encrypto<TRLWE_N>(s1, m0, 502);
            auto t1 = rep1.sample_extract_index(0);
            if ((int)Cryptor::decrypto<TRLWE_N>(TLWE{}, s1, t1) != 1) { std::printf("sample_extract WRONG\n"); bad++; }
            for (int bit = 0; bit < 2; bit++) {
                TRGSWRepF<TRLWE_N> c(TRGSWRep<TRLWE_N>::encrypto(s1, bit ? Binary::One : Binary::Zero));
                auto sel = c.cmux(rep1, rep0);                                   // TRGSW(i).cmux(rep_1, rep_0) = rep_i
                auto ph = trlwe_decrypto<TRLWE_N>(s1, sel);
                int wrong = 0;
                for (int k = 0; k < TRLWE_N; k++) if ((int)TLWEHelper::torus2binary(ph[k]) != bit) wrong++;
                if (wrong) { std::printf("cmux(%d) WRONG in %d coefficients\n", bit, wrong); bad++; }
            }
            const uint64_t ks_seed = 7;
            KeySwitchingKey<TRLWE_N, TLWE_N> ksk(s1, s0, &ks_seed);
            auto row = ksk.get(3, 0, 1);                                         // TLWE(1 * s1[3] / 4)
            Torus32 s = 0; for (int i = 0; i < TLWE_N; i++) if (s0[i] == Binary::One) s += row.p_key()[i];
            const int32_t err = (int32_t)(row.cipher() - s - (s1[3] == Binary::One ? 0x40000000u : 0u));
            if (err > (1 << 21) || err < -(1 << 21)) { std::printf("KeySwitchingKey::get WRONG\n"); bad++; }
            auto row4 = ksk.get(5, 1, TLWEHelper::IKS_T);                        // the reference's 4th entry: TLWE(4 * s1[5] / 16)
            Torus32 s4 = 0; for (int i = 0; i < TLWE_N; i++) if (s0[i] == Binary::One) s4 += row4.p_key()[i];
            const int32_t err4 = (int32_t)(row4.cipher() - s4 - (s1[5] == Binary::One ? 0x40000000u : 0u));
            if (err4 > (1 << 21) || err4 < -(1 << 21)) { std::printf("KeySwitchingKey::get(t = IKS_T) WRONG\n"); bad++; }
        }
        std::printf(bad ? "FAILED %d\n" : "all truth tables ok\n", bad);
        return bad ? 1 : 0;
    } catch (const std::exception& e) {
        std::printf("exception: %s\n", e.what());
        return 2;
    }
}
