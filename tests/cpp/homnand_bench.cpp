// C++ counterpart of the reference's hom_nand/examples/homnand-bench.rs:7-137 (BASELINE config 1): key generation,
// then for NAND/AND/OR/XOR (4 input pairs each) and NOT (2) encrypt, time the gate, decrypt, check the truth table.
// Built and run by tests/test_cpp_host.py on the GPU box:  g++ -std=c++17 homnand_bench.cpp -L rustfhe_amd -lrtfhe_hip
#include <chrono>
#include <cstdio>
#include <random>

#include "../../rustfhe_amd/host/hom_nand.hpp"

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
        uint64_t seed = 100;
        auto enc = [&](Binary b) { return Cryptor::encrypto<TLWE_N>(TLWE{}, s0, b, seed++); };
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
        std::printf(bad ? "FAILED %d\n" : "all truth tables ok\n", bad);
        return bad ? 1 : 0;
    } catch (const std::exception& e) {
        std::printf("exception: %s\n", e.what());
        return 2;
    }
}
