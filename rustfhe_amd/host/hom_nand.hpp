// hom_nand.hpp -- C++ host-side mirror of the reference's `hom_nand` crate surface for the gate path, on top
// of the C ABI (include/rtfhe.h).  The reference host is Rust; no Rust toolchain exists in this image, so the
// compiled-host mirror is C++ (header only, links against librtfhe_hip.so).  Same names, same argument meaning;
// where the reference panics this throws std::runtime_error.
//
//   Binary                      utils/src/math.rs:362-366
//   Torus32 / torus()           utils/src/math.rs:489-539, 691-696
//   TLWERep<N>                  hom_nand/src/tlwe.rs:19-79 (+ ops :88-171)
//   TLWEHelper                  hom_nand/src/tlwe.rs:173-195
//   Cryptor / TLWE strategy     hom_nand/src/digest.rs:19-34, hom_nand/src/tlwe.rs:199-241
//   TFHE<TLWE_N, TRLWE_N>       hom_nand/src/tfhe.rs:9-71   (+ hom_*_batch, the reason for the engine)
//   Logip                       nander/src/lib.rs:19-62
#pragma once

#include <array>
#include <cmath>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rtfhe.h"

namespace hom_nand {

enum class Binary : int { Zero = 0, One = 1 };

using Torus32 = uint32_t;   // Decimal<u32>: value x / 2^32, wrapping arithmetic

inline Torus32 torus(float v) {   // torus!(f32), utils/src/math.rs:691-696
    volatile float w = v - std::floor(v);
    volatile float fr = w - std::trunc(w);
    volatile float x = fr * 4294967296.0f;
    if (!(x > 0.0f)) return 0u;
    if (x >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)x;
}

struct TLWEHelper {
    static constexpr int N = 635;
    static constexpr int IKS_L = 8, IKS_BASEBIT = 2, IKS_T = 4;
    static Torus32 binary2torus(Binary b) { return torus(b == Binary::One ? 1.0f / 8.0f : -1.0f / 8.0f); }
    static Binary torus2binary(Torus32 t) { return ((float)t * (1.0f / 4294967296.0f) < 0.5f) ? Binary::One : Binary::Zero; }
};
struct TRLWEHelper { static constexpr int N = 1024; };
struct TRGSWHelper { static constexpr int BGBIT = 6, BG = 64, L = 3; };
struct TFHEHelper { static constexpr int NBIT = 10; static constexpr float COEF = 1.0f / 8.0f; };

// flat layout of the ABI: a[0..N) then b
template <int N>
struct TLWERep {
    Torus32 cipher_ = 0;                 // b
    std::array<Torus32, N> p_key_{};     // a
    TLWERep() = default;
    TLWERep(Torus32 cipher, const std::array<Torus32, N>& p_key) : cipher_(cipher), p_key_(p_key) {}
    static TLWERep trivial(Torus32 text) { TLWERep r; r.cipher_ = text; return r; }
    static TLWERep logic_true() { return trivial(TLWEHelper::binary2torus(Binary::One)); }    // AsLogic, tlwe.rs:80-87
    static TLWERep logic_false() { return trivial(TLWEHelper::binary2torus(Binary::Zero)); }
    const Torus32& cipher() const { return cipher_; }
    const std::array<Torus32, N>& p_key() const { return p_key_; }
    TLWERep operator+(const TLWERep& o) const { TLWERep r = *this; for (int i = 0; i < N; i++) r.p_key_[i] += o.p_key_[i]; r.cipher_ += o.cipher_; return r; }
    TLWERep operator-(const TLWERep& o) const { TLWERep r = *this; for (int i = 0; i < N; i++) r.p_key_[i] -= o.p_key_[i]; r.cipher_ -= o.cipher_; return r; }
    TLWERep operator-() const { TLWERep r; for (int i = 0; i < N; i++) r.p_key_[i] = 0u - p_key_[i]; r.cipher_ = 0u - cipher_; return r; }
    TLWERep operator*(int32_t k) const { TLWERep r; for (int i = 0; i < N; i++) r.p_key_[i] = p_key_[i] * (uint32_t)k; r.cipher_ = cipher_ * (uint32_t)k; return r; }
    bool operator==(const TLWERep& o) const { return cipher_ == o.cipher_ && p_key_ == o.p_key_; }
    void to_flat(uint32_t* out) const { for (int i = 0; i < N; i++) out[i] = p_key_[i]; out[N] = cipher_; }
    static TLWERep from_flat(const uint32_t* in) { TLWERep r; for (int i = 0; i < N; i++) r.p_key_[i] = in[i]; r.cipher_ = in[N]; return r; }
};

struct TLWE {};   // strategy marker, hom_nand/src/tlwe.rs:10

// Cryptor::encrypto / decrypto (digest.rs:19-34) for the TLWE strategy.  encrypto draws from the OS CSPRNG like the
// reference's thread_rng; encrypto_deterministic(.., seed) is the seeded TEST-ONLY form (not secure).
struct Cryptor {
    template <int N>
    static TLWERep<N> encrypto(TLWE, const std::array<Binary, N>& s_key, Binary item) {
        rtfhe_params p; rtfhe_default_params(&p); p.n = N;
        std::vector<int32_t> k(N); for (int i = 0; i < N; i++) k[i] = (int32_t)s_key[i];
        uint8_t bit = item == Binary::One; std::vector<uint32_t> out(N + 1);
        if (rtfhe_tlwe_encrypt_bits(&p, k.data(), &bit, out.data(), 1)) throw std::runtime_error("rtfhe_tlwe_encrypt_bits");
        return TLWERep<N>::from_flat(out.data());
    }
    template <int N>
    static TLWERep<N> encrypto_deterministic(TLWE, const std::array<Binary, N>& s_key, Binary item, uint64_t seed) {
        rtfhe_params p; rtfhe_default_params(&p); p.n = N;
        std::vector<int32_t> k(N); for (int i = 0; i < N; i++) k[i] = (int32_t)s_key[i];
        uint8_t bit = item == Binary::One; std::vector<uint32_t> out(N + 1);
        if (rtfhe_tlwe_encrypt_bits_deterministic(&p, k.data(), seed, &bit, out.data(), 1)) throw std::runtime_error("rtfhe_tlwe_encrypt_bits_deterministic");
        return TLWERep<N>::from_flat(out.data());
    }
    template <int N>
    static Binary decrypto(TLWE, const std::array<Binary, N>& s_key, const TLWERep<N>& rep) {
        Torus32 s = 0;   // tlwe.rs:230-240
        for (int i = 0; i < N; i++) if (s_key[i] == Binary::One) s += rep.p_key_[i];
        return TLWEHelper::torus2binary(rep.cipher_ - s);
    }
};

// nander/src/lib.rs:19-38
template <class R>
struct Logip {
    virtual ~Logip() = default;
    virtual R nand(const R& l, const R& r) const = 0;
    virtual R not_(const R& b) const { return nand(b, b); }
    virtual R and_(const R& l, const R& r) const { return not_(nand(l, r)); }
    virtual R or_(const R& l, const R& r) const { return nand(not_(l), not_(r)); }
    virtual R xor_(const R& l, const R& r) const { R x = nand(l, r); return nand(nand(l, x), nand(x, r)); }
};

template <int TLWE_N = TLWEHelper::N, int TRLWE_N = TRLWEHelper::N>
class TFHE : public Logip<TLWERep<TLWE_N>> {
  public:
    using Rep = TLWERep<TLWE_N>;
    // TFHE::new (tfhe.rs:21-25): generates KSK and BK for the given secret keys and loads them on `device`.  The key
    // material's masks and noise come from the OS CSPRNG (the reference draws from thread_rng).
    TFHE(const std::array<Binary, TLWE_N>& s_key_tlwelv0, const std::array<Binary, TRLWE_N>& s_key_tlwelv1, int device = 0)
        : TFHE(s_key_tlwelv0, s_key_tlwelv1, device, nullptr) {}
    // TEST ONLY (not secure): key material reproducible from `key_seed` (rtfhe_keygen_with_keys_deterministic)
    static TFHE new_deterministic(const std::array<Binary, TLWE_N>& s_key_tlwelv0, const std::array<Binary, TRLWE_N>& s_key_tlwelv1,
                                  uint64_t key_seed, int device = 0) {
        return TFHE(s_key_tlwelv0, s_key_tlwelv1, device, &key_seed);
    }

  private:
    TFHE(const std::array<Binary, TLWE_N>& s_key_tlwelv0, const std::array<Binary, TRLWE_N>& s_key_tlwelv1, int device,
         const uint64_t* key_seed) {
        rtfhe_default_params(&p_);
        p_.n = TLWE_N; p_.N = TRLWE_N; p_.nbit = 0;
        for (int v = TRLWE_N; v > 1; v >>= 1) p_.nbit++;
        // rtfhe_keygen draws its own secret keys; to honour caller-supplied keys the key material is generated against them
        std::vector<int32_t> k0(TLWE_N), k1(TRLWE_N);
        for (int i = 0; i < TLWE_N; i++) k0[i] = (int32_t)s_key_tlwelv0[i];
        for (int i = 0; i < TRLWE_N; i++) k1[i] = (int32_t)s_key_tlwelv1[i];
        std::vector<uint32_t> bk((size_t)TLWE_N * 2 * 2 * p_.l * TRLWE_N);
        std::vector<uint32_t> ksk((size_t)TRLWE_N * p_.ks_t * ((1 << p_.ks_basebit) - 1) * (TLWE_N + 1));
        check(nullptr, key_seed ? rtfhe_keygen_with_keys_deterministic(&p_, *key_seed, k0.data(), k1.data(), bk.data(), ksk.data())
                                : rtfhe_keygen_with_keys(&p_, k0.data(), k1.data(), bk.data(), ksk.data()));
        // the key-switching key goes in through the reference's own container shape, [[TLWERep; IKS_T = 4]; IKS_L = 8] per coefficient
        // (tlwe.rs:243-245), entry t = 4 included as KeySwitchingKey::new fills it (tlwe.rs:252-274)
        std::vector<uint32_t> ksk_ref((size_t)TRLWE_N * p_.ks_t * (1 << p_.ks_basebit) * (TLWE_N + 1));
        check(nullptr, key_seed ? rtfhe_ksk_expand_ref_deterministic(&p_, *key_seed ^ 0x4b534b, k0.data(), k1.data(), ksk.data(), ksk_ref.data())
                                : rtfhe_ksk_expand_ref(&p_, k0.data(), k1.data(), ksk.data(), ksk_ref.data()));
        rtfhe_ctx* c = nullptr;
        check(nullptr, rtfhe_ctx_create(&p_, device, &c));
        ctx_.reset(c, rtfhe_ctx_destroy);
        check(c, rtfhe_load_bk_torus(c, bk.data()));
        check(c, rtfhe_load_ksk_ref(c, ksk_ref.data()));
    }

  public:
    Rep hom_nand(const Rep& a, const Rep& b) const { return one(RTFHE_NAND, a, &b); }   // tfhe.rs:41-47
    Rep hom_and(const Rep& a, const Rep& b) const { return one(RTFHE_AND, a, &b); }     // tfhe.rs:48-54
    Rep hom_or(const Rep& a, const Rep& b) const { return one(RTFHE_OR, a, &b); }       // tfhe.rs:55-61
    Rep hom_xor(const Rep& a, const Rep& b) const { return one(RTFHE_XOR, a, &b); }     // tfhe.rs:62-68
    Rep hom_not(const Rep& a) const { return one(RTFHE_NOT, a, nullptr); }              // tfhe.rs:69-71
    Rep hom_mux(const Rep& c, const Rep& in0, const Rep& in1) const {                   // tfhe.rs:27-40
        std::vector<uint32_t> fc(W), f0(W), f1(W), o(W);
        c.to_flat(fc.data()); in0.to_flat(f0.data()); in1.to_flat(f1.data());
        check(ctx_.get(), rtfhe_mux_batch(ctx_.get(), fc.data(), f0.data(), f1.data(), o.data(), 1));
        return Rep::from_flat(o.data());
    }
    // batch forms: `count` independent gates in one launch
    std::vector<Rep> hom_batch(int op, const std::vector<Rep>& a, const std::vector<Rep>* b) const {
        if (b && b->size() != a.size()) throw std::runtime_error("hom_batch: length mismatch");
        std::vector<uint32_t> f0(a.size() * W), f1(b ? a.size() * W : 0), o(a.size() * W);
        for (size_t g = 0; g < a.size(); g++) { a[g].to_flat(&f0[g * W]); if (b) (*b)[g].to_flat(&f1[g * W]); }
        check(ctx_.get(), rtfhe_gate_batch(ctx_.get(), op, f0.data(), b ? f1.data() : nullptr, o.data(), a.size()));
        std::vector<Rep> r(a.size());
        for (size_t g = 0; g < a.size(); g++) r[g] = Rep::from_flat(&o[g * W]);
        return r;
    }
    std::vector<Rep> hom_nand_batch(const std::vector<Rep>& a, const std::vector<Rep>& b) const { return hom_batch(RTFHE_NAND, a, &b); }

    // Logip for TFHE (nander/src/lib.rs:40-62)
    Rep nand(const Rep& l, const Rep& r) const override { return hom_nand(l, r); }
    Rep not_(const Rep& b) const override { return hom_not(b); }
    Rep and_(const Rep& l, const Rep& r) const override { return hom_and(l, r); }
    Rep or_(const Rep& l, const Rep& r) const override { return hom_or(l, r); }
    Rep xor_(const Rep& l, const Rep& r) const override { return hom_xor(l, r); }

    rtfhe_ctx* raw() const { return ctx_.get(); }

  private:
    static constexpr size_t W = TLWE_N + 1;
    static void check(rtfhe_ctx* c, int rc) {
        if (rc != 0) throw std::runtime_error(std::string("rtfhe: ") + rtfhe_last_error(c) + " (code " + std::to_string(rc) + ")");
    }
    Rep one(int op, const Rep& a, const Rep* b) const {
        std::vector<uint32_t> f0(W), f1(W), o(W);
        a.to_flat(f0.data());
        if (b) b->to_flat(f1.data());
        check(ctx_.get(), rtfhe_gate_batch(ctx_.get(), op, f0.data(), b ? f1.data() : nullptr, o.data(), 1));
        return Rep::from_flat(o.data());
    }
    rtfhe_params p_{};
    std::shared_ptr<rtfhe_ctx> ctx_;
};

}  // namespace hom_nand
