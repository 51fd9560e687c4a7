// hom_nand_ring.hpp -- the ring-level types of the `hom_nand` crate surface on top of the C ABI: Polynomial (rotate),
// TRLWERep, TRGSWRep / TRGSWRepF (cross = external product, cmux), BootstrappingKey and KeySwitchingKey containers.
// Same names and argument meaning as the reference; the external product runs on the GPU (rtfhe_external_product_batch),
// everything else here is host-side bookkeeping exactly as in the reference (wrapping u32 arithmetic).
//
//   Polynomial<T,N>::rotate          utils/src/math.rs:85-132
//   TRLWERep                         hom_nand/src/trlwe.rs:19-72, sample_extract_index :110-121, encrypt/decrypt :127-147
//   TRGSWRep / TRGSWRepF, cross/cmux hom_nand/src/trgsw.rs:23-108, 217-229, 264-321
//   BootstrappingKey                 hom_nand/src/tfhe.rs:116-135
//   KeySwitchingKey                  hom_nand/src/tlwe.rs:243-293
#pragma once

#include "hom_nand.hpp"

namespace hom_nand {

template <int N>
struct Polynomial {
    std::array<Torus32, N> c{};
    Torus32& operator[](int i) { return c[i]; }
    const Torus32& operator[](int i) const { return c[i]; }
    // multiply by X^n modulo X^N + 1 (n taken mod_floor 2N)
    Polynomial rotate(int n) const {
        int r = n % (2 * N); if (r < 0) r += 2 * N;
        Polynomial o;
        for (int i = 0; i < N; i++) {
            const int e = ((i - r) % (2 * N) + 2 * N) % (2 * N);
            o.c[i] = (e >= N) ? (0u - c[e - N]) : c[e];
        }
        return o;
    }
    Polynomial operator+(const Polynomial& o) const { Polynomial r; for (int i = 0; i < N; i++) r.c[i] = c[i] + o.c[i]; return r; }
    Polynomial operator-(const Polynomial& o) const { Polynomial r; for (int i = 0; i < N; i++) r.c[i] = c[i] - o.c[i]; return r; }
    bool operator==(const Polynomial& o) const { return c == o.c; }
};

template <int N>
struct TRLWERep {
    Polynomial<N> cipher_, p_key_;      // b(X), a(X)
    TRLWERep() = default;
    TRLWERep(const Polynomial<N>& cipher, const Polynomial<N>& p_key) : cipher_(cipher), p_key_(p_key) {}
    static TRLWERep trivial(const Polynomial<N>& text) { return TRLWERep(text, Polynomial<N>{}); }
    template <class F> TRLWERep map(F f) const { return TRLWERep(f(cipher_), f(p_key_)); }
    const Polynomial<N>& cipher() const { return cipher_; }
    const Polynomial<N>& p_key() const { return p_key_; }
    TRLWERep operator+(const TRLWERep& o) const { return TRLWERep(cipher_ + o.cipher_, p_key_ + o.p_key_); }
    TRLWERep operator-(const TRLWERep& o) const { return TRLWERep(cipher_ - o.cipher_, p_key_ - o.p_key_); }
    bool operator==(const TRLWERep& o) const { return cipher_ == o.cipher_ && p_key_ == o.p_key_; }
    // trlwe.rs:110-121
    TLWERep<N> sample_extract_index(int index) const {
        TLWERep<N> r;
        for (int i = 0; i < N; i++) r.p_key_[i] = (i <= index) ? p_key_[index - i] : (0u - p_key_[N + index - i]);
        r.cipher_ = cipher_[index];
        return r;
    }
    void to_flat(uint32_t* out) const { for (int i = 0; i < N; i++) { out[i] = cipher_[i]; out[N + i] = p_key_[i]; } }
    static TRLWERep from_flat(const uint32_t* in) { TRLWERep r; for (int i = 0; i < N; i++) { r.cipher_[i] = in[i]; r.p_key_[i] = in[N + i]; } return r; }
};

struct TRLWE {};
struct TRGSW {};

namespace detail {
struct Xs { uint64_t s; uint64_t next() { s += 0x9e3779b97f4a7c15ull; uint64_t z = s; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); } };
template <int N>
inline Polynomial<N> mul_binary_key(const Polynomial<N>& a, const std::array<Binary, N>& key) {   // exact negacyclic a * s
    Polynomial<N> r;
    for (int j = 0; j < N; j++) {
        if (key[j] != Binary::One) continue;
        for (int k = 0; k < N - j; k++) r.c[k + j] += a.c[k];
        for (int k = N - j; k < N; k++) r.c[k + j - N] -= a.c[k];
    }
    return r;
}
}  // namespace detail

// Crypto<Polynomial<Torus32>> for TRLWE (trlwe.rs:127-147): b = a*s + m + e, sigma = 2^-25; seeded uniform a, Gaussian-ish e
template <int N>
inline TRLWERep<N> trlwe_encrypto(const std::array<Binary, N>& key, const Polynomial<N>& msg, uint64_t seed) {
    detail::Xs g{seed};
    Polynomial<N> a, e;
    for (int k = 0; k < N; k++) a.c[k] = (uint32_t)(g.next() >> 40) << 8;                    // torus!(Uniform f32): 24 random bits
    for (int k = 0; k < N; k++) {
        double z = -6.0; for (int i = 0; i < 12; i++) z += (double)(g.next() >> 11) * (1.0 / 9007199254740992.0);
        e.c[k] = torus((float)z * (1.0f / 33554432.0f));
    }
    return TRLWERep<N>(detail::mul_binary_key<N>(a, key) + msg + e, a);
}
template <int N>
inline Polynomial<N> trlwe_decrypto(const std::array<Binary, N>& key, const TRLWERep<N>& rep) {
    return rep.cipher_ - detail::mul_binary_key<N>(rep.p_key_, key);
}

// TRGSWRep (torus form, trgsw.rs:23-26): 2l rows of (cipher, p_key)
template <int N>
struct TRGSWRep {
    static constexpr int ROWS = 2 * TRGSWHelper::L;
    std::array<Polynomial<N>, ROWS> cipher_, p_key_;
    // Crypto<i32> / Crypto<Binary> for TRGSW (trgsw.rs:217-229): produced by the library's key generator with n = 1
    // seed: TEST ONLY (rtfhe_keygen_with_keys_deterministic); nullptr = OS CSPRNG like the reference's thread_rng
    static TRGSWRep encrypto(const std::array<Binary, N>& s_key, Binary item, const uint64_t* seed = nullptr) {
        rtfhe_params p; rtfhe_default_params(&p); p.n = 1; p.N = N; p.nbit = 0; for (int v = N; v > 1; v >>= 1) p.nbit++;
        int32_t k0 = (int32_t)item; std::vector<int32_t> k1(N); for (int i = 0; i < N; i++) k1[i] = (int32_t)s_key[i];
        std::vector<uint32_t> flat((size_t)2 * ROWS * N);
        if (seed ? rtfhe_keygen_with_keys_deterministic(&p, *seed, &k0, k1.data(), flat.data(), nullptr)
                 : rtfhe_keygen_with_keys(&p, &k0, k1.data(), flat.data(), nullptr)) throw std::runtime_error("TRGSW::encrypto");
        TRGSWRep r;
        for (int j = 0; j < ROWS; j++) for (int k = 0; k < N; k++) { r.cipher_[j][k] = flat[(size_t)j * N + k]; r.p_key_[j][k] = flat[((size_t)ROWS + j) * N + k]; }
        return r;
    }
    void to_flat(uint32_t* out) const { for (int j = 0; j < ROWS; j++) for (int k = 0; k < N; k++) { out[(size_t)j * N + k] = cipher_[j][k]; out[((size_t)ROWS + j) * N + k] = p_key_[j][k]; } }
};

// TRGSWRepF (trgsw.rs:64-108): the transformed TRGSW lives on the device; cross / cmux run the engine's external product
template <int N>
class TRGSWRepF {
  public:
    explicit TRGSWRepF(const TRGSWRep<N>& t, int device = 0) {            // From<&TRGSWRep>, trgsw.rs:68-76
        rtfhe_params p; rtfhe_default_params(&p); p.n = 1; p.N = N; p.nbit = 0; for (int v = N; v > 1; v >>= 1) p.nbit++;
        rtfhe_ctx* c = nullptr;
        check(nullptr, rtfhe_ctx_create(&p, device, &c));
        ctx_.reset(c, rtfhe_ctx_destroy);
        std::vector<uint32_t> flat((size_t)2 * TRGSWRep<N>::ROWS * N);
        t.to_flat(flat.data());
        check(c, rtfhe_load_bk_torus(c, flat.data()));
    }
    // Cross for TRGSWRepF (trgsw.rs:264-306)
    TRLWERep<N> cross(const TRLWERep<N>& rhs) const {
        std::vector<uint32_t> in(2 * N), out(2 * N); rhs.to_flat(in.data());
        const int32_t idx = 0;
        check(ctx_.get(), rtfhe_external_product_batch(ctx_.get(), &idx, in.data(), out.data(), 1));
        return TRLWERep<N>::from_flat(out.data());
    }
    // TRGSW(i).cmux(rep_1, rep_0) = rep_i   (trgsw.rs:319-321): cross(rep_1 - rep_0) + rep_0
    TRLWERep<N> cmux(const TRLWERep<N>& rep_1, const TRLWERep<N>& rep_0) const { return cross(rep_1 - rep_0) + rep_0; }

  private:
    static void check(rtfhe_ctx* c, int rc) { if (rc) throw std::runtime_error(std::string("rtfhe: ") + rtfhe_last_error(c)); }
    std::shared_ptr<rtfhe_ctx> ctx_;
};

// BootstrappingKey<PRE_N, N>(Vec<TRGSW>) (tfhe.rs:116-135) and KeySwitchingKey<N, M> (tlwe.rs:243-293): flat containers in the
// ABI's layouts, generated for caller-supplied secret keys; `raw()` feeds rtfhe_load_bk_torus, `raw_ref()` rtfhe_load_ksk_ref.
template <int PRE_N, int N>
struct BootstrappingKey {
    std::vector<uint32_t> flat;       // [PRE_N][2][2l][N]
    // seed: TEST ONLY (deterministic, not secure); nullptr = OS CSPRNG
    BootstrappingKey(const std::array<Binary, PRE_N>& s_key_tlwe, const std::array<Binary, N>& s_key, const uint64_t* seed = nullptr) {
        rtfhe_params p; rtfhe_default_params(&p); p.n = PRE_N; p.N = N; p.nbit = 0; for (int v = N; v > 1; v >>= 1) p.nbit++;
        std::vector<int32_t> k0(PRE_N), k1(N);
        for (int i = 0; i < PRE_N; i++) k0[i] = (int32_t)s_key_tlwe[i];
        for (int i = 0; i < N; i++) k1[i] = (int32_t)s_key[i];
        flat.resize((size_t)PRE_N * 2 * 2 * p.l * N);
        if (seed ? rtfhe_keygen_with_keys_deterministic(&p, *seed, k0.data(), k1.data(), flat.data(), nullptr)
                 : rtfhe_keygen_with_keys(&p, k0.data(), k1.data(), flat.data(), nullptr)) throw std::runtime_error("BootstrappingKey::new");
    }
    size_t size() const { return PRE_N; }
    const uint32_t* trgsw(int i) const { return flat.data() + (size_t)i * 2 * 2 * TRGSWHelper::L * N; }   // iter(): i-th TRGSW
    const uint32_t* raw() const { return flat.data(); }
};

template <int N, int M>
struct KeySwitchingKey {
    // the reference's container as it stands: Vec<[[TLWERep<M>; IKS_T]; IKS_L]> (tlwe.rs:243-245), IKS_T = 2^IKS_BASEBIT = 4 entries per
    // level of which identity_key_switch only ever reads t = 1 .. 3; flat [N][IKS_L][IKS_T][M + 1] = what rtfhe_load_ksk_ref takes
    std::vector<uint32_t> flat;
    KeySwitchingKey(const std::array<Binary, N>& pre_s_key, const std::array<Binary, M>& next_s_key, const uint64_t* seed = nullptr) {
        rtfhe_params p; rtfhe_default_params(&p); p.n = M; p.N = N; p.nbit = 0; for (int v = N; v > 1; v >>= 1) p.nbit++;
        std::vector<int32_t> k0(M), k1(N);
        for (int i = 0; i < M; i++) k0[i] = (int32_t)next_s_key[i];
        for (int i = 0; i < N; i++) k1[i] = (int32_t)pre_s_key[i];
        std::vector<uint32_t> compact((size_t)N * TLWEHelper::IKS_L * (TLWEHelper::IKS_T - 1) * (M + 1));
        flat.resize((size_t)N * TLWEHelper::IKS_L * TLWEHelper::IKS_T * (M + 1));
        if (seed ? rtfhe_keygen_with_keys_deterministic(&p, *seed, k0.data(), k1.data(), nullptr, compact.data())
                 : rtfhe_keygen_with_keys(&p, k0.data(), k1.data(), nullptr, compact.data())) throw std::runtime_error("KeySwitchingKey::new");
        if (seed ? rtfhe_ksk_expand_ref_deterministic(&p, *seed ^ 0x4b534b, k0.data(), k1.data(), compact.data(), flat.data())
                 : rtfhe_ksk_expand_ref(&p, k0.data(), k1.data(), compact.data(), flat.data())) throw std::runtime_error("KeySwitchingKey::new");
    }
    // get(i, l, t) = KS[i][l][t-1] = TLWE(t * s_i / 2^(bit (l+1))), t = 1 .. IKS_T   (tlwe.rs:281-283)
    TLWERep<M> get(int i, int l, int t) const {
        if (i < 0 || i >= N || l < 0 || l >= TLWEHelper::IKS_L || t < 1 || t > TLWEHelper::IKS_T) throw std::out_of_range("KeySwitchingKey::get");
        return TLWERep<M>::from_flat(flat.data() + (((size_t)i * TLWEHelper::IKS_L + l) * TLWEHelper::IKS_T + (t - 1)) * (M + 1));
    }
    const uint32_t* raw_ref() const { return flat.data(); }       // feeds rtfhe_load_ksk_ref
    // Round 2's raw() returned the 3-entry layout of rtfhe_load_ksk; since round 3 the container is the reference's 4-entry one.  An out-of-tree
    // caller that still writes rtfhe_load_ksk(ctx, ksk.raw()) must not compile into a mis-strided key: the name is kept, deleted.
    const uint32_t* raw() const = delete;
};

}  // namespace hom_nand
