"""What an external product MEANS, checked on the device without the restated oracle in the loop: the reference's own
property tests of TRGSW x TRLWE (hom_nand/src/trgsw.rs:365-457) through rtfhe_external_product_batch with keys the PRODUCT generated.

The bootstrapping key is n TRGSW encryptions of the bits of the lvl0 key under the lvl1 key (tfhe.rs:119-126), so bk[i] is TRGSW(1) where
key0[i] = 1 and TRGSW(0) where key0[i] = 0 -- the two operands the reference's tests build by hand:
  trgsw_cross (:365-393):  TRGSW(1) (x) TRLWE(m) decrypts to m within 2e-3 at every coefficient (m = 1/2 at even, 1/4 at odd positions);
  trgsw_cmux  (:395-426):  TRGSW(i).cmux(rep_1, rep_0) = cross(rep_1 - rep_0) + rep_0 decrypts to pol_i (all-One / all-Zero polynomials).
TRLWE encryption and decryption are done here in numpy from their definitions (trlwe.rs:127-147: b = a (*) s + m + e, phase = b - a (*) s, noise
2^-25), so nothing of the oracle's external product, decomposition or transform is involved: a wrong gadget, a swapped component or a wrong
sign in the device's product fails these tests whatever the oracle restates.  Every backend, N = 1024 and N = 2048.

Tolerance: the reference states 2e-3 at its N = 1024.  What the bound covers is the gadget decomposition's rounding error (18 of 32 bits kept, with
the biased offset of make_decomp_mask) summed over the ~N/2 set bits of the key: it is a systematic ~1e-3 at N = 1024 and ~2e-3 at
N = 2048 (measured: 0.0017-0.0023 at every coefficient), so the bound scales with N here: 2e-3 * N / 1024."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def negacyclic_by_binary(a_u32, s_bits):
    """a (*) s in Z_{2^32}[X] / (X^N + 1), s a 0/1 polynomial (exact: 64-bit integers, reduced at the end)"""
    N = a_u32.size
    full = np.convolve(a_u32.astype(np.int64), s_bits.astype(np.int64))          # < 2^32 * N < 2^63
    res = full[:N].copy()
    res[:N - 1] -= full[N:]
    return (res & 0xFFFFFFFF).astype(np.uint32)


def trlwe_encrypt(rng, key1, m_u32, alpha=2.0 ** -25):
    N = key1.size
    a = rng.integers(0, 2 ** 32, N, dtype=np.uint64).astype(np.uint32)
    e = np.rint(rng.normal(0.0, alpha, N) * 2.0 ** 32).astype(np.int64)
    b = (negacyclic_by_binary(a, key1).astype(np.int64) + m_u32.astype(np.int64) + e) & 0xFFFFFFFF
    return np.stack([b.astype(np.uint32), a])                                    # (cipher, p_key): trlwe.rs:136


def trlwe_phase(key1, ct):
    return (ct[0].astype(np.int64) - negacyclic_by_binary(ct[1], key1).astype(np.int64)) & 0xFFFFFFFF


def torus_distance(x_u32, y_u32):
    d = (x_u32.astype(np.int64) - y_u32.astype(np.int64)) & 0xFFFFFFFF
    d = np.minimum(d, 2 ** 32 - d)
    return d / 2.0 ** 32


@pytest.mark.parametrize("N, n, backends", [(1024, 40, ("mirror", "ntt", "xfft")), (2048, 24, ("mirror", "ntt", "xfft"))])
def test_trgsw_cross_and_cmux_mean_what_the_reference_says(N, n, backends):
    import rustfhe_amd as R
    p = R.Params(N=N, n=n)
    key0, key1, bk, ksk = R.keygen(p, 7000 + N)
    ones, zeros = np.flatnonzero(key0 == 1), np.flatnonzero(key0 == 0)
    assert ones.size >= 3 and zeros.size >= 3
    rng = np.random.default_rng(N)
    tol = 2e-3 * N / 1024
    eng = R.Engine(p, 0)
    try:
        eng.load_bk_torus(bk)
        # trgsw_cross's message: 1/2 at even positions, 1/4 at odd ones
        m = np.where(np.arange(N) % 2 == 0, 0x80000000, 0x40000000).astype(np.uint32)
        # trgsw_cmux's operands: TRLWE(all One) and TRLWE(all Zero), One = +1/8, Zero = -1/8 (trlwe.rs:78-87)
        one_pol, zero_pol = np.full(N, 0x20000000, np.uint32), np.full(N, 0xE0000000, np.uint32)
        for name in backends:
            eng.set_backend({"mirror": R._ffi.BACKEND_FFT64_MIRROR, "ntt": R._ffi.BACKEND_NTT_EXACT, "xfft": R._ffi.BACKEND_FFT_SPLIT_EXACT}[name])
            idx = np.concatenate([ones[:3], zeros[:3]]).astype(np.int32)
            # --- cross: TRGSW(1) keeps the message, TRGSW(0) removes it
            cts = np.stack([trlwe_encrypt(rng, key1, m) for _ in idx])
            for ct in cts:
                assert torus_distance(trlwe_phase(key1, ct), m).max() < 1e-6       # the numpy encryption itself
            out = eng.external_product_batch(idx, cts)
            for k, i in enumerate(idx):
                want = m if key0[i] else np.zeros(N, np.uint32)
                dist = torus_distance(trlwe_phase(key1, out[k]), want)
                assert dist.max() < tol, (name, N, int(i), int(key0[i]), float(dist.max()))
            # --- cmux: TRGSW(i).cmux(rep_1, rep_0) = cross(rep_1 - rep_0) + rep_0 decrypts to pol_i
            rep_1 = np.stack([trlwe_encrypt(rng, key1, one_pol) for _ in idx])
            rep_0 = np.stack([trlwe_encrypt(rng, key1, zero_pol) for _ in idx])
            crossed = eng.external_product_batch(idx, rep_1 - rep_0)               # uint32 arithmetic wraps: the torus difference
            result = crossed + rep_0
            for k, i in enumerate(idx):
                ph = trlwe_phase(key1, result[k])
                bits = (ph < 0x80000000).astype(np.uint8)                          # torus_pol2binary_pol: f < 0.5 -> One (trlwe.rs:89-99)
                assert np.all(bits == key0[i]), (name, N, int(i), int(key0[i]), int((bits != key0[i]).sum()))
                want = one_pol if key0[i] else zero_pol
                assert torus_distance(ph, want).max() < tol
    finally:
        eng.close()
