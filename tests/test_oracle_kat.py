"""CPU: the oracle's integer glue against every deterministic known-answer test the reference holds
for this path.  Citations are reference file:line."""
import numpy as np


def T(orc, f):
    return orc.lib().orc_torus_from_f32(f)


def test_rotate_kat(orc):
    # utils/src/math.rs:75-84 (doc-test) and :895-903 (polynomial_rotate)
    p = np.array([1, 2, 3, 4, 5], np.int32)
    r = lambda n: orc.rotate(p, n).tolist()
    assert r(1) == [-5, 1, 2, 3, 4]
    assert r(-1) == [2, 3, 4, 5, -1]
    assert r(5) == [-1, -2, -3, -4, -5]
    assert r(-4) == [5, -1, -2, -3, -4]
    assert r(-8) == r(2)
    assert r(10) == [1, 2, 3, 4, 5]
    assert r(3) == [-3, -4, -5, 1, 2]
    assert r(-3) == [4, 5, -1, -2, -3]


def test_torus_from_f32_kat(orc):
    # utils/src/math.rs:988-999 (decimal_from_f32)
    assert T(orc, 0.5) == 1 << 31
    assert T(orc, 0.25) == 1 << 30
    assert T(orc, 0.125) == 1 << 29
    assert T(orc, -0.5) == 1 << 31
    assert T(orc, -0.25) == (1 << 30) + (1 << 31)
    assert T(orc, 1.0 / 8.0) == 0x20000000 and T(orc, -1.0 / 8.0) == 0xE0000000
    assert T(orc, 1.0 / 64) == 1 << 26 and T(orc, 1.0 / 4096) == 1 << 20 and T(orc, 1.0 / 262144) == 1 << 14


def test_decomposition_inline_mask_kat(orc):
    # utils/src/math.rs:1207-1273 (decimal_decomposition) -- the unit-tested inline-mask variant
    L = orc.lib()
    d_i = lambda x, l, bits: orc.decomp_scalar(x, bits, L.orc_inline_decomp_mask(l, bits), l)
    assert orc.decomp_u32_scalar(0x80000000, 1, 32) == [1] + [0] * 31
    assert d_i(0x80000000, 32, 1) == [-1] + [0] * 31
    assert d_i(0x80000000, 8, 4) == [-8, 0, 0, 0, 0, 0, 0, 0]
    assert d_i(0x80000000, 7, 4) == [-8, 0, 0, 0, 0, 0, 0]
    assert orc.decomp_u32_scalar(0x80000001, 1, 31) == [1] + [0] * 29 + [1]
    assert d_i(0x80000001, 31, 1) == [0] + [-1] * 30
    assert d_i(0b000001_000010_000011_000000_000000_00, 3, 6) == [1, 2, 3]
    assert d_i(0b000001_000010_000011_100000_000000_00, 3, 6) == [1, 2, 4]
    assert d_i(0b011111_100000_100000_000000_100000_00, 3, 6) == [-32, -31, -32]
    # utils/src/math.rs:866-893 (polynomial_decomposition)
    assert d_i(0x00000001, 2, 16) == [0, 1] and d_i(0x00028000, 2, 16) == [3, -32768]


def test_decomposition_hot_path_mask(orc):
    # the variant the hot path really uses: make_decomp_mask(3, 6) = 0x02084000 (utils/src/math.rs:542-560,
    # call site hom_nand/src/trgsw.rs:269-271).  The reference has no KAT for it; these values are the
    # hand evaluation of math.rs:561-577 recorded in SURVEY 8c, pinned end-to-end by the golden gate vectors.
    L = orc.lib()
    M = L.orc_make_decomp_mask(3, 6)
    assert M == 0x02084000 and L.orc_inline_decomp_mask(3, 6) == 0x02082000
    assert L.orc_make_decomp_mask(2, 10) == 0x00201000
    d = lambda x: orc.decomp_scalar(x, 6, M, 3)
    assert d(0x0420c000) == [1, 2, 5] and d(0x0420e000) == [1, 2, 5]
    assert d(0x12345678) == [5, -29, 19]
    assert d(0x00004000) == [0, 0, 3] and d(0x00008000) == [0, 0, 2] and d(0x0000c000) == [0, 0, 5]
    assert d(0x7e080080) == [-32, -31, -32]
    assert d(0x20000000) == [8, 0, 0] and d(0xe0000000) == [-8, 0, 0]
    # reconstruction error stays within [-1, +2] units of 2^-18
    rng = np.random.default_rng(0)
    for x in rng.integers(0, 2 ** 32, 2000):
        dg = d(int(x))
        rec = sum(v << (32 - 6 * (i + 1)) for i, v in enumerate(dg))
        err = ((int(x) - rec + 2 ** 31) % 2 ** 32 - 2 ** 31) / 2 ** 14
        assert -2.0 <= err <= 1.0 + 1e-9, (hex(int(x)), dg, err)
        assert all(-32 <= v <= 31 for v in dg)


def test_negacyclic_product_kat(orc):
    # utils/src/math.rs:761-843 (polynomial_cross), :845-864 (polynomial_mul_add)
    u = lambda a: np.array(a, np.int64).astype(np.uint32)
    s = lambda r: r.astype(np.int32).tolist()
    assert s(orc.negacyclic_mul(u([2, 3, 4]), [4, 5, 6])) == [-30, -2, 43]
    assert (orc.negacyclic_mul(u([2, 3, 4]), [4, 5, 6]).astype(np.int32) + 1).tolist() == [-29, -1, 44]
    t = lambda f: T(orc, f)
    assert orc.negacyclic_mul([t(0.5), t(0.75)], [2, 3]).tolist() == [t(0.75), t(0.0)]
    assert orc.negacyclic_mul([t(0.5)], [1]).tolist() == [t(0.5)]
    assert orc.negacyclic_mul([t(0.25), t(0.5)], [1, 0]).tolist() == [t(0.25), t(0.5)]
    assert orc.negacyclic_mul([t(0.25)], [-1]).tolist() == [t(0.75)]
    assert orc.negacyclic_mul([t(0.5), t(0.25), t(0.125)], [1, -1, 1]).tolist() == [t(3 / 8), t(-3 / 8), t(3 / 8)]
    r = orc.negacyclic_mul([t(0.5), t(0.75)], [2, 3])
    assert ((r + np.array([t(0.125), t(0.25)], np.uint32)) & 0xFFFFFFFF).tolist() == [t(0.875), t(0.25)]


def test_tlwe_linear_ops_kat(orc):
    # hom_nand/src/tlwe.rs:302-326 (tlwerep_op), layout here: a[0..n), b
    t = lambda f: T(orc, f)
    l = np.array([t(0.5), t(0.25), t(0.5)], np.uint32)      # p_key = [0.5, 0.25], cipher = 0.5
    r = np.array([t(0.125), t(0.5), t(0.25)], np.uint32)
    assert (l + r).tolist() == [t(0.625), t(0.75), t(0.75)]
    assert (l - r).tolist() == [t(0.375), t(0.75), t(0.25)]
    assert (l * np.uint32(3)).tolist() == [t(0.5), t(0.75), t(0.5)]
    # gate pre-steps use exactly these ops (hom_nand/src/tfhe.rs:41-71)
    P = orc.Params(n=2)
    g = lambda op: orc.gate_linear(P, op, l, r).tolist()
    m = 2 ** 32
    assert g(orc.NAND) == [(-int(l[0] + r[0])) % m, (-int(l[1] + r[1])) % m, (t(0.125) - t(0.75)) % m]
    assert g(orc.AND) == [t(0.625), t(0.75), (t(0.75) - t(0.125)) % m]
    assert g(orc.OR) == [t(0.625), t(0.75), (t(0.75) + t(0.125)) % m]
    assert g(orc.XOR) == [(2 * t(0.625)) % m, (2 * t(0.75)) % m, (2 * t(0.75) + t(0.25)) % m]
    assert g(orc.NOT) == [(-int(l[0])) % m, (-int(l[1])) % m, (-int(l[2])) % m]


def test_fft_roundtrip_n16_kat(orc):
    # utils/src/spqlios.rs:243-276 (fft_test): fft_torus(ifft_torus(p)) == p exactly for p = X + X^2; p*p ~ X^2 + 2X^3 + X^4
    pl = orc.Plan(16)
    p = np.zeros(16, np.uint32)
    p[1] = p[2] = 1
    assert np.array_equal(pl.fft_u32(pl.ifft_i32(p.view(np.int32))), p)
    sq = pl.poly_mul(p, p).astype(np.int32)
    exp = np.zeros(16, np.int32)
    exp[2], exp[3], exp[4] = 1, 2, 1
    assert np.abs(sq - exp).max() < 1000


def test_fft_product_vs_exact_n1024(orc):
    # utils/src/math.rs:905-952 (polynomial_fft_cross): FFT product within 1e-6 of the torus of the exact one;
    # in fact never beyond +-1 LSB per external product
    rng = np.random.default_rng(3)
    pl = orc.Plan(1024)
    a = rng.integers(0, 2 ** 32, 1024, dtype=np.uint64).astype(np.uint32)
    b = rng.integers(-32, 32, 1024).astype(np.int32)
    fa, fb = pl.ifft_i32(a.view(np.int32)), pl.ifft_i32(b)
    had = np.empty(1024)
    import ctypes as C
    orc.lib().orc_hadamard(1024, had.ctypes.data_as(C.POINTER(C.c_double)), fa.ctypes.data_as(C.POINTER(C.c_double)),
                           fb.ctypes.data_as(C.POINTER(C.c_double)))
    got = pl.fft_u32(had).astype(np.int64)
    exact = orc.negacyclic_mul(a, b).astype(np.int64)
    diff = (got - exact + 2 ** 31) % 2 ** 32 - 2 ** 31
    assert np.abs(diff).max() <= 1


def test_sample_extract_and_key_switch_semantics(orc, params, keys):
    # hom_nand/src/trlwe.rs:178-230 and tlwe.rs:346-396: semantic round trips (phase preserved up to noise)
    import ctypes as C
    L = orc.lib()
    pl = orc.Plan(params.N)
    rng = orc.Rng()
    L.orc_rng_seed(C.byref(rng), 77)
    msg = np.full(params.N, 0x20000000, np.uint32)
    ct = np.empty(2 * params.N, np.uint32)
    L.orc_trlwe_encrypt(C.byref(rng), pl.h, params.N, keys.key1.ctypes.data_as(C.POINTER(C.c_int32)),
                        msg.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_float(2.0 ** -25), ct.ctypes.data_as(C.POINTER(C.c_uint32)))
    t1 = orc.sample_extract(params, ct, 0)
    ph1 = L.orc_tlwe_phase(params.N, keys.key1.ctypes.data_as(C.POINTER(C.c_int32)), t1.ctypes.data_as(C.POINTER(C.c_uint32)))
    assert abs(((ph1 - 0x20000000 + 2 ** 31) % 2 ** 32) - 2 ** 31) < 2 ** 14
    t0 = orc.key_switch(params, keys.ksk, t1)
    ph0 = keys.phase(t0)
    assert abs(((ph0 - 0x20000000 + 2 ** 31) % 2 ** 32) - 2 ** 31) < 2 ** 24
    assert orc.lib().orc_torus2binary(ph0) == 1


def test_key_switch_in_the_reference_container_shape(orc, params, keys, gold_gate):
    """KeySwitchingKey(Vec<[[TLWERep; IKS_T]; IKS_L]>), IKS_T = 2^IKS_BASEBIT = 4 (hom_nand/src/tlwe.rs:178-180, 243-245):
    get(i, l, t) = [i][l][t-1] for t = digit in 1 .. 3 -- the 4th entry of a level never enters a key switch."""
    k4 = keys.ksk_ref()
    base = 1 << params.ks_basebit
    assert k4.size == keys.ksk.size // (base - 1) * base
    rng = np.random.default_rng(21)
    t1 = rng.integers(0, 2 ** 32, (4, params.N + 1), dtype=np.uint64).astype(np.uint32)
    t1[0] = gold_gate["extract"]
    t1[1, :params.N] = 0xFFFFFFFF
    for x in t1:
        assert np.array_equal(orc.key_switch_ref(params, k4, x), orc.key_switch(params, keys.ksk, x))
    assert np.array_equal(orc.key_switch_ref(params, k4, t1[0]), gold_gate["out"][0])
    # scribbling over every 4th entry changes nothing
    k4b = k4.copy().reshape(params.N, params.ks_t, base, params.n + 1)
    k4b[:, :, base - 1] = 0xDEADBEEF
    assert np.array_equal(orc.key_switch_ref(params, k4b.reshape(-1), t1[2]), orc.key_switch(params, keys.ksk, t1[2]))


def test_exact_int_backend_decrypts_like_mirror(orc, keys):
    # an exact-integer multiply gives different ciphertext bits but the same plaintext; small n keeps it fast
    small = orc.Params(n=6)
    kk = orc.Keys(small, 99)
    c = kk.encrypt_bits([1, 0])
    mir = orc.gate(small, orc.Plan(small.N), orc.NAND, kk.bk_f, None, kk.ksk, c[0], c[1])
    exa = orc.gate(small, orc.Plan(small.N, orc.BACKEND_EXACT), orc.NAND, None, kk.bk_t, kk.ksk, c[0], c[1])
    assert kk.decrypt_bits([mir]) == kk.decrypt_bits([exa]) == [1]


def test_ntt_backend_model():
    """scripts/ntt/model.py: the NTT backend's operation sequence on exact integers (tables incl. the two digit-table stages, product == exact)
    and the worst-case interval bounds of the lean renormalisation schedule (every intermediate < 2^53 for ANY input)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "ntt", "model.py")], capture_output=True, text=True, timeout=300)
    lines = out.stdout.strip().split("\n")
    assert out.returncode == 0 and lines[0].startswith("worst-case bounds") and lines[-1].startswith("ok"), out.stdout + out.stderr

