"""GPU: the exact-integer NTT multiply backend (rtfhe_set_backend(RTFHE_BACKEND_NTT_EXACT)).  Its polynomial products are
exact, so it is bit-identical to the oracle's exact-integer backend (schoolbook negacyclic products, the semantics of the
reference's Polynomial::cross, utils/src/math.rs:238-257) -- and only decrypt-/phase-equivalent to the reference's FP64-FFT
path, which is what the default fft64-mirror backend reproduces bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ntt_engine(params, keys):
    import rustfhe_amd as R
    e = R.Engine(R.Params(), 0)
    e.load_bk_torus(keys.bk_t)
    e.load_ksk(keys.ksk)
    e.set_backend(R._ffi.BACKEND_NTT_EXACT)
    assert e.backend() == R._ffi.BACKEND_NTT_EXACT
    yield e
    e.close()


def test_ntt_external_product_is_exact(ntt_engine, orc, params, keys):
    rng = np.random.default_rng(71)
    idx = np.array([0, 5, 634, 300, 17, 99], np.int32)
    trlwe = rng.integers(0, 2 ** 32, (6, 2 * params.N), dtype=np.uint64).astype(np.uint32)
    trlwe[1] = 0
    trlwe[2] = 0xFFFFFFFF          # extreme digits / carries
    trlwe[3] = 0x7DF7C000          # every digit at the top of its range
    out = ntt_engine.external_product_batch(idx, trlwe)
    pl = orc.Plan(params.N, orc.BACKEND_EXACT)
    w = params.trgsw_words
    exp = np.stack([orc.external_product(params, pl, None, keys.bk_t[i * w:(i + 1) * w], t) for i, t in zip(idx, trlwe)])
    assert np.array_equal(out.reshape(exp.shape), exp)


@pytest.mark.parametrize("steps", [1, 4, 23])
def test_ntt_blind_rotate_prefix_is_exact(ntt_engine, orc, params, keys, gold_gate, steps):
    t = np.stack([orc.gate_linear(params, orc.NAND, a, b) for a, b in zip(gold_gate["in0"][:3], gold_gate["in1"][:3])])
    acc = ntt_engine.blind_rotate_batch(t, steps)
    pl = orc.Plan(params.N, orc.BACKEND_EXACT)
    exp = np.stack([orc.blind_rotate(params, pl, None, keys.bk_t, x, steps) for x in t])
    assert np.array_equal(acc.reshape(exp.shape), exp)


def test_ntt_whole_gate_exact_and_decrypts_like_the_reference_path(ntt_engine, engine, orc, params, keys, gold_gate):
    import rustfhe_amd as R
    pl = orc.Plan(params.N, orc.BACKEND_EXACT)
    a, b = gold_gate["in0"][3], gold_gate["in1"][3]
    out = ntt_engine.gate_batch(R.NAND, a[None], b[None])[0]
    exp = orc.gate(params, pl, orc.NAND, None, keys.bk_t, keys.ksk, a, b)          # ~10 s: 635 steps of schoolbook products
    assert np.array_equal(out, exp)
    # against the reference-exact (mirror) path: different ciphertext words, same plaintext, phases within the noise
    ref = gold_gate["out"][3]
    assert not np.array_equal(out, ref)
    assert keys.decrypt_bits([out]) == keys.decrypt_bits([ref])
    d = (int(keys.phase(out)) - int(keys.phase(ref)) + 2 ** 31) % 2 ** 32 - 2 ** 31
    assert abs(d) < 2 ** 26          # 1/64 of the torus; observed ~3e-3


def test_ntt_batch_decrypts_and_all_shapes_agree(ntt_engine, engine, params, keys):
    import rustfhe_amd as R
    rng = np.random.default_rng(72)
    G = 1536                                  # > 4 gates per CU: workgroups queue behind each other
    b0, b1 = rng.integers(0, 2, G), rng.integers(0, 2, G)
    c0, c1 = keys.encrypt_bits(b0), keys.encrypt_bits(b1)
    out = ntt_engine.gate_batch(R.NAND, c0, c1)
    assert keys.decrypt_bits(out) == list(1 - (b0 & b1))
    assert np.array_equal(ntt_engine.gate_batch(R.NAND, c0[:5], c1[:5]), out[:5])
    # launch shapes by batch size (1, 2, 3, 4 gates per workgroup for the remainder of a batch): the same words in every shape
    for k in (256, 257, 512, 513, 768, 769, 1024, 1025, 1300):
        assert np.array_equal(ntt_engine.gate_batch(R.NAND, c0[:k], c1[:k]), out[:k]), k
    for op, f in ((R.XOR, lambda x, y: x ^ y), (R.OR, lambda x, y: x | y)):
        o = ntt_engine.gate_batch(op, c0[:64], c1[:64])
        assert keys.decrypt_bits(o) == list(f(b0[:64], b1[:64]))


def test_backend_switch_back_to_mirror(ntt_engine, gold_gate):
    import rustfhe_amd as R
    ntt_engine.set_backend(R._ffi.BACKEND_FFT64_MIRROR)
    try:
        out = ntt_engine.gate_batch(R.NAND, gold_gate["in0"][:2], gold_gate["in1"][:2])
        assert np.array_equal(out, gold_gate["out"][:2])
    finally:
        ntt_engine.set_backend(R._ffi.BACKEND_NTT_EXACT)
    with pytest.raises(R.RtfheError):
        ntt_engine.set_backend(7)


def test_ntt_needs_torus_key(keys):
    import rustfhe_amd as R
    e = R.Engine(R.Params(), 0)
    try:
        e.load_bk_fft(keys.bk_f)             # FrrSeries-domain key only: no torus form to derive the NTT key from
        e.load_ksk(keys.ksk)
        e.set_backend(R._ffi.BACKEND_NTT_EXACT)
        with pytest.raises(R.RtfheError) as ei:
            e.gate_batch(R.NAND, np.zeros((1, 636), np.uint32), np.zeros((1, 636), np.uint32))
        assert ei.value.code == R._ffi.ERR_STATE
    finally:
        e.close()


def test_ntt_two_waves_per_gate_kernel_gives_the_same_words(ntt_engine, params, keys, gold_gate, monkeypatch):
    """The NTT backend's default kernel is k_bootstrap_ntt_pair (two waves per gate, symmetric split); RTFHE_FORCE_WAVES=4
    selects the one-wave-per-gate kernel: sums of the same exact integers => the same words, for gates, ragged counts and
    blind-rotate prefixes."""
    import rustfhe_amd as R
    monkeypatch.setenv("RTFHE_FORCE_WAVES", "4")
    e = R.Engine(R.Params(), 0)
    try:
        e.load_bk_torus(keys.bk_t)
        e.load_ksk(keys.ksk)
        e.set_backend(R._ffi.BACKEND_NTT_EXACT)
        in0, in1 = gold_gate["in0"][:7], gold_gate["in1"][:7]
        for cnt in (7, 1, 4):
            assert np.array_equal(e.gate_batch(R.NAND, in0[:cnt], in1[:cnt]), ntt_engine.gate_batch(R.NAND, in0[:cnt], in1[:cnt]))
        assert np.array_equal(e.gate_batch(R.XOR, in0[:3], in1[:3]), ntt_engine.gate_batch(R.XOR, in0[:3], in1[:3]))
        assert np.array_equal(e.blind_rotate_batch(in0[:5], 9), ntt_engine.blind_rotate_batch(in0[:5], 9))
    finally:
        e.close()


# ---- N = 2048 (BASELINE config 5) on the NTT backend: k_bootstrap_ntt_halves, two waves per transform ----
@pytest.fixture(scope="module")
def ntt2048(orc):
    import rustfhe_amd as R
    P = orc.Params(N=2048)
    K = orc.Keys(P, 2048)
    e = R.Engine(R.Params(N=2048), 0)
    e.load_bk_torus(K.bk_t)
    e.load_ksk(K.ksk)
    e.set_backend(R._ffi.BACKEND_NTT_EXACT)
    yield P, K, e
    e.close()


def test_ntt2048_external_product_is_exact(ntt2048, orc):
    P, K, e = ntt2048
    rng = np.random.default_rng(2049)
    idx = np.array([0, 5, 634, 300, 17], np.int32)
    trlwe = rng.integers(0, 2 ** 32, (5, 2 * P.N), dtype=np.uint64).astype(np.uint32)
    trlwe[1] = 0
    trlwe[2] = 0xFFFFFFFF
    trlwe[3] = 0x7DF7C000          # every digit at the top of its range: the row sums at their largest
    out = e.external_product_batch(idx, trlwe)
    pl = orc.Plan(P.N, orc.BACKEND_EXACT)
    w = P.trgsw_words
    exp = np.stack([orc.external_product(P, pl, None, K.bk_t[i * w:(i + 1) * w], t) for i, t in zip(idx, trlwe)])
    assert np.array_equal(out.reshape(exp.shape), exp)


def test_ntt2048_largest_sums_stay_exact(orc):
    """A key of extreme words against extreme digits: each of the two 3-row sums reaches 3 * 2048 * 32 * 2^31 = 2^48.58 < P/2; the sum
    of all six rows (2^49.58) would not fit -- the reason the kernel inverse-transforms the b-rows and the a-rows separately."""
    import rustfhe_amd as R
    P = orc.Params(n=2, N=2048)
    pl = orc.Plan(P.N, orc.BACKEND_EXACT)
    w = P.trgsw_words
    bk = np.empty(2 * w, np.uint32)
    bk[:w] = 0x80000000            # -2^31 everywhere
    bk[w:] = 0x7FFFFFFF
    e = R.Engine(R.Params(n=2, N=2048), 0)
    try:
        e.load_bk_torus(bk)
        e.set_backend(R._ffi.BACKEND_NTT_EXACT)
        trlwe = np.empty((4, 2 * P.N), np.uint32)
        trlwe[0] = 0x7DF7C000      # digits +31 (after the offset: top of the range) in every row
        trlwe[1] = 0x82082000      # digits -32
        trlwe[2] = np.where(np.arange(2 * P.N) % 2 == 0, 0x7DF7C000, 0x82082000).astype(np.uint32)
        trlwe[3] = 0x7DF7C000
        idx = np.array([0, 0, 1, 1], np.int32)
        out = e.external_product_batch(idx, trlwe)
        exp = np.stack([orc.external_product(P, pl, None, bk[i * w:(i + 1) * w], t) for i, t in zip(idx, trlwe)])
        assert np.array_equal(out.reshape(exp.shape), exp)
    finally:
        e.close()


@pytest.mark.parametrize("steps", [1, 2, 9])
def test_ntt2048_blind_rotate_prefix_is_exact(ntt2048, orc, steps):
    P, K, e = ntt2048
    c0, c1 = K.encrypt_bits([0, 1, 1]), K.encrypt_bits([1, 1, 0])
    t = np.stack([orc.gate_linear(P, orc.NAND, a, b) for a, b in zip(c0, c1)])
    acc = e.blind_rotate_batch(t, steps)
    pl = orc.Plan(P.N, orc.BACKEND_EXACT)
    exp = np.stack([orc.blind_rotate(P, pl, None, K.bk_t, x, steps) for x in t])
    assert np.array_equal(acc.reshape(exp.shape), exp)


def test_ntt2048_gates_exact_and_all_shapes_agree(ntt2048, orc):
    import os
    import rustfhe_amd as R
    P, K, e = ntt2048
    rng = np.random.default_rng(2050)
    G = 1100                                  # > 4 gates per CU
    b0, b1 = rng.integers(0, 2, G), rng.integers(0, 2, G)
    c0, c1 = K.encrypt_bits(b0), K.encrypt_bits(b1)
    out = e.gate_batch(R.NAND, c0, c1)
    assert K.decrypt_bits(out) == list(1 - (b0 & b1))
    for k in (1, 5, 256, 257, 513, 769, 1024, 1025):
        assert np.array_equal(e.gate_batch(R.NAND, c0[:k], c1[:k]), out[:k]), k
    # whole gates against the oracle's exact backend (schoolbook products: ~40 s of CPU per gate, one per thread)
    nt = min(8, os.cpu_count() or 1)
    pick = rng.choice(G, nt, replace=False)
    exp, _ = orc.gate_batch_mt(P, orc.NAND, None, K.bk_t, K.ksk, c0[pick], c1[pick], nthreads=nt, backend=orc.BACKEND_EXACT)
    assert np.array_equal(out[pick], exp)
    o = e.gate_batch(R.XOR, c0[:64], c1[:64])
    assert K.decrypt_bits(o) == list(b0[:64] ^ b1[:64])


def test_ntt_backend_runs_netlists(ntt_engine, orc, params, keys):
    """Circuit waves on the NTT backend: netlist-mode launches (wire table, per-gate opcodes) go through k_bootstrap_ntt_wg for
    small waves and the two-waves-per-gate kernel for large ones; one HIP-graph submission replays the same words; replica 0 is
    bit-exact, gate by gate, against the oracle's exact backend walking the same netlist."""
    from rustfhe_amd.circuit import CircuitRunner, ripple_carry_adder
    net = ripple_carry_adder(4, nand_only=True)
    pl = orc.Plan(params.N, orc.BACKEND_EXACT)
    for reps in (3, 200):                     # wave sizes 3..9 gates (latency kernel) and 200..600 (throughput kernels)
        rng = np.random.default_rng(900 + reps)
        bits = rng.integers(0, 2, (reps, 8))
        cts = keys.encrypt_bits(bits.reshape(-1)).reshape(reps, 8, params.n + 1)
        g, w = CircuitRunner(ntt_engine, net, reps), CircuitRunner(ntt_engine, net, reps)
        g.set_inputs(cts)
        w.set_inputs(cts)
        a = g.run(graph=True).outputs()
        b = w.run(graph=False).outputs()
        assert np.array_equal(a, b)
        dec = np.array(keys.decrypt_bits(a.reshape(-1, params.n + 1))).reshape(reps, 5)
        A = (bits[:, :4] * (1 << np.arange(4))).sum(axis=1)
        B = (bits[:, 4:] * (1 << np.arange(4))).sum(axis=1)
        assert np.array_equal((dec * (1 << np.arange(5))).sum(axis=1), A + B)
        if reps == 3:
            wires = [None, None] + list(cts[0])
            triv = np.zeros((2, params.n + 1), np.uint32)
            triv[0, -1], triv[1, -1] = 0xE0000000, 0x20000000
            wires[0], wires[1] = triv[0], triv[1]
            for op, x, y in net.gates[:6]:            # ~10 s of schoolbook products per gate on the CPU: the first six gates
                wires.append(orc.gate(params, pl, op, None, keys.bk_t, keys.ksk, wires[x], wires[y]))
            base = 2 + net.num_inputs
            for k in range(6):
                assert np.array_equal(w.wire(base + k)[0], wires[base + k]), k
        g.close()
        w.close()
