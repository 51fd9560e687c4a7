"""GPU: the split-FFT exact backend (rtfhe_set_backend(RTFHE_BACKEND_FFT_SPLIT_EXACT)): exact negacyclic products through an FMA-contracted
FP64 FFT with the key split into signed 16-bit halves (rtfhe_xfft.hpp; model and error bound: scripts/xfft/model.py).  Its products are
exact integers, so it must be bit-identical to the oracle's exact-integer backend (schoolbook products, the semantics of the reference's
Polynomial::cross, utils/src/math.rs:238-257, KAT :761-843) and to the NTT backend -- for EVERY input, including the largest magnitudes
the path can produce -- and only decrypt-/phase-equivalent to the reference's FP64-FFT path."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def xe(params, keys):
    import rustfhe_amd as R
    e = R.Engine(R.Params(), 0)
    e.load_bk_torus(keys.bk_t)
    e.load_ksk(keys.ksk)
    e.set_backend(R._ffi.BACKEND_FFT_SPLIT_EXACT)
    assert e.backend() == R._ffi.BACKEND_FFT_SPLIT_EXACT
    yield e
    e.close()


def test_xfft_external_product_is_exact(xe, orc, params, keys):
    rng = np.random.default_rng(171)
    idx = np.array([0, 5, 634, 300, 17, 99, 1], np.int32)
    trlwe = rng.integers(0, 2 ** 32, (7, 2 * params.N), dtype=np.uint64).astype(np.uint32)
    trlwe[1] = 0
    trlwe[2] = 0xFFFFFFFF          # extreme digits / carries
    trlwe[3] = 0x7DF7C000          # every digit at the top of its range
    trlwe[4] = 0x82082000          # ... at the bottom
    out = xe.external_product_batch(idx, trlwe)
    pl = orc.Plan(params.N, orc.BACKEND_EXACT)
    w = params.trgsw_words
    exp = np.stack([orc.external_product(params, pl, None, keys.bk_t[i * w:(i + 1) * w], t) for i, t in zip(idx, trlwe)])
    assert np.array_equal(out.reshape(exp.shape), exp)


def test_xfft_largest_sums_stay_exact(orc):
    """Keys of extreme words against extreme digits: every half-product sum at the largest magnitude the path can produce
    (6 * 1024 * 32 * 2^15 = 2^33.6), with all terms of an output coefficient aligned in sign; key words chosen so that both 16-bit halves sit
    at the ends of their ranges (0x80008000: hi = -2^15 + 1... the split is lo = sign-extended low half, hi = (k - lo) >> 16, so 0x7FFF8000 has
    hi = +2^15, the one value outside int16).  The proven bound on the distance from an integer is 2^-8.3 (scripts/xfft/model.py)."""
    import rustfhe_amd as R
    P = orc.Params(n=6, N=1024)
    pl = orc.Plan(P.N, orc.BACKEND_EXACT)
    w = P.trgsw_words
    words = [0x80000000, 0x7FFFFFFF, 0x7FFF8000, 0x80008000, 0x8000FFFF, 0x00008000]
    bk = np.empty(len(words) * w, np.uint32)
    for i, v in enumerate(words):
        bk[i * w:(i + 1) * w] = v
    bk[5 * w:6 * w:2] = 0x7FFF7FFF          # alternating signs in the last key
    e = R.Engine(R.Params(n=6, N=1024), 0)
    try:
        e.load_bk_torus(bk)
        e.set_backend(R._ffi.BACKEND_FFT_SPLIT_EXACT)
        pats = np.empty((4, 2 * P.N), np.uint32)
        pats[0] = 0x7DF7C000      # digits +31 in every row
        pats[1] = 0x82082000      # digits -32
        pats[2] = np.where(np.arange(2 * P.N) % 2 == 0, 0x7DF7C000, 0x82082000).astype(np.uint32)
        pats[3] = np.where(np.arange(2 * P.N) % P.N < P.N // 2, 0x7DF7C000, 0x82082000).astype(np.uint32)
        trlwe = np.concatenate([pats] * len(words))
        idx = np.repeat(np.arange(len(words), dtype=np.int32), len(pats))
        out = e.external_product_batch(idx, trlwe)
        exp = np.stack([orc.external_product(P, pl, None, bk[i * w:(i + 1) * w], t) for i, t in zip(idx, trlwe)])
        assert np.array_equal(out.reshape(exp.shape), exp)
        # the same extremes through the bootstrap kernel's own CMUX steps (rotation, decomposition, both barriers, the key ring)
        t = np.zeros((3, P.n + 1), np.uint32)
        t[1] = 0x12345678
        t[2] = np.arange(P.n + 1, dtype=np.uint32) * 0x1F2E3D4C
        for steps in (1, 6):
            acc = e.blind_rotate_batch(t, steps)
            ex = np.stack([orc.blind_rotate(P, pl, None, bk, x, steps) for x in t])
            assert np.array_equal(acc.reshape(ex.shape), ex), steps
    finally:
        e.close()


@pytest.mark.parametrize("steps", [1, 4, 23])
def test_xfft_blind_rotate_prefix_is_exact(xe, orc, params, keys, gold_gate, steps):
    t = np.stack([orc.gate_linear(params, orc.NAND, a, b) for a, b in zip(gold_gate["in0"][:3], gold_gate["in1"][:3])])
    acc = xe.blind_rotate_batch(t, steps)
    pl = orc.Plan(params.N, orc.BACKEND_EXACT)
    exp = np.stack([orc.blind_rotate(params, pl, None, keys.bk_t, x, steps) for x in t])
    assert np.array_equal(acc.reshape(exp.shape), exp)


def test_xfft_whole_gate_exact_and_decrypts_like_the_reference_path(xe, orc, params, keys, gold_gate):
    import rustfhe_amd as R
    pl = orc.Plan(params.N, orc.BACKEND_EXACT)
    a, b = gold_gate["in0"][3], gold_gate["in1"][3]
    out = xe.gate_batch(R.NAND, a[None], b[None])[0]
    exp = orc.gate(params, pl, orc.NAND, None, keys.bk_t, keys.ksk, a, b)          # ~10 s: 635 steps of schoolbook products
    assert np.array_equal(out, exp)
    ref = gold_gate["out"][3]
    assert not np.array_equal(out, ref)
    assert keys.decrypt_bits([out]) == keys.decrypt_bits([ref])
    d = (int(keys.phase(out)) - int(keys.phase(ref)) + 2 ** 31) % 2 ** 32 - 2 ** 31
    assert abs(d) < 2 ** 26


def test_xfft_equals_the_ntt_backend_on_a_large_batch_in_every_launch_shape(xe, params, keys):
    """Two independent exact backends (modular NTT in doubles / split FP64 FFT) must give the same words on every gate: 1,536 gates
    (> 4 gates per CU: workgroups queue), then every launch shape by batch size, fused and split key switch."""
    import rustfhe_amd as R
    rng = np.random.default_rng(172)
    G = 1536
    b0, b1 = rng.integers(0, 2, G), rng.integers(0, 2, G)
    c0, c1 = keys.encrypt_bits(b0), keys.encrypt_bits(b1)
    out = xe.gate_batch(R.NAND, c0, c1)
    assert keys.decrypt_bits(out) == list(1 - (b0 & b1))
    ntt = R.Engine(R.Params(), 0)
    try:
        ntt.load_bk_torus(keys.bk_t)
        ntt.load_ksk(keys.ksk)
        ntt.set_backend(R._ffi.BACKEND_NTT_EXACT)
        assert np.array_equal(out, ntt.gate_batch(R.NAND, c0, c1))
        for op in (R.XOR, R.OR, R.AND, R.NOT):
            assert np.array_equal(xe.gate_batch(op, c0[:70], c1[:70]), ntt.gate_batch(op, c0[:70], c1[:70])), op
        assert np.array_equal(xe.blind_rotate_batch(c0[:300], 7), ntt.blind_rotate_batch(c0[:300], 7))
        assert np.array_equal(xe.mux_batch(c0[:9], c1[:9], c0[9:18]), ntt.mux_batch(c0[:9], c1[:9], c0[9:18]))      # three bootstraps, ANDNY among them
        assert np.array_equal(xe.bootstrap_batch(c0[:5]), ntt.bootstrap_batch(c0[:5]))
    finally:
        ntt.close()
    for k in (1, 5, 256, 257, 512, 513, 768, 769, 1024, 1025, 1300):
        assert np.array_equal(xe.gate_batch(R.NAND, c0[:k], c1[:k]), out[:k]), k


def test_xfft_fused_key_switch_gives_the_same_words(xe, keys, monkeypatch):
    import rustfhe_amd as R
    monkeypatch.setenv("RTFHE_KS_MM_MIN", "0")
    e = R.Engine(R.Params(), 0)
    try:
        e.load_bk_torus(keys.bk_t)
        e.load_ksk(keys.ksk)
        e.set_backend(R._ffi.BACKEND_FFT_SPLIT_EXACT)
        rng = np.random.default_rng(173)
        b0, b1 = rng.integers(0, 2, 9), rng.integers(0, 2, 9)
        c0, c1 = keys.encrypt_bits(b0), keys.encrypt_bits(b1)
        assert np.array_equal(e.gate_batch(R.NAND, c0, c1), xe.gate_batch(R.NAND, c0, c1))
    finally:
        e.close()


def test_xfft_needs_torus_key(keys):
    import rustfhe_amd as R
    e = R.Engine(R.Params(), 0)
    try:
        e.load_bk_fft(keys.bk_f)
        e.load_ksk(keys.ksk)
        e.set_backend(R._ffi.BACKEND_FFT_SPLIT_EXACT)
        with pytest.raises(R.RtfheError) as ei:
            e.gate_batch(R.NAND, np.zeros((1, 636), np.uint32), np.zeros((1, 636), np.uint32))
        assert ei.value.code == R._ffi.ERR_STATE
    finally:
        e.close()


# ---- N = 2048 (BASELINE config 5) on the split-FFT backend: k_bootstrap_xquad, four waves per gate (rtfhe_kernels_xfft2.hpp) ----
@pytest.fixture(scope="module")
def x2048(orc):
    import rustfhe_amd as R
    P = orc.Params(N=2048)
    K = orc.Keys(P, 2048)
    e = R.Engine(R.Params(N=2048), 0)
    e.load_bk_torus(K.bk_t)
    e.load_ksk(K.ksk)
    e.set_backend(R._ffi.BACKEND_FFT_SPLIT_EXACT)
    yield P, K, e
    e.close()


@pytest.mark.parametrize("steps", [0, 1, 2, 9])
def test_xfft2048_blind_rotate_prefix_is_exact(x2048, orc, steps):
    """One, two, nine CMUX steps of the four-wave kernel (one gate per workgroup: the latency shape of the dispatch) against the oracle's
    schoolbook products: stage 1 across the halves, both hand-offs, the sibling trade of the last inverse stage, the update."""
    P, K, e = x2048
    c0, c1 = K.encrypt_bits([0, 1, 1]), K.encrypt_bits([1, 1, 0])
    t = np.stack([orc.gate_linear(P, orc.NAND, a, b) for a, b in zip(c0, c1)])
    acc = e.blind_rotate_batch(t, steps)
    pl = orc.Plan(P.N, orc.BACKEND_EXACT)
    exp = np.stack([orc.blind_rotate(P, pl, None, K.bk_t, x, steps) for x in t])
    assert np.array_equal(acc.reshape(exp.shape), exp)


def test_xfft2048_largest_sums_stay_exact(orc):
    """Keys of extreme words against extreme digits at N = 2048: every half-product sum at 6 * 2048 * 32 * 2^15 = 2^34.6, all terms of an
    output coefficient aligned; proven bound on the distance from an integer 2^-6.6 (scripts/xfft/model.py).  Through the bootstrap kernel's
    own CMUX steps (the stage-level external product of this backend is N = 1024 only), in both launch shapes."""
    import rustfhe_amd as R
    P = orc.Params(n=6, N=2048)
    pl = orc.Plan(P.N, orc.BACKEND_EXACT)
    w = P.trgsw_words
    words = [0x80000000, 0x7FFFFFFF, 0x7FFF8000, 0x80008000, 0x8000FFFF, 0x00008000]
    bk = np.empty(len(words) * w, np.uint32)
    for i, v in enumerate(words):
        bk[i * w:(i + 1) * w] = v
    bk[5 * w:6 * w:2] = 0x7FFF7FFF          # alternating signs in the last key
    e = R.Engine(R.Params(n=6, N=2048), 0)
    try:
        e.load_bk_torus(bk)
        e.set_backend(R._ffi.BACKEND_FFT_SPLIT_EXACT)
        t = np.zeros((3, P.n + 1), np.uint32)
        t[1] = 0x12345678
        t[2] = np.arange(P.n + 1, dtype=np.uint32) * 0x1F2E3D4C
        for steps in (1, 6):
            ex = np.stack([orc.blind_rotate(P, pl, None, bk, x, steps) for x in t])
            acc = e.blind_rotate_batch(t, steps)
            assert np.array_equal(acc.reshape(ex.shape), ex), steps
            many = np.tile(t, (100, 1))                       # 300 gates: two gates per workgroup
            accm = e.blind_rotate_batch(many, steps).reshape(100, 3, -1)
            assert np.array_equal(accm, np.broadcast_to(ex, accm.shape)), steps
    finally:
        e.close()


def test_xfft2048_equals_the_ntt_backend_in_every_launch_shape(x2048):
    """Two independent exact backends must give the same words on every gate at N = 2048 too: 1,100 gates (> 2 gates per CU: workgroups
    queue), then both launch shapes by batch size, the fused and the batch key switch, other gates, blind-rotate prefixes."""
    import rustfhe_amd as R
    P, K, e = x2048
    rng = np.random.default_rng(2051)
    G = 1100
    b0, b1 = rng.integers(0, 2, G), rng.integers(0, 2, G)
    c0, c1 = K.encrypt_bits(b0), K.encrypt_bits(b1)
    out = e.gate_batch(R.NAND, c0, c1)
    assert K.decrypt_bits(out) == list(1 - (b0 & b1))
    ntt = R.Engine(R.Params(N=2048), 0)
    try:
        ntt.load_bk_torus(K.bk_t)
        ntt.load_ksk(K.ksk)
        ntt.set_backend(R._ffi.BACKEND_NTT_EXACT)
        assert np.array_equal(out, ntt.gate_batch(R.NAND, c0, c1))
        for op in (R.XOR, R.OR, R.NOT):
            assert np.array_equal(e.gate_batch(op, c0[:70], c1[:70]), ntt.gate_batch(op, c0[:70], c1[:70])), op
        assert np.array_equal(e.blind_rotate_batch(c0[:300], 5), ntt.blind_rotate_batch(c0[:300], 5))
        assert np.array_equal(e.mux_batch(c0[:9], c1[:9], c0[9:18]), ntt.mux_batch(c0[:9], c1[:9], c0[9:18]))
        assert K.decrypt_bits(e.mux_batch(c0[:9], c1[:9], c0[9:18])) == list(np.where(b0[:9], b0[9:18], b1[:9]))
    finally:
        ntt.close()
    for k in (1, 5, 256, 257, 512, 513, 700, 1024):
        assert np.array_equal(e.gate_batch(R.NAND, c0[:k], c1[:k]), out[:k]), k


def test_xfft2048_fused_key_switch_gives_the_same_words(x2048, monkeypatch):
    import rustfhe_amd as R
    P, K, xe = x2048
    monkeypatch.setenv("RTFHE_KS_MM_MIN", "0")
    e = R.Engine(R.Params(N=2048), 0)
    try:
        e.load_bk_torus(K.bk_t)
        e.load_ksk(K.ksk)
        e.set_backend(R._ffi.BACKEND_FFT_SPLIT_EXACT)
        rng = np.random.default_rng(2052)
        b0, b1 = rng.integers(0, 2, 9), rng.integers(0, 2, 9)
        c0, c1 = K.encrypt_bits(b0), K.encrypt_bits(b1)
        assert np.array_equal(e.gate_batch(R.NAND, c0, c1), xe.gate_batch(R.NAND, c0, c1))
        big = np.tile(c0, (40, 1)), np.tile(c1, (40, 1))      # 360 gates: two per workgroup, fused key switch with four waves summing quarters
        assert np.array_equal(e.gate_batch(R.NAND, *big), np.tile(xe.gate_batch(R.NAND, c0, c1), (40, 1)))
    finally:
        e.close()


def test_xfft2048_external_product_is_exact(x2048, orc):
    """The stage-level external product of this backend at N = 2048 (two waves per sample, the bootstrap kernel's own device functions): random
    and extreme TRLWE samples against the oracle's schoolbook products."""
    P, K, e = x2048
    rng = np.random.default_rng(2053)
    idx = np.array([0, 5, 634, 300, 17, 99, 1], np.int32)
    trlwe = rng.integers(0, 2 ** 32, (7, 2 * P.N), dtype=np.uint64).astype(np.uint32)
    trlwe[1] = 0
    trlwe[2] = 0xFFFFFFFF          # extreme digits / carries
    trlwe[3] = 0x7DF7C000          # every digit at the top of its range
    trlwe[4] = 0x82082000          # ... at the bottom
    out = e.external_product_batch(idx, trlwe)
    pl = orc.Plan(P.N, orc.BACKEND_EXACT)
    w = P.trgsw_words
    exp = np.stack([orc.external_product(P, pl, None, K.bk_t[i * w:(i + 1) * w], t) for i, t in zip(idx, trlwe)])
    assert np.array_equal(out.reshape(exp.shape), exp)


def test_xfft2048_external_product_largest_sums(orc):
    """... and on keys of extreme words against extreme digits: the sums at 2^34.6, every term of an output coefficient aligned."""
    import rustfhe_amd as R
    P = orc.Params(n=6, N=2048)
    pl = orc.Plan(P.N, orc.BACKEND_EXACT)
    w = P.trgsw_words
    words = [0x80000000, 0x7FFFFFFF, 0x7FFF8000, 0x80008000, 0x8000FFFF, 0x00008000]
    bk = np.empty(len(words) * w, np.uint32)
    for i, v in enumerate(words):
        bk[i * w:(i + 1) * w] = v
    bk[5 * w:6 * w:2] = 0x7FFF7FFF
    e = R.Engine(R.Params(n=6, N=2048), 0)
    try:
        e.load_bk_torus(bk)
        e.set_backend(R._ffi.BACKEND_FFT_SPLIT_EXACT)
        pats = np.empty((4, 2 * P.N), np.uint32)
        pats[0] = 0x7DF7C000
        pats[1] = 0x82082000
        pats[2] = np.where(np.arange(2 * P.N) % 2 == 0, 0x7DF7C000, 0x82082000).astype(np.uint32)
        pats[3] = np.where(np.arange(2 * P.N) % P.N < P.N // 2, 0x7DF7C000, 0x82082000).astype(np.uint32)
        trlwe = np.concatenate([pats] * len(words))
        idx = np.repeat(np.arange(len(words), dtype=np.int32), len(pats))
        out = e.external_product_batch(idx, trlwe)
        exp = np.stack([orc.external_product(P, pl, None, bk[i * w:(i + 1) * w], t) for i, t in zip(idx, trlwe)])
        assert np.array_equal(out.reshape(exp.shape), exp)
    finally:
        e.close()


def test_xfft2048_runs_netlists_and_shards():
    """N = 2048 on the split-FFT backend beyond plain batches: netlist waves through k_bootstrap_xquad's netlist mode (wire table, per-gate
    opcodes; one HIP-graph submission replays the same words as wave-by-wave launches, and the adder adds) in both launch shapes, a batch
    resident on the device sharded over a two-entry context, MUX on two streams.  Small TLWE dimension: key generation stays short."""
    import torch
    import rustfhe_amd as R
    from rustfhe_amd.circuit import CircuitRunner, ripple_carry_adder
    p = R.Params(N=2048, n=40)
    key0, key1, bk, ksk = R.keygen(p, 20480)
    e = R.Engine(p, 0)
    m = R.Engine(p, devices=[0, 0])
    try:
        for eng in (e, m):
            eng.load_bk_torus(bk); eng.load_ksk(ksk)
            eng.set_backend(R._ffi.BACKEND_FFT_SPLIT_EXACT)
        net = ripple_carry_adder(4, nand_only=True)
        for reps in (3, 150):                      # waves of 3..9 gates (one gate per workgroup) and 150..450 (two per workgroup)
            rng = np.random.default_rng(2100 + reps)
            bits = rng.integers(0, 2, (reps, 8)).astype(np.uint8)
            cts = R.encrypt_bits(p, key0, bits.reshape(-1), 7).reshape(reps, 8, p.n + 1)
            g, w = CircuitRunner(e, net, reps), CircuitRunner(e, net, reps)
            g.set_inputs(cts); w.set_inputs(cts)
            a = g.run(graph=True).outputs()
            b = w.run(graph=False).outputs()
            assert np.array_equal(a, b)
            dec = np.array(R.decrypt_bits(p, key0, a.reshape(-1, p.n + 1))).reshape(reps, 5)
            A = (bits[:, :4] * (1 << np.arange(4))).sum(axis=1)
            B = (bits[:, 4:] * (1 << np.arange(4))).sum(axis=1)
            assert np.array_equal((dec * (1 << np.arange(5))).sum(axis=1), A + B)
            g.close(); w.close()
        # sharded device-resident batch == single-device batch, word for word; ragged count
        k = 777
        rng = np.random.default_rng(2200)
        b0, b1 = rng.integers(0, 2, k).astype(np.uint8), rng.integers(0, 2, k).astype(np.uint8)
        d0 = torch.from_numpy(R.encrypt_bits(p, key0, b0, 1).view(np.int32)).cuda()
        d1 = torch.from_numpy(R.encrypt_bits(p, key0, b1, 2).view(np.int32)).cuda()
        o1, o2 = torch.empty_like(d0), torch.empty_like(d0)
        st = torch.cuda.current_stream().cuda_stream
        e.gate_batch_dev(R.XOR, d0, d1, o1, k, st); m.gate_batch_dev(R.XOR, d0, d1, o2, k, st); e.sync(st); m.sync(st)
        assert torch.equal(o1, o2)
        assert list(R.decrypt_bits(p, key0, o1.cpu().numpy().view(np.uint32))) == list(b0 ^ b1)
        # MUX batches overlapping on two streams
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        torch.cuda.synchronize()
        ref = torch.empty_like(d0[:200]); e.mux_batch_dev(d0, d1, o1, ref, 200, st); e.sync(st)
        x1, x2 = torch.zeros_like(ref), torch.zeros_like(ref)
        for _ in range(2):
            e.mux_batch_dev(d0, d1, o1, x1, 200, s1.cuda_stream)
            e.mux_batch_dev(d0, d1, o1, x2, 200, s2.cuda_stream)
        torch.cuda.synchronize()
        assert torch.equal(x1, ref) and torch.equal(x2, ref)
    finally:
        e.close(); m.close()
