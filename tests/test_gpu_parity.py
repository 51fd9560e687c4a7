"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle and the committed
golden vectors (generated with the reference's own compiled FFT).  Bit-exact everywhere: the work is
integer/torus arithmetic plus an FP64 transform mirrored operation for operation."""
import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu


def test_native_library_is_loaded(engine):
    import rustfhe_amd as R
    assert R.load().rtfhe_device_count() >= 1
    with open("/proc/self/maps") as f:
        assert "librtfhe_hip.so" in f.read()


def test_twiddles_match_reference_tables(engine, orc, params):
    """The tables this host's libm builds (reference_twiddles=False: no shipped table is consulted) are the reference build's, byte for
    byte; so the session engine found nothing to replace when it checked the shipped tables."""
    import rustfhe_amd as R
    g = golden("fft_N1024.npz")
    raw = R.Engine(engine.p, 0, reference_twiddles=False)
    try:
        a, b = raw.twiddles()
    finally:
        raw.close()
    assert a.tobytes() == g["ifft_table"].tobytes()
    assert b.tobytes() == g["fft_table"].tobytes()
    oa, ob = orc.Plan(params.N).tables()
    assert a.tobytes() == oa.tobytes() and b.tobytes() == ob.tobytes()
    assert engine.twiddle_entries_replaced == 0
    assert engine.twiddles()[0].tobytes() == a.tobytes()


def test_a_host_libm_that_disagrees_is_overridden_by_the_shipped_reference_tables(keys, gold_gate, tmp_path):
    """the twiddle tables are libm outputs, two libms may differ by an ulp in a few entries, and one differing entry changes
    torus words.  A host whose libm disagrees is stood in for through the public boundary: ONE entry of the tables a context has built is
    moved by one ulp (rtfhe_get_twiddles / rtfhe_set_twiddles -- no hook inside the library).  Left alone, that context no longer reproduces
    the reference build's tables (nor, in general, its spectra); rtfhe_twiddles_load -- what the default engine runs at construction --
    notices the disagreement with rustfhe_amd/assets/twiddles_N1024.bin, installs the shipped tables and the golden gates come out word for word."""
    import rustfhe_amd as R
    g = golden("fft_N1024.npz")
    p = R.Params()

    def perturbed():
        e = R.Engine(p, 0, reference_twiddles=False)
        a, b = e.twiddles()
        a[8 * 9 + 1] = np.nextafter(a[8 * 9 + 1], 2.0)            # a cosine of the twist block: every forward transform reads it
        e.set_twiddles(a, b)
        return e
    raw, fixed = perturbed(), perturbed()
    fixed.twiddle_entries_replaced = fixed.twiddles_load(R.engine.reference_twiddle_file(1024))
    try:
        ra, rb = raw.twiddles()
        assert int((ra.view(np.uint64) != g["ifft_table"].view(np.uint64)).sum()) == 1 and rb.tobytes() == g["fft_table"].tobytes()
        raw.load_bk_torus(keys.bk_t)
        assert raw.export_bk_fft().tobytes() != keys.bk_f.tobytes()            # the perturbed entry reaches the key spectra
        assert fixed.twiddle_entries_replaced == 1
        fa, fb = fixed.twiddles()
        assert fa.tobytes() == g["ifft_table"].tobytes() and fb.tobytes() == g["fft_table"].tobytes()
        fixed.load_bk_torus(keys.bk_t)
        fixed.load_ksk(keys.ksk)
        assert fixed.export_bk_fft().tobytes() == keys.bk_f.tobytes()
        ops, in0, in1 = gold_gate["ops"], gold_gate["in0"], gold_gate["in1"]
        for k in range(len(ops)):
            assert np.array_equal(fixed.gate_batch(int(ops[k]), in0[k:k + 1], in1[k:k + 1])[0], gold_gate["out"][k]), "gate %d" % k
        # the file round trip: what the fixed context writes is the shipped file, and loading it into the raw context repairs that one too
        path = str(tmp_path / "tw.bin")
        fixed.twiddles_write(path)
        assert open(path, "rb").read() == open(R.engine.reference_twiddle_file(1024), "rb").read()
        assert raw.twiddles_load(path) == 1 and raw.twiddles_load(path) == 0
        assert raw.export_bk_fft().tobytes() == keys.bk_f.tobytes()            # the torus-form key was re-transformed
        with open(path, "r+b") as f:                                            # a corrupted file is refused
            f.seek(100); f.write(b"\x01")
        with pytest.raises(R.RtfheError):
            raw.twiddles_load(path)
    finally:
        raw.close(); fixed.close()


def test_forward_transform_golden_and_oracle(engine, orc, params):
    g = golden("fft_N1024.npz")
    res = engine.ifft_i32_batch(g["fft_src"])
    assert res.tobytes() == g["fft_fwd"].tobytes()
    rng = np.random.default_rng(11)
    src = np.concatenate([rng.integers(-32, 32, (40, params.N)), rng.integers(-2 ** 31, 2 ** 31, (40, params.N)),
                          rng.integers(0, 2, (19, params.N)), np.zeros((1, params.N))]).astype(np.int32)
    res = engine.ifft_i32_batch(src)
    pl = orc.Plan(params.N)
    exp = np.stack([pl.ifft_i32(s) for s in src])
    assert res.tobytes() == exp.tobytes()


def test_inverse_transform_golden_and_oracle(engine, orc, params):
    g = golden("fft_N1024.npz")
    res = engine.fft_u32_batch(g["inv_src"])
    assert np.array_equal(res, g["inv_out"])
    rng = np.random.default_rng(12)
    pl = orc.Plan(params.N)
    spec = np.stack([pl.ifft_i32(rng.integers(-2 ** 31, 2 ** 31, params.N).astype(np.int32)) * float(rng.integers(1, 2 ** 16))
                     for _ in range(64)])
    spec[3] = -spec[3]
    spec[5] = 0.0
    res = engine.fft_u32_batch(spec)
    exp = np.stack([pl.fft_u32(s) for s in spec])
    assert np.array_equal(res, exp)


def test_transform_roundtrip_property(engine, params):
    """fft_torus(ifft_torus(p)) == p exactly (the reference's fft_test, utils/src/spqlios.rs:243-276, at N=1024),
    for small-magnitude polynomials where the FP64 error stays below 1/2."""
    rng = np.random.default_rng(13)
    src = rng.integers(-2 ** 20, 2 ** 20, (32, params.N)).astype(np.int32)
    back = engine.fft_u32_batch(engine.ifft_i32_batch(src))
    # truncation toward zero of x +/- eps can be off by one; the reference asserts equality only for tiny inputs
    diff = (back.astype(np.int64) - src.astype(np.uint32).astype(np.int64) + 2 ** 31) % 2 ** 32 - 2 ** 31
    assert np.abs(diff).max() <= 1
    tiny = np.zeros((1, params.N), np.int32)
    tiny[0, 1] = tiny[0, 2] = 1
    assert np.array_equal(engine.fft_u32_batch(engine.ifft_i32_batch(tiny)).astype(np.int32), tiny)


def test_bk_spectra_match_oracle(engine, keys):
    got = engine.export_bk_fft()
    assert got.tobytes() == keys.bk_f.tobytes()


def test_external_product_golden_and_oracle(engine, orc, params, keys, gold_gate):
    out = engine.external_product_batch(gold_gate["ep_idx"], gold_gate["ep_in"])
    assert np.array_equal(out.reshape(gold_gate["ep_out"].shape), gold_gate["ep_out"])
    rng = np.random.default_rng(14)
    idx = rng.integers(0, params.n, 24).astype(np.int32)
    trlwe = rng.integers(0, 2 ** 32, (24, 2 * params.N), dtype=np.uint64).astype(np.uint32)
    trlwe[1] = 0
    out = engine.external_product_batch(idx, trlwe)
    pl = orc.Plan(params.N)
    w = params.trgsw_words
    exp = np.stack([orc.external_product(params, pl, keys.bk_f[i * w:(i + 1) * w], None, t) for i, t in zip(idx, trlwe)])
    assert np.array_equal(out.reshape(exp.shape), exp)


@pytest.mark.parametrize("steps", [0, 1, 3, 17])
def test_blind_rotate_prefix(engine, orc, params, keys, gold_gate, steps):
    t = np.stack([orc.gate_linear(params, orc.NAND, a, b) for a, b in zip(gold_gate["in0"][:4], gold_gate["in1"][:4])])
    acc = engine.blind_rotate_batch(t, steps)
    pl = orc.Plan(params.N)
    exp = np.stack([orc.blind_rotate(params, pl, keys.bk_f, None, x, steps) for x in t])
    assert np.array_equal(acc.reshape(exp.shape), exp)
    if steps == 3:
        assert np.array_equal(acc[0].reshape(-1), gold_gate["acc_steps3"])


def test_blind_rotate_full_golden(engine, orc, params, gold_gate):
    t = orc.gate_linear(params, orc.NAND, gold_gate["in0"][0], gold_gate["in1"][0])[None]
    acc = engine.blind_rotate_batch(t)
    assert np.array_equal(acc.reshape(-1), gold_gate["acc_full"])


def test_key_switch_golden_and_oracle(engine, orc, params, keys, gold_gate):
    rng = np.random.default_rng(15)
    t1 = rng.integers(0, 2 ** 32, (9, params.N + 1), dtype=np.uint64).astype(np.uint32)
    t1[0] = gold_gate["extract"]
    t1[1, :params.N] = 0                      # every digit zero: only b' survives
    t1[2, :params.N] = 0xFFFFFFFF             # rounding carries out of the top digit
    out = engine.key_switch_batch(t1)
    exp = np.stack([orc.key_switch(params, keys.ksk, x) for x in t1])
    assert np.array_equal(out, exp)
    assert np.array_equal(out[0], gold_gate["out"][0])
    assert np.array_equal(out[1][:params.n], np.zeros(params.n, np.uint32)) and out[1][params.n] == t1[1, params.N]


def test_reference_shaped_ksk_reproduces_the_golden_gates(orc, params, keys, gold_gate):
    """rtfhe_load_ksk_ref takes KeySwitchingKey(Vec<[[TLWERep; IKS_T = 4]; IKS_L = 8]>) flattened as the reference holds it
    (hom_nand/src/tlwe.rs:178-180, 243-245, get(i, l, t) = [i][l][t-1]): a 4-wide key built by the oracle -- the golden key set's
    rows plus the never-read entries t = 4 -- must give the golden gates word for word, and the stage-level key switch must agree
    with the oracle reading the 4-wide key."""
    import rustfhe_amd as R
    ksk_ref = keys.ksk_ref()
    base = 1 << params.ks_basebit
    assert ksk_ref.size == params.N * params.ks_t * base * (params.n + 1)
    k4 = ksk_ref.reshape(params.N, params.ks_t, base, params.n + 1)
    assert np.array_equal(k4[:, :, :base - 1].reshape(-1), keys.ksk)           # entries t = 1 .. 3 are the compact key's rows
    # entry t = 4 is a real encryption of 4 s_i / 4^(l+1) (phase within the key-switch noise), not padding
    ph = int(keys.phase(k4[5, 2, base - 1]))
    want = ((int(keys.key1[5]) * base) << (32 - params.ks_basebit * 3)) % 2 ** 32
    assert abs(((ph - want + 2 ** 31) % 2 ** 32) - 2 ** 31) < 2 ** 22
    p = R.Params(n=params.n, N=params.N, l=params.l, bgbit=params.bgbit, ks_t=params.ks_t, ks_basebit=params.ks_basebit)
    e = R.Engine(p, 0)
    try:
        e.load_bk_torus(keys.bk_t)
        e.load_ksk_ref(ksk_ref)
        ops, in0, in1 = gold_gate["ops"], gold_gate["in0"], gold_gate["in1"]
        for g in range(len(ops)):
            out = e.gate_batch(int(ops[g]), in0[g:g + 1], in1[g:g + 1])
            assert np.array_equal(out[0], gold_gate["out"][g]), "gate %d (op %d)" % (g, ops[g])
        rng = np.random.default_rng(16)
        t1 = rng.integers(0, 2 ** 32, (5, params.N + 1), dtype=np.uint64).astype(np.uint32)
        t1[0] = gold_gate["extract"]
        t1[1, :params.N] = 0xFFFFFFFF
        out = e.key_switch_batch(t1)
        assert np.array_equal(out, np.stack([orc.key_switch_ref(params, ksk_ref, x) for x in t1]))
        assert np.array_equal(out[0], gold_gate["out"][0])
    finally:
        e.close()


def test_gates_golden(engine, gold_gate):
    ops, in0, in1 = gold_gate["ops"], gold_gate["in0"], gold_gate["in1"]
    for g in range(len(ops)):
        out = engine.gate_batch(int(ops[g]), in0[g:g + 1], in1[g:g + 1])
        assert np.array_equal(out[0], gold_gate["out"][g]), "gate %d (op %d)" % (g, ops[g])


def test_mux_golden(engine, gold_gate):
    out = engine.mux_batch(gold_gate["in0"][2:3], gold_gate["in0"][0:1], gold_gate["in1"][1:2])
    assert np.array_equal(out[0], gold_gate["mux_out"])


def test_truth_tables_all_gates(engine, orc, params, keys):
    """The reference's end-to-end check (hom_nand/src/tfhe.rs:164-278, examples/homnand-bench.rs:22-136):
    every gate, every input pair, decrypts to its truth table -- here also bit-exact against the oracle."""
    import rustfhe_amd as R
    b0 = [0, 0, 1, 1]
    b1 = [0, 1, 0, 1]
    c0, c1 = keys.encrypt_bits(b0), keys.encrypt_bits(b1)
    table = {R.NAND: [1, 1, 1, 0], R.AND: [0, 0, 0, 1], R.OR: [0, 1, 1, 1], R.XOR: [0, 1, 1, 0]}
    pl = orc.Plan(params.N)
    for op, tt in table.items():
        out = engine.gate_batch(op, c0, c1)
        assert keys.decrypt_bits(out) == tt
        exp = np.stack([orc.gate(params, pl, op, keys.bk_f, None, keys.ksk, a, b) for a, b in zip(c0, c1)])
        assert np.array_equal(out, exp)
    out = engine.gate_batch(R.NOT, c0[1:3])
    assert keys.decrypt_bits(out) == [1, 0]
    # trivial (noiseless) inputs, as the reference's nander front-end feeds them (AsLogic, tlwe.rs:80-87)
    triv = np.zeros((2, params.n + 1), np.uint32)
    triv[0, params.n], triv[1, params.n] = 0xE0000000, 0x20000000
    out = engine.gate_batch(R.NAND, triv[[0, 0, 1, 1]], triv[[0, 1, 0, 1]])
    assert keys.decrypt_bits(out) == [1, 1, 1, 0]


def test_batch_1024_bit_exact_vs_oracle(engine, orc, params, keys):
    """BASELINE config 2: 1024 independent HomNAND gates on one GPU, every output word compared with
    the CPU oracle run on the same inputs."""
    import os
    import rustfhe_amd as R
    rng = np.random.default_rng(16)
    b0, b1 = rng.integers(0, 2, 1024), rng.integers(0, 2, 1024)
    c0, c1 = keys.encrypt_bits(b0), keys.encrypt_bits(b1)
    out = engine.gate_batch(R.NAND, c0, c1)
    assert keys.decrypt_bits(out) == list(1 - (b0 & b1))
    exp, _ = orc.gate_batch_mt(params, orc.NAND, keys.bk_f, None, keys.ksk, c0, c1, nthreads=min(32, os.cpu_count() or 1))
    assert np.array_equal(out, exp)


def test_ragged_and_empty_batches(engine, params, keys):
    import rustfhe_amd as R
    c = keys.encrypt_bits([1, 0, 1, 1, 0])
    assert engine.gate_batch(R.NAND, c[:0], c[:0]).shape == (0, params.n + 1)
    full = engine.gate_batch(R.NAND, c, c[::-1].copy())
    for k in (1, 2, 3, 5):      # counts that do not fill a workgroup of 4 gates
        part = engine.gate_batch(R.NAND, c[:k], c[::-1][:k].copy())
        assert np.array_equal(part, full[:k])


def test_error_paths(engine, params):
    import rustfhe_amd as R
    with pytest.raises(R.RtfheError):
        engine._ck(engine.L.rtfhe_gate_batch(engine.h, 99, None, None, None, 1))
    with pytest.raises(R.RtfheError):
        R.Engine(R.Params(N=512))
    fresh = R.Engine(R.Params())
    with pytest.raises(R.RtfheError) as ei:
        fresh.gate_batch(R.NAND, np.zeros((1, params.n + 1), np.uint32), np.zeros((1, params.n + 1), np.uint32))
    assert ei.value.code == R._ffi.ERR_STATE
    fresh.close()


@pytest.mark.parametrize("force", ["1", "2", "4", "8"])
def test_all_kernel_shapes_match_golden(params, keys, gold_gate, force, monkeypatch):
    """The three launch shapes (workgroup-per-gate, 4-wave and 8-wave wave-per-gate) are the same arithmetic:
    each one, forced through RTFHE_FORCE_WAVES, reproduces the golden gates bit for bit."""
    import rustfhe_amd as R
    monkeypatch.setenv("RTFHE_FORCE_WAVES", force)
    e = R.Engine(R.Params(), 0)
    try:
        e.load_bk_fft(keys.bk_f)          # also covers the FrrSeries-domain key import (reference's BootstrappingKey form)
        e.load_ksk(keys.ksk)
        ops, in0, in1 = gold_gate["ops"], gold_gate["in0"], gold_gate["in1"]
        nand = [g for g in range(len(ops)) if ops[g] == R.NAND]
        out = e.gate_batch(R.NAND, in0[nand], in1[nand])
        assert np.array_equal(out, gold_gate["out"][nand])
        for g in range(len(ops)):
            if ops[g] != R.NAND:
                assert np.array_equal(e.gate_batch(int(ops[g]), in0[g:g + 1], in1[g:g + 1])[0], gold_gate["out"][g])
        t = np.stack([keys.encrypt_bits([0])[0]] * 3)
        acc = e.blind_rotate_batch(t, 2)
        assert np.array_equal(acc[0], acc[1]) and np.array_equal(acc[0], acc[2])
    finally:
        e.close()


def test_segmented_dispatch_matches_single_launch(engine, params, keys, monkeypatch):
    """Batches that are not whole rounds of 4 gates per CU are split into a two-waves-per-gate launch plus a remainder
    launch (latency kernel for <= 2 gates per CU, one more round otherwise): same words as one wave-per-gate launch of
    the whole batch, for plain batches and for blind-rotate outputs (different output stride)."""
    import rustfhe_amd as R
    rng = np.random.default_rng(77)
    count = 1024 + 700
    b0, b1 = rng.integers(0, 2, count), rng.integers(0, 2, count)
    c0, c1 = keys.encrypt_bits(b0), keys.encrypt_bits(b1)
    monkeypatch.setenv("RTFHE_FORCE_WAVES", "4")
    single = R.Engine(R.Params(), 0)
    try:
        single.load_bk_fft(keys.bk_f)
        single.load_ksk(keys.ksk)
        for cnt in (1024 + 100, 1024 + 700):
            out = engine.gate_batch(R.XOR, c0[:cnt], c1[:cnt])
            assert keys.decrypt_bits(out) == list(b0[:cnt] ^ b1[:cnt])
            assert np.array_equal(out, single.gate_batch(R.XOR, c0[:cnt], c1[:cnt]))
        acc = engine.blind_rotate_batch(c0[:1024 + 8], 3)
        assert np.array_equal(acc, single.blind_rotate_batch(c0[:1024 + 8], 3))
    finally:
        single.close()


def test_twiddle_import_retransforms_a_torus_form_key(params, keys, gold_gate):
    """rtfhe_set_twiddles after rtfhe_load_bk_torus: the key's spectra are recomputed with the imported tables (importing
    a perturbed table and then the original one must give the golden words again)."""
    import rustfhe_amd as R
    e = R.Engine(R.Params(), 0)
    try:
        e.load_bk_torus(keys.bk_t)
        e.load_ksk(keys.ksk)
        ifft, fft = e.twiddles()
        g = [k for k in range(len(gold_gate["ops"])) if gold_gate["ops"][k] == R.NAND][:4]
        in0, in1, want = gold_gate["in0"][g], gold_gate["in1"][g], gold_gate["out"][g]
        assert np.array_equal(e.gate_batch(R.NAND, in0, in1), want)
        bad = ifft.copy()
        bad[8] = np.nextafter(bad[8], 2.0)                 # one forward-twist cosine off by one ulp
        e.set_twiddles(bad, fft)
        e.set_twiddles(ifft, fft)
        assert np.array_equal(e.gate_batch(R.NAND, in0, in1), want)
        # The bootstrap kernels skip the multiplies of the butterfly whose twiddle is the first entry of the halfnn = 4 stage, which every
        # table the reference builds holds as exactly (cos 0, sin 0) = (1, 0).  A table with anything else there is refused, and the refusal
        # leaves the context as it was.  (Reference layout: per stage blocks of 4 cosines then 4 sines; the halfnn = 4 stage is the last
        # 8 doubles written of the forward table and the first 8 of the inverse table.)
        N = params.N
        fwd_at = N + 2 * (N // 2 - 8)                           # twist (N doubles) + the stages halfnn = N/4 .. 8 (2 halfnn doubles each)
        assert ifft[fwd_at] == 1.0 and ifft[fwd_at + 4] == 0.0 and fft[0] == 1.0 and fft[4] == 0.0
        for table, at in ((ifft, fwd_at), (fft, 0)):
            for off in (0, 4):
                broken = table.copy()
                broken[at + off] = np.nextafter(broken[at + off], 2.0)
                with pytest.raises(R.RtfheError):
                    e.set_twiddles(broken if table is ifft else ifft, broken if table is fft else fft)
                a, b = e.twiddles()
                assert a.tobytes() == ifft.tobytes() and b.tobytes() == fft.tobytes()
        assert np.array_equal(e.gate_batch(R.NAND, in0, in1), want)
    finally:
        e.close()


def test_inverse_transform_beyond_2_pow_51(engine, orc, params):
    """rtfhe_fft_u32_batch replaces Spqlios_fft_u32, whose Torus32(int64_t(x)) is defined up to 2^63
    (fft_processor_spqlios.cpp:182); the gate path stays below 2^50 but this entry point takes arbitrary spectra."""
    rng = np.random.default_rng(77)
    pl = orc.Plan(params.N)
    rows = []
    for e in (40, 50, 52, 55, 58, 60):
        rows.append(rng.standard_normal(params.N) * float(2 ** e))
    spec = np.stack(rows)
    spec[1, :4] = [2.0 ** 61, -(2.0 ** 61), 2.0 ** 53 + 2.0, -(2.0 ** 52) - 1.0]
    res = engine.fft_u32_batch(spec)
    exp = np.stack([pl.fft_u32(s) for s in spec])
    vals = np.stack([pl.fft_f64(s) for s in spec]) if hasattr(pl, "fft_f64") else None
    if vals is not None:
        assert np.abs(vals).max() < 2.0 ** 63 and np.abs(vals).max() > 2.0 ** 52      # in the reference's range, past the old one
    assert np.array_equal(res, exp)


def test_dev_entry_points_refuse_host_pointers(engine, params, keys):
    """A pageable host pointer handed to a *_dev call would make the kernel fault the GPU; the ABI checks what kind of memory
    it was given and refuses before launching (pinned rtfhe_host_alloc memory is GPU-addressable and accepted)."""
    import torch
    import rustfhe_amd as R
    c = keys.encrypt_bits([1, 0, 1])
    d = torch.from_numpy(c.view(np.int32)).cuda()
    out = torch.empty_like(d)
    with pytest.raises(R.RtfheError) as ei:
        engine.gate_batch_dev(R.NAND, int(c.ctypes.data), d, out, 3)
    assert ei.value.code == R._ffi.ERR_INVALID and "device pointers" in str(ei.value)
    with pytest.raises(R.RtfheError):
        engine.gate_batch_dev(R.NAND, d, d, int(c.ctypes.data), 3)
    pin = R.pinned_empty(c.shape)
    pin[:] = c
    engine.gate_batch_dev(R.NAND, int(pin.ctypes.data), d, out, 3)
    engine.sync()
    assert keys.decrypt_bits(out.cpu().numpy().view(np.uint32)) == [0, 1, 0]


def test_rest_of_the_reference_fft_ffi(engine, orc, params):
    """Spqlios_ifft / Spqlios_fft on doubles and Spqlios_poly_mul (spqlios-wrapper.cpp:18-20, 30-32, 38-53; off the gate path):
    bit-exact against the oracle's restatement; the reference KAT of poly_mul's meaning (X + X^2)^2 = X^2 + 2 X^3 + X^4
    (utils/src/spqlios.rs:243-276, there at N = 16) holds at N = 1024 too."""
    rng = np.random.default_rng(31)
    pl = orc.Plan(params.N)
    src = rng.standard_normal((5, params.N)) * 2.0 ** rng.integers(0, 40, (5, 1))
    assert engine.ifft_f64_batch(src).tobytes() == np.stack([pl.ifft_f64(s) for s in src]).tobytes()
    spec = np.stack([pl.ifft_i32(rng.integers(-2 ** 31, 2 ** 31, params.N).astype(np.int32)) for _ in range(5)])
    assert engine.fft_f64_batch(spec).tobytes() == np.stack([pl.fft_f64(s) for s in spec]).tobytes()
    a = rng.integers(0, 2 ** 32, (6, params.N), dtype=np.uint64).astype(np.uint32)
    b = rng.integers(-64, 64, (6, params.N)).astype(np.int32).view(np.uint32)          # small multiplier: the product stays exact-ish
    got = engine.poly_mul_batch(a, b)
    assert np.array_equal(got, np.stack([pl.poly_mul(x, y) for x, y in zip(a, b)]))
    p = np.zeros((1, params.N), np.uint32)
    p[0, 1] = p[0, 2] = 1
    sq = engine.poly_mul_batch(p, p)[0]
    want = np.zeros(params.N, np.uint32)
    want[2], want[3], want[4] = 1, 2, 1
    assert np.array_equal(sq, want)
