"""GPU: batches between whole rounds at N = 1024.  A last whole round of 4 x CUs gates and the remainder behind it run as ONE launch with five or
six gates on the four wave pairs of every CU (k_bootstrap_pair_rr: the gates' CMUX steps time-sliced over the pairs, a gate changing hands from
step to step through LDS).  The arithmetic is k_bootstrap_pair's, so every word must equal what whole rounds + a tail give (RTFHE_PAIR_RR=0:
the dispatch of rounds 1-5) -- in every mode of the kernel -- and the oracle's."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _engine(R, params, monkeypatch, env):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    e = R.Engine(R.Params(n=params.n, N=params.N, l=params.l, bgbit=params.bgbit, ks_t=params.ks_t, ks_basebit=params.ks_basebit), 0)
    for k in env:
        monkeypatch.delenv(k)
    return e


def test_time_sliced_launch_equals_whole_rounds_and_the_oracle(engine, orc, params, keys, monkeypatch):
    import rustfhe_amd as R
    import torch
    cus = torch.cuda.get_device_properties(0).multi_processor_count      # 256 on an MI355X
    rng = np.random.default_rng(4096)
    G = 3 * 4 * cus + 2 * cus                      # three whole rounds + two gates per CU
    c0 = rng.integers(0, 2 ** 32, (G, params.n + 1), dtype=np.uint64).astype(np.uint32)      # arbitrary TLWE words
    c1 = rng.integers(0, 2 ** 32, (G, params.n + 1), dtype=np.uint64).astype(np.uint32)
    whole = _engine(R, params, monkeypatch, {"RTFHE_PAIR_RR": "0"})
    fused = _engine(R, params, monkeypatch, {"RTFHE_KS_MM_MIN": "0"})                      # key switch inside the bootstrap kernel (MODE_GATE)
    fused_whole = _engine(R, params, monkeypatch, {"RTFHE_KS_MM_MIN": "0", "RTFHE_PAIR_RR": "0"})
    try:
        for e in (whole, fused, fused_whole):
            e.load_bk_torus(keys.bk_t)
            e.load_ksk(keys.ksk)
        ref = whole.gate_batch(R.NAND, c0, c1)
        # one gate past a round, every CU with five, between five and six, every CU with six, one gate too many for the launch and seven per CU (whole rounds
        # + a tail again), and the same behind one and three whole rounds
        sizes = [4 * cus + 1, 4 * cus + 70, 5 * cus - 1, 5 * cus, 5 * cus + 1, 5 * cus + 99, 6 * cus - 1, 6 * cus, 6 * cus + 1, 7 * cus - 1, 7 * cus,
                 7 * cus + 1, 8 * cus + 1, 9 * cus, 10 * cus, 11 * cus - 5, 12 * cus + 3, G]
        for k in sizes:
            out = engine.gate_batch(R.NAND, c0[:k], c1[:k])
            assert np.array_equal(out, ref[:k]), k
        ref_f = fused_whole.gate_batch(R.XOR, c0[:7 * cus], c1[:7 * cus])
        for k in (4 * cus + 1, 5 * cus, 5 * cus + 7, 6 * cus, 7 * cus - 2):
            assert np.array_equal(fused.gate_batch(R.XOR, c0[:k], c1[:k]), ref_f[:k]), k
        assert np.array_equal(whole.gate_batch(R.XOR, c0[:7 * cus], c1[:7 * cus]), ref_f)
        # blind-rotation prefixes (MODE_BLIND_ROTATE) and the raw bootstrap
        k = 5 * cus + 33
        t = np.stack([orc.gate_linear(params, orc.NAND, x, y) for x, y in zip(c0[:k], c1[:k])])
        pl = orc.Plan(params.N)
        for steps in (1, 2, 7):
            got = engine.blind_rotate_batch(t, steps)
            assert np.array_equal(got, whole.blind_rotate_batch(t, steps)), steps
            for g in (0, 4 * cus + 5, k - 1):
                assert np.array_equal(got[g].reshape(-1), orc.blind_rotate(params, pl, keys.bk_f, None, t[g], steps)), (steps, g)
        assert np.array_equal(engine.bootstrap_batch(t), whole.bootstrap_batch(t))
        # whole gates against the oracle, picked from every part of a five- and a six-gates-per-CU launch
        for k in (5 * cus - 3, 6 * cus - 3):
            out = engine.gate_batch(R.NAND, c0[:k], c1[:k])
            for g in (0, 4, 5, 6, 7, 4 * cus - 1, 4 * cus + 9, k - 1):
                assert np.array_equal(out[g], orc.gate(params, pl, orc.NAND, keys.bk_f, None, keys.ksk, c0[g], c1[g])), (k, g)
        # encrypted bits decrypt to the truth table
        b0, b1 = rng.integers(0, 2, 5 * cus + 11), rng.integers(0, 2, 5 * cus + 11)
        e0, e1 = keys.encrypt_bits(b0), keys.encrypt_bits(b1)
        assert keys.decrypt_bits(engine.gate_batch(R.OR, e0, e1)) == list(b0 | b1)
    finally:
        whole.close(); fused.close(); fused_whole.close()


def test_time_sliced_launch_in_a_netlist_wave(engine, keys):
    """A netlist wave of more than a round of gates (wire-table indices, per-gate opcodes) through the time-sliced launch: one gate with a wire
    index out of range is skipped and reported, its neighbours run."""
    import torch
    import rustfhe_amd as R
    n1 = engine.p.n + 1
    rng = np.random.default_rng(9)
    cnt = 1024 + 200
    bits = rng.integers(0, 2, 64)
    W = 64 + cnt
    wires = torch.zeros((W, n1), dtype=torch.int32, device="cuda")
    wires[:64] = torch.from_numpy(keys.encrypt_bits(bits).view(np.int32)).cuda()
    i0, i1 = rng.integers(0, 64, cnt), rng.integers(0, 64, cnt)
    ops = rng.choice([R.NAND, R.AND, R.OR, R.XOR], cnt)
    bad = 1024 + 100
    i0[bad] = 10 ** 7
    dev = lambda v: torch.tensor(np.asarray(v), dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    engine.circuit_wave_dev(dev(ops), dev(i0), dev(i1), dev(64 + np.arange(cnt)), wires, W, cnt, st)
    with pytest.raises(R.RtfheError) as ei:
        engine.sync(st)
    assert ei.value.code == R._ffi.ERR_INVALID
    out = wires[64:].cpu().numpy().view(np.uint32)
    assert not out[bad].any()
    x, y = bits[np.where(np.arange(cnt) == bad, 0, i0)], bits[i1]
    exp = np.select([ops == R.NAND, ops == R.AND, ops == R.OR, ops == R.XOR], [1 - (x & y), x & y, x | y, x ^ y])
    keep = np.arange(cnt) != bad
    assert list(np.asarray(keys.decrypt_bits(out[keep]))) == list(exp[keep])


def test_time_sliced_launch_on_the_split_fft_exact_backend(orc, params, keys, monkeypatch):
    """The same launch shape on the split-FFT exact backend (k_bootstrap_xpair_rr): the words of whole rounds + tail, of the NTT backend (exact
    products have one value), in the split and the fused key-switch form and for blind-rotation prefixes."""
    import rustfhe_amd as R
    import torch
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    rng = np.random.default_rng(77)
    G = 2 * 4 * cus + 2 * cus
    c0 = rng.integers(0, 2 ** 32, (G, params.n + 1), dtype=np.uint64).astype(np.uint32)
    c1 = rng.integers(0, 2 ** 32, (G, params.n + 1), dtype=np.uint64).astype(np.uint32)
    sliced = _engine(R, params, monkeypatch, {})
    whole = _engine(R, params, monkeypatch, {"RTFHE_PAIR_RR": "0"})
    fused = _engine(R, params, monkeypatch, {"RTFHE_KS_MM_MIN": "0"})
    try:
        for e in (sliced, whole, fused):
            e.load_bk_torus(keys.bk_t)
            e.load_ksk(keys.ksk)
            e.set_backend(R._ffi.BACKEND_FFT_SPLIT_EXACT)
        ref = whole.gate_batch(R.NAND, c0, c1)
        for k in (4 * cus + 1, 5 * cus - 1, 5 * cus, 5 * cus + 40, 6 * cus, 6 * cus + 1, 8 * cus + 9, G):
            assert np.array_equal(sliced.gate_batch(R.NAND, c0[:k], c1[:k]), ref[:k]), k
        for k in (4 * cus + 3, 6 * cus - 1):
            assert np.array_equal(fused.gate_batch(R.NAND, c0[:k], c1[:k]), ref[:k]), k
        k = 5 * cus + 17
        t = np.stack([orc.gate_linear(params, orc.NAND, x, y) for x, y in zip(c0[:k], c1[:k])])
        for steps in (1, 3):
            assert np.array_equal(sliced.blind_rotate_batch(t, steps), whole.blind_rotate_batch(t, steps)), steps
        whole.set_backend(R._ffi.BACKEND_NTT_EXACT)
        k = 5 * cus + 40
        assert np.array_equal(sliced.gate_batch(R.XOR, c0[:k], c1[:k]), whole.gate_batch(R.XOR, c0[:k], c1[:k]))
        b0, b1 = rng.integers(0, 2, 6 * cus - 5), rng.integers(0, 2, 6 * cus - 5)
        assert keys.decrypt_bits(sliced.gate_batch(R.AND, keys.encrypt_bits(b0), keys.encrypt_bits(b1))) == list(b0 & b1)
    finally:
        sliced.close(); whole.close(); fused.close()
