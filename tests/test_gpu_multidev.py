"""The multi-device context (rtfhe_ctx_create_multi) with n_dev > 1 on a ONE-GPU box: the test-only switch
RTFHE_TEST_ALLOW_DUP_DEVICES=1 lets device 0 appear several times, so that everything behind the call runs -- one full context per
entry, keys transformed once and copied device-to-device (hipMemcpyPeer), one host thread + stream per entry, contiguous shard
ranges [count d / D, count (d+1) / D), error aggregation -- and every result is compared word for word with the single-device
engine.  (A node with several GPUs runs exactly this code with distinct ids; that run is the driver's.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pair(R, p, bk, ksk, devices, monkeypatch, bk_fft=None):
    monkeypatch.setenv("RTFHE_TEST_ALLOW_DUP_DEVICES", "1")
    m = R.Engine(p, devices=devices)
    monkeypatch.delenv("RTFHE_TEST_ALLOW_DUP_DEVICES")
    s = R.Engine(p, 0)
    for e in (m, s):
        e.load_bk_torus(bk)
        e.load_ksk(ksk)
    return m, s


def test_duplicate_ids_are_refused_without_the_switch(params):
    import rustfhe_amd as R
    with pytest.raises(R.RtfheError):
        R.Engine(R.Params(), devices=[0, 0])


@pytest.mark.parametrize("n_dev", [2, 3])
def test_sharded_host_batches_equal_single_device_n1024(params, keys, gold_gate, monkeypatch, n_dev):
    import rustfhe_amd as R
    p = R.Params()
    m, s = _pair(R, p, keys.bk_t, keys.ksk, [0] * n_dev, monkeypatch)
    try:
        assert m.device_count() == n_dev and s.device_count() == 1
        ops, in0, in1 = gold_gate["ops"], gold_gate["in0"], gold_gate["in1"]
        # the golden gates as ONE ragged batch per opcode group and one by one (count = 1 < n_dev: all but one shard are empty)
        for g in range(len(ops)):
            assert np.array_equal(m.gate_batch(int(ops[g]), in0[g:g + 1], in1[g:g + 1])[0], gold_gate["out"][g])
        rng = np.random.default_rng(300 + n_dev)
        b0, b1 = rng.integers(0, 2, 301), rng.integers(0, 2, 301)
        c0, c1 = keys.encrypt_bits(b0), keys.encrypt_bits(b1)
        ref = s.gate_batch(R.NAND, c0, c1)
        for k in (1, 2, n_dev - 1, n_dev, n_dev + 1, 7, 100, 301):      # not divisible by n_dev, smaller than n_dev, ...
            assert np.array_equal(m.gate_batch(R.NAND, c0[:k], c1[:k]), ref[:k]), k
        assert keys.decrypt_bits(ref) == list(1 - (b0 & b1))
        assert np.array_equal(m.gate_batch(R.NOT, c0[:5]), s.gate_batch(R.NOT, c0[:5]))
        assert np.array_equal(m.bootstrap_batch(c0[:4]), s.bootstrap_batch(c0[:4]))
        assert np.array_equal(m.mux_batch(c0[:5], c1[:5], c0[5:10]), s.mux_batch(c0[:5], c1[:5], c0[5:10]))
        assert np.array_equal(m.mux_batch(in0[2:3], in0[0:1], in1[1:2])[0], gold_gate["mux_out"])
        for steps in (0, 3):
            assert np.array_equal(m.blind_rotate_batch(c0[:4], steps), s.blind_rotate_batch(c0[:4], steps))
        # the exact-integer NTT backend on every shard (its key is derived per device from the replicated torus key)
        m.set_backend(R._ffi.BACKEND_NTT_EXACT); s.set_backend(R._ffi.BACKEND_NTT_EXACT)
        for k in (1, 5):
            assert np.array_equal(m.gate_batch(R.XOR, c0[:k], c1[:k]), s.gate_batch(R.XOR, c0[:k], c1[:k]))
        assert np.array_equal(m.blind_rotate_batch(c0[:3], 2), s.blind_rotate_batch(c0[:3], 2))
        m.set_backend(R._ffi.BACKEND_FFT64_MIRROR); s.set_backend(R._ffi.BACKEND_FFT64_MIRROR)
        assert np.array_equal(m.gate_batch(R.AND, c0[:3], c1[:3]), s.gate_batch(R.AND, c0[:3], c1[:3]))
        # the ranges the shards took are the ones rtfhe_shard_range reports
        lo, hi = R.shard_range(301, n_dev - 1, n_dev)
        assert hi == 301 and np.array_equal(m.gate_batch(R.NAND, c0, c1)[lo:hi], ref[lo:hi])
        with pytest.raises(R.RtfheError):
            m.gate_batch(99, c0[:4], c1[:4])
    finally:
        m.close(); s.close()


def test_sharded_host_batches_equal_single_device_n2048(monkeypatch):
    """N = 2048 (BASELINE config 5's ring) with a small TLWE dimension so that key generation stays short: the halves-layout key
    and the NTT tables are rebuilt on every peer (build_halves_bk, ntt_prepare)."""
    import rustfhe_amd as R
    p = R.Params(N=2048, n=24)
    key0, key1, bk, ksk = R.keygen(p, 2048)
    m, s = _pair(R, p, bk, ksk, [0, 0], monkeypatch)
    try:
        rng = np.random.default_rng(5)
        b0, b1 = rng.integers(0, 2, 9).astype(np.uint8), rng.integers(0, 2, 9).astype(np.uint8)
        c0, c1 = R.encrypt_bits(p, key0, b0, 1), R.encrypt_bits(p, key0, b1, 2)
        for k in (1, 2, 9):
            out = m.gate_batch(R.NAND, c0[:k], c1[:k])
            assert np.array_equal(out, s.gate_batch(R.NAND, c0[:k], c1[:k])), k
        assert list(R.decrypt_bits(p, key0, m.gate_batch(R.NAND, c0, c1))) == list(1 - (b0 & b1))
        assert np.array_equal(m.mux_batch(c0[:3], c1[:3], c0[3:6]), s.mux_batch(c0[:3], c1[:3], c0[3:6]))
        assert np.array_equal(m.blind_rotate_batch(c0[:3], 4), s.blind_rotate_batch(c0[:3], 4))
        m.set_backend(R._ffi.BACKEND_NTT_EXACT); s.set_backend(R._ffi.BACKEND_NTT_EXACT)
        assert np.array_equal(m.gate_batch(R.OR, c0[:3], c1[:3]), s.gate_batch(R.OR, c0[:3], c1[:3]))
    finally:
        m.close(); s.close()


def test_circuit_handle_survives_its_context(params, keys):
    """rtfhe_circuit_destroy after rtfhe_ctx_destroy used to read freed memory: the context now releases its circuits' graphs and
    detaches them; a late launch fails with RTFHE_ERR_STATE, a late destroy only frees the handle."""
    import torch
    import rustfhe_amd as R
    e = R.Engine(R.Params(), 0)
    e.load_bk_torus(keys.bk_t)
    e.load_ksk(keys.ksk)
    c = keys.encrypt_bits([1, 0])
    n1 = params.n + 1
    wires = torch.zeros((3, n1), dtype=torch.int32, device="cuda")
    wires[:2] = torch.from_numpy(c.view(np.int32)).cuda()
    i32 = lambda v: torch.tensor(v, dtype=torch.int32, device="cuda")
    ops, i0, i1, io = i32([R.NAND]), i32([0]), i32([1]), i32([2])
    circ = e.circuit_create(ops, i0, i1, io, np.array([0, 1], np.int32), wires, 3)
    e.circuit_launch(circ)
    e.sync()
    assert keys.decrypt_bits(wires[2:3].cpu().numpy().view(np.uint32)) == [1]
    L, h = e.L, e.h
    e.h = None
    L.rtfhe_ctx_destroy(h)                       # context first ...
    assert L.rtfhe_circuit_launch(circ, None) == R._ffi.ERR_STATE
    L.rtfhe_circuit_destroy(circ)                # ... circuit afterwards: no use-after-free
