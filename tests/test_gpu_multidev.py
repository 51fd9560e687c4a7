"""The multi-device context (rtfhe_ctx_create_multi) with n_dev > 1 on a ONE-GPU box: a device may be named more than once (every entry is
a full context of its own), so device 0 appears several times and everything behind the call runs -- one full context per entry, keys
transformed once and copied device-to-device (hipMemcpyPeer), contiguous shard ranges [count d / D, count (d+1) / D), one host thread +
stream per entry for host batches, peer pulls / pushes and cross-stream events for batches resident on the primary, error aggregation --
and every result is compared word for word with the single-device engine.  (A node with several GPUs runs exactly this code with
distinct ids; that run is the driver's.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pair(R, p, bk, ksk, devices, bk_fft=None):
    m = R.Engine(p, devices=devices)
    s = R.Engine(p, 0)
    for e in (m, s):
        e.load_bk_torus(bk)
        e.load_ksk(ksk)
    return m, s


@pytest.mark.parametrize("n_dev", [2, 3])
def test_sharded_host_batches_equal_single_device_n1024(params, keys, gold_gate, n_dev):
    import rustfhe_amd as R
    p = R.Params()
    m, s = _pair(R, p, keys.bk_t, keys.ksk, [0] * n_dev)
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


def test_sharded_host_batches_equal_single_device_n2048():
    """N = 2048 (BASELINE config 5's ring) with a small TLWE dimension so that key generation stays short: the parity-split key
    layout and the NTT tables are rebuilt on every peer (ntt_prepare)."""
    import rustfhe_amd as R
    p = R.Params(N=2048, n=24)
    key0, key1, bk, ksk = R.keygen(p, 2048)
    m, s = _pair(R, p, bk, ksk, [0, 0])
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


@pytest.mark.parametrize("n_dev", [2, 3])
def test_a_batch_resident_on_the_primary_is_sharded_behind_the_c_abi(params, keys, n_dev):
    """rtfhe_gate_batch_dev / rtfhe_mux_batch_dev / rtfhe_bootstrap_batch_dev on a multi-device context: the batch lives on the primary; the
    other entries pull their ranges (hipMemcpyPeerAsync on their own streams), bootstrap them and push the outputs back while the primary
    computes its own range; the caller's stream waits for their events.  Word for word the single-device call at 8,192 and 65,536 gates
    (BASELINE config 3's per-GPU and whole-node counts), at ragged counts, with entries that get no gate, back to back on one stream
    without a host synchronisation in between, and with the inputs overwritten right behind the call (stream order must hold)."""
    import torch
    import rustfhe_amd as R
    p = R.Params()
    m, s = _pair(R, p, keys.bk_t, keys.ksk, [0] * n_dev)
    try:
        G = 65536
        rng = np.random.default_rng(77 + n_dev)
        base = 1031                                                   # distinct ciphertexts, tiled: every shard boundary falls between different rows
        b0, b1 = rng.integers(0, 2, base), rng.integers(0, 2, base)
        c0, c1 = keys.encrypt_bits(b0), keys.encrypt_bits(b1)
        reps = (G + base - 1) // base
        d0 = torch.from_numpy(np.tile(c0, (reps, 1))[:G].view(np.int32)).cuda()
        d1 = torch.from_numpy(np.tile(c1, (reps, 1))[:G].view(np.int32)).cuda()
        st = torch.cuda.current_stream().cuda_stream
        ref = torch.empty_like(d0)
        s.gate_batch_dev(R.NAND, d0, d1, ref, G, st); s.sync(st)
        assert keys.decrypt_bits(ref[:base].cpu().numpy().view(np.uint32)) == list(1 - (b0 & b1))
        for count in (G, 8192, 8191, 1030, n_dev + 1, n_dev, n_dev - 1, 1, 0):
            out = torch.full_like(d0, -1)
            m.gate_batch_dev(R.NAND, d0, d1, out, count, st); m.sync(st)
            assert torch.equal(out[:count], ref[:count]), count
            assert bool((out[count:] == -1).all()), "rows beyond the batch were written"
        # unary gate and raw bootstrap (no second input), MUX (three inputs, two intermediate batches per entry)
        k = 3000
        for op in (R.NOT, R.COPY):
            a, b = torch.empty_like(d0[:k]), torch.empty_like(d0[:k])
            s.gate_batch_dev(op, d0, None, a, k, st); m.gate_batch_dev(op, d0, None, b, k, st); m.sync(st)
            assert torch.equal(a, b), op
        a, b = torch.empty_like(d0[:k]), torch.empty_like(d0[:k])
        s.bootstrap_batch_dev(d0, a, k, st); m.bootstrap_batch_dev(d0, b, k, st); m.sync(st)
        assert torch.equal(a, b)
        sel = d1[5:5 + k].contiguous()
        s.mux_batch_dev(sel, d0, d1, a, k, st); m.mux_batch_dev(sel, d0, d1, b, k, st); m.sync(st)
        assert torch.equal(a, b)
        assert np.array_equal(a[:7].cpu().numpy().view(np.uint32), s.mux_batch(sel[:7].cpu().numpy().view(np.uint32), c0[:7], c1[:7]))
        # two batches back to back, then the inputs clobbered on the same stream: what the peers pull must be what stream order promises
        x0, x1 = d0[:4096].clone(), d1[:4096].clone()
        o1, o2 = torch.empty_like(x0), torch.empty_like(x0)
        m.gate_batch_dev(R.NAND, x0, x1, o1, 4096, st)
        m.gate_batch_dev(R.XOR, o1, x1, o2, 4096, st)                 # reads the first batch's outputs, gathered from every entry
        x0.zero_(); x1.zero_()                                        # enqueued behind both batches
        m.sync(st)
        r1, r2 = torch.empty_like(o1), torch.empty_like(o1)
        s.gate_batch_dev(R.NAND, d0[:4096], d1[:4096], r1, 4096, st); s.gate_batch_dev(R.XOR, r1, d1[:4096], r2, 4096, st); s.sync(st)
        assert torch.equal(o1, r1) and torch.equal(o2, r2)
        # footprint: every entry holds its own key replica; the primary's staging is not used by a device-resident batch
        assert m.memory_bytes(1) > 0 and m.memory_bytes(0) > 0
        with pytest.raises(R.RtfheError):
            m.memory_bytes(n_dev)
    finally:
        m.close(); s.close()


def test_peer_info_says_what_the_runtime_reported(params, keys):
    """rtfhe_ctx_peer_info: every entry after the primary reports whether it names the primary's own device, the peer-access
    answers in both directions, whether enabling it worked, the link the runtime reports, and -- after a device-resident sharded batch has
    completed -- the three phases of its share (pull, bootstrap, push) from events on its own stream.  On this one-GPU box every entry names
    device 0: same_device, nothing to enable, no link; the phases are real."""
    import torch
    import rustfhe_amd as R
    p = R.Params()
    m, s = _pair(R, p, keys.bk_t, keys.ksk, [0, 0, 0])
    try:
        for d in (1, 2):
            i = m.peer_info(d)
            assert i["device"] == 0 and i["same_device"] == 1
            assert (i["can_access_from_primary"], i["can_access_to_primary"], i["enabled_from_primary"], i["enabled_to_primary"]) == (0, 0, 0, 0)
            assert i["link"] == "not reported" and i["scatter_ms"] is None and i["compute_ms"] is None and i["gather_ms"] is None
        for d in (0, 3, -1):
            with pytest.raises(R.RtfheError):
                m.peer_info(d)
        with pytest.raises(R.RtfheError):
            s.peer_info(1)                                 # a single-device context has no entry 1
        G = 2 * 8192
        rng = np.random.default_rng(9)
        b0, b1 = rng.integers(0, 2, 512), rng.integers(0, 2, 512)
        d0 = torch.from_numpy(np.tile(keys.encrypt_bits(b0), (G // 512, 1)).view(np.int32)).cuda()
        d1 = torch.from_numpy(np.tile(keys.encrypt_bits(b1), (G // 512, 1)).view(np.int32)).cuda()
        out, ref = torch.empty_like(d0), torch.empty_like(d0)
        st = torch.cuda.current_stream().cuda_stream
        m.gate_batch_dev(R.NAND, d0, d1, out, G, st); m.sync(st)
        s.gate_batch_dev(R.NAND, d0, d1, ref, G, st); s.sync(st)
        assert torch.equal(out, ref)
        lo, hi = R.shard_range(G, 1, 3)
        assert (lo, hi) == (G // 3, 2 * G // 3)            # [count d / D, count (d + 1) / D)
        for d in (1, 2):
            i = m.peer_info(d)
            assert i["scatter_ms"] is not None and i["scatter_ms"] >= 0 and i["gather_ms"] >= 0
            assert i["compute_ms"] > 1.0, i                # ~ 5,461 gates: tens of milliseconds
        # a batch that gives an entry no gate leaves its phases unreported: one gate over three entries is [0, 0), [0, 0), [0, 1)
        assert [R.shard_range(1, d, 3) for d in range(3)] == [(0, 0), (0, 0), (0, 1)]
        m.gate_batch_dev(R.NAND, d0, d1, out, 1, st); m.sync(st)
        assert m.peer_info(1)["compute_ms"] is None and m.peer_info(2)["compute_ms"] is not None
    finally:
        m.close(); s.close()


def test_batch_pointers_that_are_not_the_primarys_device_memory_take_the_default_copy(params, keys):
    """gpu_accessible admits pinned-host buffers as *_dev arguments; on a multi-device context the shards of such a batch may not be handed to
    hipMemcpyPeerAsync as memory of the primary (advisor r5): they are copied with hipMemcpyDefault and the results are the same words."""
    import ctypes as C
    import torch
    import rustfhe_amd as R
    p = R.Params()
    m, s = _pair(R, p, keys.bk_t, keys.ksk, [0, 0])
    try:
        k, w = 600, p.n + 1
        rng = np.random.default_rng(10)
        b0, b1 = rng.integers(0, 2, k), rng.integers(0, 2, k)
        c0, c1 = keys.encrypt_bits(b0), keys.encrypt_bits(b1)
        h0, h1, ho = R.pinned_empty((k, w)), R.pinned_empty((k, w)), R.pinned_empty((k, w))
        h0[:], h1[:], ho[:] = c0, c1, 0
        ptr = lambda a: C.c_void_p(a.ctypes.data)
        assert m.L.rtfhe_gate_batch_dev(m.h, R.NAND, ptr(h0), ptr(h1), ptr(ho), k, None) == 0
        m.sync()
        assert np.array_equal(np.asarray(ho), s.gate_batch(R.NAND, c0, c1))
        assert keys.decrypt_bits(np.asarray(ho)) == list(1 - (b0 & b1))
    finally:
        m.close(); s.close()


def test_the_split_fft_backend_on_every_entry(params, keys):
    """RTFHE_BACKEND_FFT_SPLIT_EXACT on a multi-device context: the split key spectra are derived per entry from the replicated torus key; a
    sharded device-resident batch equals the single-device one and the NTT backend's, word for word."""
    import torch
    import rustfhe_amd as R
    p = R.Params()
    m, s = _pair(R, p, keys.bk_t, keys.ksk, [0, 0])
    try:
        k = 1500
        rng = np.random.default_rng(12)
        b0, b1 = rng.integers(0, 2, k), rng.integers(0, 2, k)
        d0 = torch.from_numpy(keys.encrypt_bits(b0).view(np.int32)).cuda()
        d1 = torch.from_numpy(keys.encrypt_bits(b1).view(np.int32)).cuda()
        st = torch.cuda.current_stream().cuda_stream
        a, b, c = torch.empty_like(d0), torch.empty_like(d0), torch.empty_like(d0)
        s.set_backend(R._ffi.BACKEND_NTT_EXACT)
        s.gate_batch_dev(R.NAND, d0, d1, c, k, st); s.sync(st)
        m.set_backend(R._ffi.BACKEND_FFT_SPLIT_EXACT); s.set_backend(R._ffi.BACKEND_FFT_SPLIT_EXACT)
        s.gate_batch_dev(R.NAND, d0, d1, a, k, st); m.gate_batch_dev(R.NAND, d0, d1, b, k, st); m.sync(st); s.sync(st)
        assert torch.equal(a, b) and torch.equal(a, c)
        assert keys.decrypt_bits(a.cpu().numpy().view(np.uint32)) == list(1 - (b0 & b1))
    finally:
        m.close(); s.close()


def test_a_device_resident_batch_inside_a_callers_capture_stays_on_the_primary(params, keys):
    """Inside a stream capture the staging buffers of the other entries are not the capture's to bake in: the whole batch runs on the primary
    (fused kernel, canonical key layout), and the captured graph replays to the same words."""
    import torch
    import rustfhe_amd as R
    p = R.Params()
    m, s = _pair(R, p, keys.bk_t, keys.ksk, [0, 0])
    try:
        b0, b1 = [0, 1, 1, 0, 1], [1, 1, 0, 0, 1]
        d0 = torch.from_numpy(keys.encrypt_bits(b0).view(np.int32)).cuda()
        d1 = torch.from_numpy(keys.encrypt_bits(b1).view(np.int32)).cuda()
        ref, out = torch.empty_like(d0), torch.zeros_like(d0)
        s.gate_batch_dev(R.AND, d0, d1, ref, 5); s.sync()
        side = torch.cuda.Stream()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            m.gate_batch_dev(R.AND, d0, d1, out, 5, torch.cuda.current_stream().cuda_stream)
        g.replay(); torch.cuda.synchronize()
        assert torch.equal(out, ref)
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
