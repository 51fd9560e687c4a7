"""GPU: the remaining BASELINE.json configurations at (or at the per-GPU share of) their full sizes.
config 3: 65,536 gates over 8 GPUs = 8,192 per GPU -> one 8,192-gate launch here (8-wave workgroups, 2 waves/SIMD)
config 5: N = 2048 (NBIT = 11), l = 3 -- not a reference parameter set; parity is against the oracle."""
import os

import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu


def test_config2_batch_1024_every_output_bit_exact(engine, orc, params, keys):
    """BASELINE config 2 (the bench workload): 1,024 independent gates on one GPU, EVERY output compared bit for bit with the CPU
    oracle (all host threads, one independent gate stream each)."""
    import rustfhe_amd as R
    G = 1024
    rng = np.random.default_rng(1024)
    b0, b1 = rng.integers(0, 2, G), rng.integers(0, 2, G)
    c0, c1 = keys.encrypt_bits(b0), keys.encrypt_bits(b1)
    out = engine.gate_batch(R.NAND, c0, c1)
    exp, _ = orc.gate_batch_mt(params, orc.NAND, keys.bk_f, None, keys.ksk, c0, c1, nthreads=min(64, os.cpu_count() or 1))
    assert np.array_equal(out, exp)
    assert keys.decrypt_bits(out) == list(1 - (b0 & b1))


def test_config3_per_gpu_shard_8192(engine, orc, params, keys):
    import rustfhe_amd as R
    G = 8192
    rng = np.random.default_rng(33)
    b0, b1 = rng.integers(0, 2, G), rng.integers(0, 2, G)
    c0, c1 = keys.encrypt_bits(b0), keys.encrypt_bits(b1)
    out = engine.gate_batch(R.NAND, c0, c1)
    assert keys.decrypt_bits(out) == list(1 - (b0 & b1))
    # determinism and gate independence at full size: a permuted batch gives the permuted outputs, bit for bit
    perm = rng.permutation(G)
    out_p = engine.gate_batch(R.NAND, c0[perm], c1[perm])
    assert np.array_equal(out_p, out[perm])
    # bit-exact against the oracle on a random sample (all 8,192 on the CPU would take minutes)
    pick = rng.choice(G, 128, replace=False)
    exp, _ = orc.gate_batch_mt(params, orc.NAND, keys.bk_f, None, keys.ksk, c0[pick], c1[pick], nthreads=min(32, os.cpu_count() or 1))
    assert np.array_equal(out[pick], exp)
    # the 4-wave and the 8-wave workgroup shapes are the same arithmetic: first 1,000 gates alone == inside the big launch
    assert np.array_equal(engine.gate_batch(R.NAND, c0[:1000], c1[:1000]), out[:1000])


def test_config3_whole_batch_65536_on_one_gpu(engine, orc, params, keys):
    """BASELINE config 3 at its full size (65,536 gates), here on one GPU: every output decrypts; the eight contiguous ranges the
    node's GPUs would take (8,192 each) computed on their own give the same words as the one big call; a random
    sample of 1,024 outputs is bit-exact against the oracle.  Device-resident input / output (333 MB in, 167 MB out)."""
    import torch
    import rustfhe_amd as R
    from rustfhe_amd.shard import partition
    G = 65536
    rng = np.random.default_rng(65536)
    b0, b1 = rng.integers(0, 2, G), rng.integers(0, 2, G)
    c0, c1 = keys.encrypt_bits(b0), keys.encrypt_bits(b1)
    d0 = torch.from_numpy(c0.view(np.int32)).cuda()
    d1 = torch.from_numpy(c1.view(np.int32)).cuda()
    do = torch.empty_like(d0)
    st = torch.cuda.current_stream().cuda_stream
    engine.gate_batch_dev(R.NAND, d0, d1, do, G, st)
    engine.sync(st)
    out = do.cpu().numpy().view(np.uint32)
    assert keys.decrypt_bits(out) == list(1 - (b0 & b1))
    part = torch.empty_like(d0)
    for lo, hi in partition(G, 8):
        engine.gate_batch_dev(R.NAND, d0[lo:hi], d1[lo:hi], part[lo:hi], hi - lo, st)
    engine.sync(st)
    assert torch.equal(part, do)
    pick = rng.choice(G, 1024, replace=False)
    exp, _ = orc.gate_batch_mt(params, orc.NAND, keys.bk_f, None, keys.ksk, c0[pick], c1[pick], nthreads=min(64, os.cpu_count() or 1))
    assert np.array_equal(out[pick], exp)


def test_four_waves_per_gate_agree_with_two_at_n1024(engine, orc, params, keys, monkeypatch):
    """N = 1024, batches and tails of more than 256 and up to 768 gates (two or three gates per CU): the default dispatch gives a gate four waves --
    (polynomial, parity of the point index), k_bootstrap_pair4 -- where RTFHE_PAIR4=0 keeps k_bootstrap_pair's two: the same arithmetic, so identical words for whole gates,
    for a tail behind a full round, and for blind-rotation prefixes (the latter also against the oracle)."""
    import rustfhe_amd as R
    rng = np.random.default_rng(44)
    G = 1024 + 300
    b0, b1 = rng.integers(0, 2, G), rng.integers(0, 2, G)
    c0, c1 = keys.encrypt_bits(b0), keys.encrypt_bits(b1)
    monkeypatch.setenv("RTFHE_PAIR4", "0")
    two = R.Engine(R.Params(n=params.n, N=params.N, l=params.l, bgbit=params.bgbit, ks_t=params.ks_t, ks_basebit=params.ks_basebit), 0)
    monkeypatch.delenv("RTFHE_PAIR4")
    try:
        two.load_bk_torus(keys.bk_t)
        two.load_ksk(keys.ksk)
        ref = two.gate_batch(R.NAND, c0, c1)
        for k in (257, 300, 511, 512, 513, 700, 768, G):
            out = engine.gate_batch(R.NAND, c0[:k], c1[:k])
            assert np.array_equal(out, ref[:k]), k
        assert keys.decrypt_bits(engine.gate_batch(R.XOR, c0[:400], c1[:400])) == list(b0[:400] ^ b1[:400])
        t = np.stack([orc.gate_linear(params, orc.NAND, x, y) for x, y in zip(c0[:260], c1[:260])])
        pl = orc.Plan(params.N)
        for steps in (1, 3):
            got = engine.blind_rotate_batch(t, steps)
            assert np.array_equal(got, two.blind_rotate_batch(t, steps))
            assert np.array_equal(got[7].reshape(-1), orc.blind_rotate(params, pl, keys.bk_f, None, t[7], steps))
    finally:
        two.close()


@pytest.fixture(scope="module")
def setup2048(orc):
    import rustfhe_amd as R
    P = orc.Params(N=2048)
    K = orc.Keys(P, 2048)
    e = R.Engine(R.Params(N=2048), 0)
    e.load_bk_torus(K.bk_t)
    e.load_ksk(K.ksk)
    yield P, K, e
    e.close()


def test_config5_transforms_n2048(setup2048, orc):
    P, K, e = setup2048
    g = golden("fft_N2048.npz")
    a, b = e.twiddles()
    assert a.tobytes() == g["ifft_table"].tobytes() and b.tobytes() == g["fft_table"].tobytes()
    assert e.ifft_i32_batch(g["fft_src"]).tobytes() == g["fft_fwd"].tobytes()
    assert np.array_equal(e.fft_u32_batch(g["inv_src"]), g["inv_out"])
    assert e.export_bk_fft().tobytes() == K.bk_f.tobytes()


def test_config5_gates_n2048(setup2048, orc):
    import rustfhe_amd as R
    P, K, e = setup2048
    pl = orc.Plan(P.N)
    rng = np.random.default_rng(44)
    trlwe = rng.integers(0, 2 ** 32, (5, 2 * P.N), dtype=np.uint64).astype(np.uint32)
    idx = np.array([0, 1, 300, 634, 77], np.int32)
    w = P.trgsw_words
    exp = np.stack([orc.external_product(P, pl, K.bk_f[i * w:(i + 1) * w], None, t) for i, t in zip(idx, trlwe)])
    assert np.array_equal(e.external_product_batch(idx, trlwe).reshape(exp.shape), exp)
    b0, b1 = [0, 0, 1, 1, 1, 0], [0, 1, 0, 1, 1, 1]
    c0, c1 = K.encrypt_bits(b0), K.encrypt_bits(b1)
    t = np.stack([orc.gate_linear(P, orc.NAND, x, y) for x, y in zip(c0[:3], c1[:3])])
    acc = e.blind_rotate_batch(t, 5)
    assert np.array_equal(acc.reshape(3, -1), np.stack([orc.blind_rotate(P, pl, K.bk_f, None, x, 5) for x in t]))
    for op, tt in ((R.NAND, [1, 1, 1, 0, 0, 1]), (R.XOR, [0, 1, 1, 0, 0, 1])):
        out = e.gate_batch(op, c0, c1)
        assert K.decrypt_bits(out) == tt
        exp = np.stack([orc.gate(P, pl, op, K.bk_f, None, K.ksk, x, y) for x, y in zip(c0, c1)])
        assert np.array_equal(out, exp)
    # a 1,024-gate batch at N = 2048 (BASELINE config 5): all decrypt, all bit-exact against the oracle
    bb0, bb1 = rng.integers(0, 2, 1024), rng.integers(0, 2, 1024)
    d0, d1 = K.encrypt_bits(bb0), K.encrypt_bits(bb1)
    out = e.gate_batch(R.NAND, d0, d1)
    assert K.decrypt_bits(out) == list(1 - (bb0 & bb1))
    exp, _ = orc.gate_batch_mt(P, orc.NAND, K.bk_f, None, K.ksk, d0, d1, nthreads=min(64, os.cpu_count() or 1))
    assert np.array_equal(out, exp)          # every one of the 1,024 outputs, word for word


@pytest.mark.parametrize("n", [1, 60, 767])
def test_other_tlwe_dimensions(orc, n):
    """n is a runtime parameter of the engine (1 <= n <= 767); the reference exercises small dimensions in its
    key-switch test (hom_nand/src/tlwe.rs:346-396: 256 -> 60).  Bit-exact against the oracle for each."""
    import rustfhe_amd as R
    P = orc.Params(n=n)
    K = orc.Keys(P, 1000 + n)
    e = R.Engine(R.Params(n=n), 0)
    try:
        e.load_bk_torus(K.bk_t)
        e.load_ksk(K.ksk)
        pl = orc.Plan(P.N)
        b0, b1 = [0, 1, 1, 0, 1], [1, 1, 0, 0, 1]
        c0, c1 = K.encrypt_bits(b0), K.encrypt_bits(b1)
        for op in (R.NAND, R.OR):
            out = e.gate_batch(op, c0, c1)
            exp = np.stack([orc.gate(P, pl, op, K.bk_f, None, K.ksk, x, y) for x, y in zip(c0, c1)])
            assert np.array_equal(out, exp)
        if n >= 60:     # enough key entropy for the noise budget to hold
            assert K.decrypt_bits(e.gate_batch(R.NAND, c0, c1)) == [1, 0, 1, 1, 0]
    finally:
        e.close()


def test_degenerate_inputs(engine, orc, params, keys):
    """All-zero and all-ones words: every mod-switched rotation is 0 (no-op CMUX: cross(0) = 0 exactly) or 2N-1."""
    import rustfhe_amd as R
    pl = orc.Plan(params.N)
    z = np.zeros((1, params.n + 1), np.uint32)
    f = np.full((1, params.n + 1), 0xFFFFFFFF, np.uint32)
    half = np.full((1, params.n + 1), 0x80000000, np.uint32)
    for a, b, op in ((z, z, R.NAND), (z, z, R.COPY), (f, f, R.AND), (half, z, R.XOR), (f, z, R.NOT)):
        out = e_out = engine.gate_batch(op, a, b)
        exp = orc.gate(params, pl, op, keys.bk_f, None, keys.ksk, a[0], b[0])
        assert np.array_equal(e_out[0], exp)
    # a bootstrap of a pre-combined TLWE equals the COPY gate
    t = keys.encrypt_bits([1])
    assert np.array_equal(engine.bootstrap_batch(t), engine.gate_batch(R.COPY, t))


def test_config1_example_and_sharded_path_on_one_gpu(engine, params, keys):
    """BASELINE config 1: the homnand-bench example counterpart runs its truth tables; and the scatter/compute/gather
    helper of the multi-GPU path (rustfhe_amd/shard.py) with its real GPU callback at world size 1."""
    import contextlib, io, os, runpy
    import torch
    import rustfhe_amd as R
    from rustfhe_amd.shard import ShardedGates, engine_compute
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):      # in-process: a GPU-initialised process must not spawn GPU children here
        runpy.run_path(os.path.join(root, "examples", "homnand_bench.py"), run_name="__main__")
        import sys
        argv, sys.argv = sys.argv, ["nander_adder.py", "200", "57", "(1|0)&!(1^1)"]
        try:
            runpy.run_path(os.path.join(root, "examples", "nander_adder.py"), run_name="__main__")
        finally:
            sys.argv = argv
    text = buf.getvalue()
    assert text.count("200 + 57 = 257") == 3 and "(1|0)&!(1^1)" in text, text[-2000:]
    assert "all truth tables ok" in text, text[-2000:]
    assert text.count("micro-seconds") == 18                 # the reference example's 18 bootstraps
    rng = np.random.default_rng(91)
    b0, b1 = rng.integers(0, 2, 700), rng.integers(0, 2, 700)
    c0, c1 = keys.encrypt_bits(b0), keys.encrypt_bits(b1)
    sg = ShardedGates(engine_compute(engine), params.n + 1, torch.device("cuda", 0))
    res = sg.run(R.NAND, torch.from_numpy(c0.view(np.int32)), torch.from_numpy(c1.view(np.int32)), 700)
    got = res.cpu().numpy().view(np.uint32)
    assert np.array_equal(got, engine.gate_batch(R.NAND, c0, c1))
    assert keys.decrypt_bits(got) == list(1 - (b0 & b1))


def test_dispatch_boundaries_are_seamless(engine, params, keys):
    """Batch sizes around every launch-shape boundary (1, 2, 3, 4 gates per CU on 256 CUs: 256 | 512 | 768 | 1024 gates, alone and as the remainder of a larger batch) and odd sizes: gate g's output must
    not depend on the batch it travels in."""
    import rustfhe_amd as R
    rng = np.random.default_rng(123)
    G = 2051
    b0, b1 = rng.integers(0, 2, G), rng.integers(0, 2, G)
    c0, c1 = keys.encrypt_bits(b0), keys.encrypt_bits(b1)
    full = engine.gate_batch(R.NAND, c0, c1)
    assert keys.decrypt_bits(full) == list(1 - (b0 & b1))
    for k in (1, 2, 7, 63, 65, 255, 256, 257, 300, 511, 512, 513, 700, 767, 768, 769, 1023, 1024, 1025, 1281, 1537, 1793, 2047, 2049):
        part = engine.gate_batch(R.NAND, c0[:k], c1[:k])
        assert np.array_equal(part, full[:k]), k
    # a long run of launches leaves results unchanged (no state carried between calls)
    for _ in range(20):
        assert np.array_equal(engine.gate_batch(R.NAND, c0[:300], c1[:300]), full[:300])


def test_multi_device_context_with_one_device_equals_single(params, keys, gold_gate):
    """rtfhe_ctx_create_multi (the entry a Rust hom_nand_batch binds for a whole node): with n_dev = 1 every host-pointer
    batch call gives the words rtfhe_gate_batch gives; the golden gate and the MUX come out bit for bit."""
    import rustfhe_amd as R
    p = R.Params(n=params.n, N=params.N, l=params.l, bgbit=params.bgbit, ks_t=params.ks_t, ks_basebit=params.ks_basebit)
    m = R.Engine(p, devices=[0])
    assert m.device_count() == 1
    m.load_bk_torus(keys.bk_t)
    m.load_ksk(keys.ksk)
    ops, in0, in1 = gold_gate["ops"], gold_gate["in0"], gold_gate["in1"]
    for g in range(len(ops)):
        assert np.array_equal(m.gate_batch(int(ops[g]), in0[g:g + 1], in1[g:g + 1])[0], gold_gate["out"][g])
    assert np.array_equal(m.mux_batch(in0[2:3], in0[0:1], in1[1:2])[0], gold_gate["mux_out"])
    rng = np.random.default_rng(17)
    b0, b1 = rng.integers(0, 2, 300), rng.integers(0, 2, 300)
    c0, c1 = keys.encrypt_bits(b0), keys.encrypt_bits(b1)
    s = R.Engine(p, 0)
    s.load_bk_torus(keys.bk_t)
    s.load_ksk(keys.ksk)
    assert np.array_equal(m.gate_batch(R.NAND, c0, c1), s.gate_batch(R.NAND, c0, c1))
    with pytest.raises(R.RtfheError):
        R.Engine(p, devices=[0, 4096])
    m.close()
    s.close()


def test_pinned_and_pageable_host_buffers_agree(engine, params, keys):
    """Host-pointer calls DMA straight from rtfhe_host_alloc memory and stage any other pointer through the context's
    pinned buffers: same words either way."""
    import rustfhe_amd as R
    rng = np.random.default_rng(23)
    b0, b1 = rng.integers(0, 2, 200), rng.integers(0, 2, 200)
    c0, c1 = keys.encrypt_bits(b0), keys.encrypt_bits(b1)
    p0, p1 = R.pinned_empty(c0.shape), R.pinned_empty(c1.shape)
    p0[:], p1[:] = c0, c1
    a = engine.gate_batch(R.XOR, c0, c1)
    b = engine.gate_batch(R.XOR, p0, p1)
    assert np.array_equal(a, b) and keys.decrypt_bits(a) == list(b0 ^ b1)
    del p0, p1


def test_key_file_feeds_an_engine(tmp_path, params, keys, gold_gate):
    """The wire format on the path it exists for: a key file written by rtfhe_keys_write is read back, loaded into a
    fresh context and produces the golden gate outputs."""
    import rustfhe_amd as R
    p = R.Params(n=params.n, N=params.N, l=params.l, bgbit=params.bgbit, ks_t=params.ks_t, ks_basebit=params.ks_basebit)
    path = str(tmp_path / "server.rtfhe")
    R.save_keys(path, p, None, None, keys.bk_t, keys.ksk)            # public part only: what a server holds
    q, k0, k1, bk, ksk = R.load_keys(path)
    assert k0 is None and k1 is None and (q.n, q.N) == (p.n, p.N)
    e = R.Engine(q, 0)
    e.load_bk_torus(bk)
    e.load_ksk(ksk)
    g = 0
    out = e.gate_batch(int(gold_gate["ops"][g]), gold_gate["in0"][g:g + 1], gold_gate["in1"][g:g + 1])
    assert np.array_equal(out[0], gold_gate["out"][g])
    cts = str(tmp_path / "batch.rtfhe")
    R.save_tlwe(cts, q, out)
    assert np.array_equal(R.load_tlwe(cts), out)
    e.close()


def test_config5_both_kernel_shapes_agree(setup2048, orc, monkeypatch):
    """N = 2048: the default dispatch (transforms split over two waves by the parity of the point index: k_bootstrap_eo4, four waves per gate, for
    up to two gates per CU -- the ragged 37-gate batch, the blind-rotate prefixes and the 257-gate shape below run on it -- and k_bootstrap_eo, two
    waves per gate, beyond) against the one-wave-per-gate kernel (RTFHE_FORCE_WAVES=4, k_bootstrap<11>: a whole transform in one wave, no
    split at all) -- the same arithmetic: identical words for a ragged batch, for blind-rotate prefixes (both compared with the oracle above)
    and for every launch shape."""
    import rustfhe_amd as R
    P, K, e = setup2048
    rng = np.random.default_rng(2048)
    b0, b1 = rng.integers(0, 2, 37), rng.integers(0, 2, 37)
    c0, c1 = K.encrypt_bits(b0), K.encrypt_bits(b1)
    knob, val = "RTFHE_FORCE_WAVES", "4"
    monkeypatch.setenv(knob, val)
    one = R.Engine(R.Params(N=2048), 0)
    monkeypatch.delenv(knob)
    try:
        one.load_bk_torus(K.bk_t)
        one.load_ksk(K.ksk)
        for op in (R.NAND, R.OR, R.NOT):
            assert np.array_equal(e.gate_batch(op, c0, c1), one.gate_batch(op, c0, c1))
        t = np.stack([orc.gate_linear(P, orc.XOR, x, y) for x, y in zip(c0[:5], c1[:5])])
        for steps in (0, 1, 2, 9):
            assert np.array_equal(e.blind_rotate_batch(t, steps), one.blind_rotate_batch(t, steps))
        assert K.decrypt_bits(e.gate_batch(R.NAND, c0, c1)) == list(1 - (b0 & b1))
        # launch shapes by batch size (1, 2, 3 gates per workgroup below a full round): the same words as one wave per gate
        bb0, bb1 = rng.integers(0, 2, 1030), rng.integers(0, 2, 1030)
        d0, d1 = K.encrypt_bits(bb0), K.encrypt_bits(bb1)
        ref = one.gate_batch(R.NAND, d0, d1)
        for k in (1, 2, 257, 511, 513, 770, 1030):
            assert np.array_equal(e.gate_batch(R.NAND, d0[:k], d1[:k]), ref[:k]), k
    finally:
        one.close()


def test_bench_stdout_is_one_json_line_with_an_rccl_process_group():
    """bench.py's contract is ONE JSON line on stdout.  RCCL prints a version banner to stdout when its first communicator comes
    up: with RTFHE_BENCH_FORCE_PG=1 a single rank builds the RCCL process group (communicator, key broadcast, barriers,
    max-reduction run as in the N > 1 job) and stdout must still be exactly the line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RTFHE_BENCH_FORCE_PG="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout[:2000]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["outputs_decrypt_correctly"] and j["config"]["comm_backend"] == "nccl"
    assert 0 < j["roofline"]["frac"] < 1


def test_two_contexts_on_two_host_threads_and_context_churn(params, keys, gold_gate):
    """The ABI's threading rule is one context per host thread.  Two contexts driven from two threads at the same time (ctypes drops the
    GIL inside the call) give the golden words; creating and destroying contexts in a loop leaves the device's free memory where it
    was (keys, staging buffers, graphs and streams are all released)."""
    import threading
    import torch
    import rustfhe_amd as R
    free0 = None
    for round_ in range(6):
        e = R.Engine(R.Params(), 0)
        e.load_bk_torus(keys.bk_t)
        e.load_ksk(keys.ksk)
        rows = np.flatnonzero(np.array(gold_gate["ops"]) == R.NAND)[:3]
        out = e.gate_batch(R.NAND, gold_gate["in0"][rows], gold_gate["in1"][rows])
        assert np.array_equal(out, gold_gate["out"][rows])
        e.close()
        torch.cuda.synchronize()
        free = torch.cuda.mem_get_info()[0]
        if round_ == 1:
            free0 = free                      # after the first rounds the runtime's own pools have settled
        elif round_ > 1:
            assert free >= free0 - (8 << 20), (free0, free)
    engines = []
    for _ in range(2):
        e = R.Engine(R.Params(), 0)
        e.load_bk_torus(keys.bk_t)
        e.load_ksk(keys.ksk)
        engines.append(e)
    results, errors = [None, None], []
    nand = np.array(gold_gate["ops"]) == R.NAND                       # the fixture's rows carry their own opcodes
    g_in0, g_in1, g_out = np.array(gold_gate["in0"])[nand], np.array(gold_gate["in1"])[nand], np.array(gold_gate["out"])[nand]

    def work(k):
        try:
            for _ in range(5):
                results[k] = engines[k].gate_batch(R.NAND, g_in0, g_in1)
                assert np.array_equal(results[k], g_out[:len(g_in0)])
        except Exception as ex:               # noqa: BLE001 -- reported in the main thread
            errors.append(ex)
    threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for e in engines:
        e.close()
    assert not errors, errors


@pytest.mark.parametrize("n", [635, 60])
def test_batch_key_switch_on_the_matrix_pipe_equals_the_fused_kernel(orc, monkeypatch, n):
    """Plain batches of >= 1024 gates take the split path: blind rotate + sample extract (k_bootstrap_pair, MODE_EXTRACT), then the
    key switch of the whole batch as one exact i8 contraction (k_key_switch_mm: one-hot digits x signed byte limbs of the key,
    i32 accumulation, limbs recombined mod 2^32).  Same torus words as the fused kernel (RTFHE_KS_MM_MIN=0) for whole rounds, for
    rounds + a remainder and with the key-switch rows driven to their extremes, and bit-exact against the oracle."""
    import rustfhe_amd as R
    P = orc.Params(n=n)
    K = orc.Keys(P, 4242 + n)
    # extreme key rows: all-ones / 0x80.. / 0x7f.. words exercise the signed-limb carries
    ksk = K.ksk.copy().reshape(P.N, P.ks_t, 3, n + 1)
    ksk[0, 0, 0, :] = 0xFFFFFFFF; ksk[1, 1, 1, :] = 0x80808080; ksk[2, 2, 2, :] = 0x7F7F7F7F; ksk[3, 7, 0, :] = 0x80000000
    ksk[4, 0, 1, :] = 0xFFFFFF80; ksk[5, 3, 2, :] = 0x00000080
    ksk = ksk.reshape(-1)
    p = R.Params(n=n)
    monkeypatch.setenv("RTFHE_KS_MM_MIN", "0")
    fused = R.Engine(p, 0)
    monkeypatch.delenv("RTFHE_KS_MM_MIN")
    split = R.Engine(p, 0)
    try:
        for e in (fused, split):
            e.load_bk_torus(K.bk_t)
            e.load_ksk(ksk)
        rng = np.random.default_rng(n)
        G = 2048 + 300
        c0 = rng.integers(0, 2 ** 32, (G, n + 1), dtype=np.uint64).astype(np.uint32)      # arbitrary TLWE words: every digit pattern
        c1 = rng.integers(0, 2 ** 32, (G, n + 1), dtype=np.uint64).astype(np.uint32)
        for k in (1024, 2048, G):
            a, b = fused.gate_batch(R.NAND, c0[:k], c1[:k]), split.gate_batch(R.NAND, c0[:k], c1[:k])
            assert np.array_equal(a, b), k
        pl = orc.Plan(P.N)
        pick = rng.choice(1024, 6, replace=False)
        exp = np.stack([orc.gate(P, pl, orc.NAND, K.bk_f, None, ksk, c0[g], c1[g]) for g in pick])
        assert np.array_equal(b[pick], exp)
        # the exact-integer NTT backend takes the same split path behind its own blind rotation
        fused.set_backend(R._ffi.BACKEND_NTT_EXACT); split.set_backend(R._ffi.BACKEND_NTT_EXACT)
        assert np.array_equal(fused.gate_batch(R.XOR, c0[:1024], c1[:1024]), split.gate_batch(R.XOR, c0[:1024], c1[:1024]))
    finally:
        fused.close(); split.close()


def test_batch_key_switch_sample_tiles_at_every_ragged_count(orc, monkeypatch):
    """The lvl1 samples travel from the extract launch to k_key_switch_mm in tiles of 16 gates (ext_slot, rtfhe_kernels.hpp): counts below, at and
    just past a tile and a workgroup of the key switch (512 gates), a remainder behind whole rounds, and segments of one batch that start inside
    the sample buffer (ext_first) must all give the fused kernel's torus words."""
    import rustfhe_amd as R
    n = 60
    P = orc.Params(n=n)
    K = orc.Keys(P, 77)
    p = R.Params(n=n)
    monkeypatch.setenv("RTFHE_KS_MM_MIN", "0")
    fused = R.Engine(p, 0)
    monkeypatch.delenv("RTFHE_KS_MM_MIN")
    split = R.Engine(p, 0)
    try:
        for e in (fused, split):
            e.load_bk_torus(K.bk_t)
            e.load_ksk(K.ksk)
        rng = np.random.default_rng(5)
        G = 2048 + 1024 + 17
        c0 = rng.integers(0, 2 ** 32, (G, n + 1), dtype=np.uint64).astype(np.uint32)
        c1 = rng.integers(0, 2 ** 32, (G, n + 1), dtype=np.uint64).astype(np.uint32)
        for k in (1, 15, 16, 17, 100, 511, 512, 513, 1025, 1040, 1024 + 300, G):
            a, b = fused.gate_batch(R.OR, c0[:k], c1[:k]), split.gate_batch(R.OR, c0[:k], c1[:k])
            assert np.array_equal(a, b), k
    finally:
        fused.close(); split.close()


def test_batch_key_switch_split_path_n2048(monkeypatch):
    """N = 2048 (config 5): both backends, whole round + remainder, fused (RTFHE_KS_MM_MIN=0) against split, word for word."""
    import rustfhe_amd as R
    p = R.Params(N=2048, n=48)
    key0, key1, bk, ksk = R.keygen(p, 777)
    monkeypatch.setenv("RTFHE_KS_MM_MIN", "0")
    fused = R.Engine(p, 0)
    monkeypatch.delenv("RTFHE_KS_MM_MIN")
    split = R.Engine(p, 0)
    try:
        for e in (fused, split):
            e.load_bk_torus(bk)
            e.load_ksk(ksk)
        rng = np.random.default_rng(48)
        G = 1024 + 77
        c0 = rng.integers(0, 2 ** 32, (G, p.n + 1), dtype=np.uint64).astype(np.uint32)
        c1 = rng.integers(0, 2 ** 32, (G, p.n + 1), dtype=np.uint64).astype(np.uint32)
        for backend in (R._ffi.BACKEND_FFT64_MIRROR, R._ffi.BACKEND_NTT_EXACT):
            fused.set_backend(backend); split.set_backend(backend)
            assert np.array_equal(fused.gate_batch(R.NAND, c0, c1), split.gate_batch(R.NAND, c0, c1)), backend
        bits = rng.integers(0, 2, (2, 1024)).astype(np.uint8)
        fused.set_backend(R._ffi.BACKEND_FFT64_MIRROR); split.set_backend(R._ffi.BACKEND_FFT64_MIRROR)
        e0, e1 = R.encrypt_bits(p, key0, bits[0], 1), R.encrypt_bits(p, key0, bits[1], 2)
        assert list(R.decrypt_bits(p, key0, split.gate_batch(R.NAND, e0, e1))) == list(1 - (bits[0] & bits[1]))
    finally:
        fused.close(); split.close()


def test_batches_on_two_streams_of_one_context_do_not_share_scratch(engine, keys):
    """The split path keeps the lvl1 samples of a batch in a scratch buffer between its two launches: one buffer PER STREAM, so that
    batches enqueued on different streams of one context (which may overlap on the device) cannot overwrite each other's samples."""
    import torch
    import rustfhe_amd as R
    rng = np.random.default_rng(99)
    G = 700
    bits = rng.integers(0, 2, (4, G))
    c = [torch.from_numpy(keys.encrypt_bits(b).view(np.int32)).cuda() for b in bits]
    ref0 = engine.gate_batch(R.NAND, c[0].cpu().numpy().view(np.uint32), c[1].cpu().numpy().view(np.uint32))
    ref1 = engine.gate_batch(R.XOR, c[2].cpu().numpy().view(np.uint32), c[3].cpu().numpy().view(np.uint32))
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    o0, o1 = torch.empty_like(c[0]), torch.empty_like(c[0])
    torch.cuda.synchronize()
    for _ in range(3):                              # several rounds: the two streams' launches interleave differently each time
        engine.gate_batch_dev(R.NAND, c[0], c[1], o0, G, s0.cuda_stream)
        engine.gate_batch_dev(R.XOR, c[2], c[3], o1, G, s1.cuda_stream)
        engine.gate_batch_dev(R.NAND, c[0], c[1], o0, G, s0.cuda_stream)
        engine.sync(s0.cuda_stream); engine.sync(s1.cuda_stream)
        assert np.array_equal(o0.cpu().numpy().view(np.uint32), ref0)
        assert np.array_equal(o1.cpu().numpy().view(np.uint32), ref1)


def test_a_callers_own_stream_capture_survives_a_later_larger_batch_on_that_stream(engine, keys):
    """A batch enqueued inside a CALLER'S stream capture must not bake the stream's split-path scratch pointer into the caller's graph
    (round-3 advisor finding): a later, larger eager batch on that stream frees and reallocates the buffer, and a replay would then write
    freed memory.  Inside a foreign capture the library therefore launches the fused kernel (one node, no scratch).  Capture a small batch,
    grow the stream's scratch with a much larger eager batch, replay the graph twice: same words as the eager path."""
    import torch
    import rustfhe_amd as R
    rng = np.random.default_rng(4242)
    small, large = 300, 3000
    bits = rng.integers(0, 2, (2, large))
    c0 = torch.from_numpy(keys.encrypt_bits(bits[0]).view(np.int32)).cuda()
    c1 = torch.from_numpy(keys.encrypt_bits(bits[1]).view(np.int32)).cuda()
    ref = engine.gate_batch(R.NAND, c0[:small].cpu().numpy().view(np.uint32), c1[:small].cpu().numpy().view(np.uint32))
    s = torch.cuda.Stream()
    out_small, out_large = torch.zeros_like(c0[:small]), torch.empty_like(c0)
    with torch.cuda.stream(s):
        engine.gate_batch_dev(R.NAND, c0, c1, out_small, small, s.cuda_stream)         # the stream now owns a scratch buffer for >= 1024 gates
        engine.sync(s.cuda_stream)
        assert np.array_equal(out_small.cpu().numpy().view(np.uint32), ref)
        out_small.zero_()
        g = torch.cuda.CUDAGraph()
        engine.timer_begin(s.cuda_stream)                                              # resets the context's launch counter (outside the capture)
        with torch.cuda.graph(g, stream=s):
            engine.gate_batch_dev(R.NAND, c0, c1, out_small, small, s.cuda_stream)
        _, launches_per_capture = engine.timer_end(s.cuda_stream)                      # kernel launches the call enqueued: the split path would make two
        engine.gate_batch_dev(R.NAND, c0, c1, out_large, large, s.cuda_stream)         # grows (frees + reallocates) the stream's scratch
        engine.sync(s.cuda_stream)
        for _ in range(2):
            out_small.zero_()
            g.replay()
            torch.cuda.synchronize()
            assert np.array_equal(out_small.cpu().numpy().view(np.uint32), ref)
    assert list(R.decrypt_bits(engine.p, keys.key0, out_large.cpu().numpy().view(np.uint32))) == list(1 - (bits[0] & bits[1]))
    assert launches_per_capture == 1, "a batch inside a caller's capture must be ONE fused kernel node (no scratch buffer in the caller's graph)"


def test_second_key_layouts_are_built_by_the_first_batch_that_reads_them(params, keys):
    """rtfhe_ctx_memory_bytes: loading the keys allocates the canonical spectra (+ the torus form, both key-switch forms, scratch); the layout of
    the four-wave kernel (N = 1024, tails of 257-768 gates) appears with the first batch whose dispatch reads it -- not at load time, and not for a
    batch that runs on the two-wave or the latency kernel -- and is rebuilt (same footprint, same words) after the key is loaded again."""
    import torch
    import rustfhe_amd as R
    p = R.Params()
    e = R.Engine(p, 0)
    try:
        spectra = p.n * 2 * 2 * p.l * (p.N // 2) * 16
        assert e.memory_bytes() == 0
        e.load_bk_torus(keys.bk_t)
        e.load_ksk(keys.ksk)
        loaded = e.memory_bytes()
        assert loaded >= spectra + p.bk_words * 4 + 156306144 + 83886080
        rng = np.random.default_rng(99)
        b0, b1 = rng.integers(0, 2, 1024), rng.integers(0, 2, 1024)
        d0 = torch.from_numpy(keys.encrypt_bits(b0).view(np.int32)).cuda()
        d1 = torch.from_numpy(keys.encrypt_bits(b1).view(np.int32)).cuda()
        out = torch.empty_like(d0)
        e.gate_batch_dev(R.NAND, d0, d1, out, 1024); e.sync()       # a full round of the two-wave kernel (and this stream's sample buffer)
        assert loaded < e.memory_bytes() < loaded + spectra
        loaded = e.memory_bytes()
        e.gate_batch_dev(R.NAND, d0, d1, out, 100); e.sync()        # a tail on the latency kernel: still the canonical layout only
        assert e.memory_bytes() == loaded
        e.gate_batch_dev(R.NAND, d0, d1, out, 300); e.sync()          # 257-768 gates: four waves per gate, its own layout of the spectra
        assert e.memory_bytes() == loaded + spectra
        first = out[:300].clone()
        e.load_bk_torus(keys.bk_t)                                     # a new key invalidates the layout; the next such batch rebuilds it in place
        e.gate_batch_dev(R.NAND, d0, d1, out, 300); e.sync()
        assert e.memory_bytes() == loaded + spectra and torch.equal(out[:300], first)
        assert keys.decrypt_bits(first.cpu().numpy().view(np.uint32)) == list(1 - (b0[:300] & b1[:300]))
    finally:
        e.close()


def test_mux_batches_on_two_streams_and_inside_a_capture(params, keys):
    """Advisor r5: the MUX intermediates belong to the stream the batch runs on.  Two MUX batches enqueued on two streams without a host
    synchronisation between them give the single-stream words; inside a caller's capture the call is refused until the stream's buffers exist
    (nothing may be allocated there), then captured, and the captured buffers survive a later, larger eager batch on the same stream."""
    import torch
    import rustfhe_amd as R
    p = R.Params()
    e = R.Engine(p, 0)
    try:
        e.load_bk_torus(keys.bk_t)
        e.load_ksk(keys.ksk)
        k = 700
        rng = np.random.default_rng(321)
        bits = rng.integers(0, 2, (6, k))
        d = [torch.from_numpy(keys.encrypt_bits(b).view(np.int32)).cuda() for b in bits]
        ref_a, ref_b = torch.empty_like(d[0]), torch.empty_like(d[0])
        e.mux_batch_dev(d[0], d[1], d[2], ref_a, k); e.sync()
        e.mux_batch_dev(d[3], d[4], d[5], ref_b, k); e.sync()
        assert keys.decrypt_bits(ref_a.cpu().numpy().view(np.uint32)) == list(np.where(bits[0], bits[2], bits[1]))
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        torch.cuda.synchronize()
        out_a, out_b = torch.zeros_like(d[0]), torch.zeros_like(d[0])
        for _ in range(3):                                   # overlapping launches: the streams run side by side on the card
            e.mux_batch_dev(d[0], d[1], d[2], out_a, k, s1.cuda_stream)
            e.mux_batch_dev(d[3], d[4], d[5], out_b, k, s2.cuda_stream)
        torch.cuda.synchronize()
        assert torch.equal(out_a, ref_a) and torch.equal(out_b, ref_b)
        # capture: refused on a stream that has no buffers yet, accepted once it has
        s3 = torch.cuda.Stream()
        cap_out = torch.zeros_like(d[0])
        g = torch.cuda.CUDAGraph()
        with pytest.raises(R.RtfheError) as err:
            with torch.cuda.graph(g, stream=s3):
                e.mux_batch_dev(d[0], d[1], d[2], cap_out, 16, torch.cuda.current_stream().cuda_stream)
        assert err.value.code == R._ffi.ERR_STATE
        torch.cuda.synchronize()
        e.mux_batch_dev(d[0], d[1], d[2], cap_out, 16, s3.cuda_stream); torch.cuda.synchronize()      # eager, sizes this stream's buffers
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s3):
            e.mux_batch_dev(d[0], d[1], d[2], cap_out, 16, torch.cuda.current_stream().cuda_stream)
        cap_out.zero_(); g.replay(); torch.cuda.synchronize()
        assert torch.equal(cap_out[:16], ref_a[:16])
        before = e.memory_bytes()
        e.mux_batch_dev(d[0], d[1], d[2], out_a, k, s3.cuda_stream); torch.cuda.synchronize()          # larger: new buffers, the captured ones stay
        assert torch.equal(out_a, ref_a) and e.memory_bytes() > before
        cap_out.zero_(); g.replay(); torch.cuda.synchronize()
        assert torch.equal(cap_out[:16], ref_a[:16])
    finally:
        e.close()
