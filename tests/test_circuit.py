"""Gate-netlist front-end (BASELINE config 4).  CPU: grammar, levelisation and adder netlists on plain bits.
GPU: the 8-bit ripple-carry adder as a NAND netlist run wave by wave, many replicas at once."""
import numpy as np
import pytest


def test_parser_matches_reference_grammar():
    from rustfhe_amd.circuit import parse_logic_expr
    # nander/src/lib.rs:90-172: left-assoc, no precedence, '!' prefix, parentheses
    cases = {"1": 1, "0": 0, "1&0": 0, "1|0": 1, "1^1": 0, "1$1": 0, "0$1": 1, "!0": 1, "!!1": 1,
             "1&0|1": 1, "1|0&0": 0, "(1|0)&(1^0)": 1, "!(1&1)$1": 1, " 1 & ( 0 | ! 0 ) ": 1, "1^1^1": 1}
    for text, want in cases.items():
        net, _ = parse_logic_expr(text)
        assert net.evaluate_plain([]) == [want], text
    for bad, msg in {"": "this is none", "(1&0": "braket is not closed", "2": "invalid element", "1&": "this is none"}.items():
        with pytest.raises(ValueError) as e:
            parse_logic_expr(bad)
        assert msg in str(e.value)


def test_adder_netlists_on_plain_bits():
    from rustfhe_amd.circuit import ripple_carry_adder
    rng = np.random.default_rng(1)
    for nand_only, gates_per_fa in ((True, 9), (False, 5)):
        net = ripple_carry_adder(8, nand_only)
        d = net.describe()
        assert d["gates"] == 8 * gates_per_fa and d["inputs"] == 16 and sum(d["wave_sizes"]) == d["gates"]
        for _ in range(200):
            a, b = int(rng.integers(0, 256)), int(rng.integers(0, 256))
            bits = [(a >> i) & 1 for i in range(8)] + [(b >> i) & 1 for i in range(8)]
            out = net.evaluate_plain(bits)
            assert sum(v << i for i, v in enumerate(out)) == a + b
    # levelisation respects dependencies
    net = ripple_carry_adder(4, True)
    level = {}
    base = 2 + net.num_inputs
    for li, wave in enumerate(net.levels()):
        for g in wave:
            level[base + g] = li + 1
    for g, (op, a, b) in enumerate(net.gates):
        assert level.get(a, 0) < level[base + g] and level.get(b, 0) < level[base + g]


def test_prefix_adder_netlist_on_plain_bits():
    from rustfhe_amd.circuit import prefix_adder, ripple_carry_adder
    rng = np.random.default_rng(2)
    for nbits in (1, 2, 3, 4, 8, 13):
        net = prefix_adder(nbits)
        for _ in range(200):
            a, b = int(rng.integers(0, 1 << nbits)), int(rng.integers(0, 1 << nbits))
            bits = [(a >> i) & 1 for i in range(nbits)] + [(b >> i) & 1 for i in range(nbits)]
            out = net.evaluate_plain(bits)
            assert sum(v << i for i, v in enumerate(out)) == a + b, (nbits, a, b)
    d = prefix_adder(8).describe()
    assert d["depth"] == 7 and d["gates"] == 70 and ripple_carry_adder(8, False).describe()["depth"] == 17
    # NAND gates only (the reference's nander evaluates NAND trees): same function, depth 11 against the ripple-carry netlist's 20
    from rustfhe_amd.circuit import NAND
    for nbits in (1, 2, 3, 4, 8, 13):
        net = prefix_adder(nbits, nand_only=True)
        assert {op for op, _, _ in net.gates} == {NAND}
        for _ in range(200):
            a, b = int(rng.integers(0, 1 << nbits)), int(rng.integers(0, 1 << nbits))
            bits = [(a >> i) & 1 for i in range(nbits)] + [(b >> i) & 1 for i in range(nbits)]
            assert sum(v << i for i, v in enumerate(net.evaluate_plain(bits))) == a + b, (nbits, a, b)
    d = prefix_adder(8, nand_only=True).describe()
    assert d["depth"] == 11 and d["gates"] == 162 and ripple_carry_adder(8, True).describe()["depth"] == 20


def test_mux_netlist_plain():
    from rustfhe_amd.circuit import Netlist
    net = Netlist()
    c, x, y = net.inputs(3)
    net.output(net.mux(c, x, y))
    for bits in range(8):
        cb, xb, yb = bits & 1, (bits >> 1) & 1, (bits >> 2) & 1
        assert net.evaluate_plain([cb, xb, yb]) == [yb if cb else xb]
    assert net.describe()["depth"] == 2


@pytest.mark.gpu
def test_repl_expressions_on_gpu(engine, keys):
    from rustfhe_amd.circuit import eval_logic_expr
    for text, want in {"1$1": 0, "(1|0)&!(1^1)": 1, "!(0$0)": 0, "1": 1}.items():
        ct = eval_logic_expr(engine, text)
        assert keys.decrypt_bits([ct]) == [want], text


@pytest.mark.gpu
def test_ripple_carry_adder_8bit_waves(engine, orc, params, keys):
    import rustfhe_amd as R
    from rustfhe_amd.circuit import CircuitRunner, ripple_carry_adder
    net = ripple_carry_adder(8, nand_only=True)
    reps = 32
    rng = np.random.default_rng(8)
    A, B = rng.integers(0, 256, reps), rng.integers(0, 256, reps)
    bits = np.array([[(a >> i) & 1 for i in range(8)] + [(b >> i) & 1 for i in range(8)] for a, b in zip(A, B)])
    cts = keys.encrypt_bits(bits.reshape(-1)).reshape(reps, 16, params.n + 1)
    run = CircuitRunner(engine, net, reps)
    run.set_inputs(cts)
    out = run.run().outputs()
    dec = np.array(keys.decrypt_bits(out.reshape(-1, params.n + 1))).reshape(reps, 9)
    got = (dec * (1 << np.arange(9))).sum(axis=1)
    assert np.array_equal(got, A + B)
    # replica 0, gate by gate, bit-exact against the oracle walking the same netlist
    pl = orc.Plan(params.N)
    w = [None, None] + list(cts[0])
    triv = np.zeros((2, params.n + 1), np.uint32)
    triv[0, -1], triv[1, -1] = 0xE0000000, 0x20000000
    w[0], w[1] = triv[0], triv[1]
    for op, a, b in net.gates:
        w.append(orc.gate(params, pl, op, keys.bk_f, None, keys.ksk, w[a], w[b]))
    for k, wi in enumerate(net.outputs):
        assert np.array_equal(out[0, k], w[wi])
    # mixed-gate adder (XOR/AND/OR bootstraps): same sums, fewer waves
    net2 = ripple_carry_adder(8, nand_only=False)
    run2 = CircuitRunner(engine, net2, reps)
    run2.set_inputs(cts)
    dec2 = np.array(keys.decrypt_bits(run2.run().outputs().reshape(-1, params.n + 1))).reshape(reps, 9)
    assert np.array_equal((dec2 * (1 << np.arange(9))).sum(axis=1), A + B)
    assert net2.describe()["depth"] < net.describe()["depth"]


@pytest.mark.gpu
def test_mux_via_netlist_matches_mux_batch(engine, keys, gold_gate):
    from rustfhe_amd.circuit import CircuitRunner, Netlist
    net = Netlist()
    c, x, y = net.inputs(3)
    net.output(net.mux(c, x, y))
    run = CircuitRunner(engine, net, 1)
    run.set_inputs(np.stack([gold_gate["in0"][2], gold_gate["in0"][0], gold_gate["in1"][1]])[None])
    assert np.array_equal(run.run().outputs()[0, 0], gold_gate["mux_out"])


@pytest.mark.gpu
def test_netlist_wave_rejects_bad_wire_indices(engine, keys):
    """Wire indices / opcodes of a netlist wave are checked on the device: the offending gate is skipped, the other gates
    of the wave run, the next sync reports RTFHE_ERR_INVALID once, and the context stays usable."""
    import torch
    import rustfhe_amd as R
    n1 = engine.p.n + 1
    ct = keys.encrypt_bits([1, 0, 1])
    W = 8
    wires = torch.zeros((W, n1), dtype=torch.int32, device="cuda")
    wires[:3] = torch.from_numpy(ct.view(np.int32)).cuda()
    dev = lambda v: torch.tensor(v, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    #            ok: w3 = NAND(w0, w1)   bad input index      bad output index   bad opcode
    ops, i0, i1, io = dev([R.NAND, R.NAND, R.NAND, 99]), dev([0, 12345678, 0, 0]), dev([1, 1, 1, 1]), dev([3, 4, -1, 5])
    engine.circuit_wave_dev(ops, i0, i1, io, wires, W, 4, st)
    with pytest.raises(R.RtfheError) as ei:
        engine.sync(st)
    assert ei.value.code == R._ffi.ERR_INVALID
    out = wires.cpu().numpy().view(np.uint32)
    assert keys.decrypt_bits(out[3:4]) == [1]                       # NAND(1, 0)
    assert not out[4].any() and not out[5].any()                    # skipped gates wrote nothing
    engine.sync(st)                                                 # reported once
    engine.circuit_wave_dev(dev([R.AND]), dev([0]), dev([2]), dev([6]), wires, W, 1, st)
    engine.sync(st)
    assert keys.decrypt_bits(wires[6:7].cpu().numpy().view(np.uint32)) == [1]


@pytest.mark.gpu
def test_circuit_graph_replay_equals_wave_by_wave(engine, params, keys):
    """The netlist recorded once into a HIP graph (rtfhe_circuit_create) and replayed as one submission gives the words
    the wave-by-wave launches give, and can be replayed on new inputs."""
    from rustfhe_amd.circuit import CircuitRunner, ripple_carry_adder
    net = ripple_carry_adder(4, nand_only=True)
    reps = 3
    rng = np.random.default_rng(5)
    for trial in range(2):
        bits = rng.integers(0, 2, (reps, 8))
        cts = keys.encrypt_bits(bits.reshape(-1)).reshape(reps, 8, params.n + 1)
        if trial == 0:
            g, w = CircuitRunner(engine, net, reps), CircuitRunner(engine, net, reps)
        g.set_inputs(cts)
        w.set_inputs(cts)
        a = g.run(graph=True).outputs()
        b = w.run(graph=False).outputs()
        assert np.array_equal(a, b)
        dec = np.array(keys.decrypt_bits(a.reshape(-1, params.n + 1))).reshape(reps, 5)
        A = (bits[:, :4] * (1 << np.arange(4))).sum(axis=1)
        B = (bits[:, 4:] * (1 << np.arange(4))).sum(axis=1)
        assert np.array_equal((dec * (1 << np.arange(5))).sum(axis=1), A + B)
    g.close()
    w.close()


@pytest.mark.gpu
def test_circuit_runner_reports_a_skipped_gate_at_run(engine, keys):
    """A netlist with a wire index out of range is caught on the device; CircuitRunner.run() surfaces it itself (it used to
    leave the sticky fault for some later, unrelated sync)."""
    import rustfhe_amd as R
    from rustfhe_amd.circuit import CircuitRunner, Netlist
    for graph in (False, True):
        net = Netlist()
        a, b = net.inputs(2)
        net.output(net.nand(a, b))
        run = CircuitRunner(engine, net, 1)
        run.set_inputs(keys.encrypt_bits([1, 1])[None])
        ops, i0, i1, io, cnt = run.waves[0]
        i0[0] = 1 << 20                                   # corrupt the recorded wave: far outside the wire table
        with pytest.raises(R.RtfheError) as ei:
            run.run(graph=graph)
        assert ei.value.code == R._ffi.ERR_INVALID
        run.close()
    engine.sync()                                         # reported once: the context is clean again


@pytest.mark.gpu
def test_prefix_adder_8bit_on_gpu(engine, orc, params, keys):
    """The parallel-prefix adder of the reference's AND / OR / XOR gates: correct sums, replica 0 bit-exact gate by gate against the
    oracle, same words through the HIP-graph submission."""
    from rustfhe_amd.circuit import CircuitRunner, prefix_adder
    net = prefix_adder(8)
    reps = 16
    rng = np.random.default_rng(80)
    A, B = rng.integers(0, 256, reps), rng.integers(0, 256, reps)
    A[0], B[0], A[1], B[1] = 255, 255, 255, 1
    bits = np.array([[(a >> i) & 1 for i in range(8)] + [(b >> i) & 1 for i in range(8)] for a, b in zip(A, B)])
    cts = keys.encrypt_bits(bits.reshape(-1)).reshape(reps, 16, params.n + 1)
    run = CircuitRunner(engine, net, reps)
    run.set_inputs(cts)
    out = run.run(graph=True).outputs()
    dec = np.array(keys.decrypt_bits(out.reshape(-1, params.n + 1))).reshape(reps, 9)
    assert np.array_equal((dec * (1 << np.arange(9))).sum(axis=1), A + B)
    plain = CircuitRunner(engine, net, reps)
    plain.set_inputs(cts)
    assert np.array_equal(plain.run(graph=False).outputs(), out)
    pl = orc.Plan(params.N)
    w = [None, None] + list(cts[0])
    triv = np.zeros((2, params.n + 1), np.uint32)
    triv[0, -1], triv[1, -1] = 0xE0000000, 0x20000000
    w[0], w[1] = triv[0], triv[1]
    for op, x, y in net.gates:
        w.append(orc.gate(params, pl, op, keys.bk_f, None, keys.ksk, w[x], w[y]))
    for k, wi in enumerate(net.outputs):
        assert np.array_equal(out[0, k], w[wi])
    run.close()
    plain.close()
    # the NAND-only form: correct sums through one graph submission
    nn = CircuitRunner(engine, prefix_adder(8, nand_only=True), reps)
    nn.set_inputs(cts)
    dec = np.array(keys.decrypt_bits(nn.run(graph=True).outputs().reshape(-1, params.n + 1))).reshape(reps, 9)
    assert np.array_equal((dec * (1 << np.arange(9))).sum(axis=1), A + B)
    nn.close()


@pytest.mark.gpu
def test_a_recorded_circuit_follows_a_key_change(params, keys):
    """Advisor r5: a circuit's graph bakes the addresses of the key forms its kernels read.  A DIFFERENT key loaded afterwards must be what a replay
    computes with -- every form that exists is rebuilt in place by the load -- and where that is impossible (an exact backend's form after a key
    that has no torus form) the replay must fail instead of using the old key.  One wave of 300 gates takes the four-waves-per-gate kernel,
    i.e. the SECOND layout of the spectra; the same wave is recorded on the split-FFT backend as well."""
    import torch
    import rustfhe_amd as R
    p = R.Params()
    e = R.Engine(p, 0)
    try:
        e.load_bk_torus(keys.bk_t)
        e.load_ksk(keys.ksk)
        G, n1 = 300, p.n + 1
        rng = np.random.default_rng(44)
        b0, b1 = rng.integers(0, 2, G).astype(np.uint8), rng.integers(0, 2, G).astype(np.uint8)
        wires = torch.zeros((3 * G, n1), dtype=torch.int32, device="cuda")
        i32 = lambda v: torch.tensor(np.asarray(v, np.int32), dtype=torch.int32, device="cuda")
        ops, i0, i1, io = i32([R.NAND] * G), i32(np.arange(G)), i32(G + np.arange(G)), i32(2 * G + np.arange(G))
        offs = np.array([0, G], np.int32)

        def put_inputs(key0, s0, s1):
            wires[:G] = torch.from_numpy(R.encrypt_bits(p, key0, b0, s0).view(np.int32)).cuda()
            wires[G:2 * G] = torch.from_numpy(R.encrypt_bits(p, key0, b1, s1).view(np.int32)).cuda()
            wires[2 * G:] = 0

        def outputs(key0):
            return list(R.decrypt_bits(p, key0, wires[2 * G:].cpu().numpy().view(np.uint32)))

        put_inputs(keys.key0, 1, 2)
        c_mirror = e.circuit_create(ops, i0, i1, io, offs, wires, 3 * G)
        e.set_backend(R._ffi.BACKEND_FFT_SPLIT_EXACT)
        c_exact = e.circuit_create(ops, i0, i1, io, offs, wires, 3 * G)
        e.set_backend(R._ffi.BACKEND_FFT64_MIRROR)
        for c in (c_mirror, c_exact):
            wires[2 * G:] = 0
            e.circuit_launch(c); e.sync()
            assert outputs(keys.key0) == list(1 - (b0 & b1))
        # a different key set, loaded over the first
        k0b, k1b, bkb, kskb = R.keygen(p, 424242)
        assert not np.array_equal(k0b, keys.key0)
        e.load_bk_torus(bkb)
        e.load_ksk(kskb)
        put_inputs(k0b, 3, 4)
        eager = e.gate_batch(R.NAND, wires[:G].cpu().numpy().view(np.uint32), wires[G:2 * G].cpu().numpy().view(np.uint32))
        for c in (c_mirror, c_exact):
            wires[2 * G:] = 0
            e.circuit_launch(c); e.sync()
            assert outputs(k0b) == list(1 - (b0 & b1)), "the replay computed with the old key"
        e.circuit_launch(c_mirror); e.sync()
        assert np.array_equal(wires[2 * G:].cpu().numpy().view(np.uint32), eager)
        # the same key as spectra: no torus form any more -- the mirror circuit follows, the exact one refuses
        spectra = e.export_bk_fft()
        e.load_bk_fft(spectra)
        wires[2 * G:] = 0
        e.circuit_launch(c_mirror); e.sync()
        assert np.array_equal(wires[2 * G:].cpu().numpy().view(np.uint32), eager)
        with pytest.raises(R.RtfheError) as err:
            e.circuit_launch(c_exact)
        assert err.value.code == R._ffi.ERR_STATE
        # a torus-form key again: a NEW exact circuit works (the old one stays refused)
        e.load_bk_torus(bkb)
        e.set_backend(R._ffi.BACKEND_FFT_SPLIT_EXACT)
        c_exact2 = e.circuit_create(ops, i0, i1, io, offs, wires, 3 * G)
        wires[2 * G:] = 0
        e.circuit_launch(c_exact2); e.sync()
        assert outputs(k0b) == list(1 - (b0 & b1))
        for c in (c_mirror, c_exact, c_exact2):
            e.circuit_destroy(c)
    finally:
        e.close()
