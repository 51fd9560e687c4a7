#!/usr/bin/env python3
"""Times the 8-bit adder netlists (BASELINE config 4: ripple-carry, NAND-only and XOR/AND/OR; and the parallel-prefix netlist of the
same function) for several replica counts."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rustfhe_amd as R
from rustfhe_amd.circuit import CircuitRunner, prefix_adder, ripple_carry_adder

P = R.Params()
key0, key1, bk, ksk = R.keygen(P, 20211003)
eng = R.Engine(P, 0)
eng.load_bk_torus(bk); eng.load_ksk(ksk)
rng = np.random.default_rng(0)
for kind in ("nand-only", "xor/and/or", "prefix nand-only", "prefix xor/and/or"):
    net = prefix_adder(8, kind.endswith("nand-only")) if kind.startswith("prefix") else ripple_carry_adder(8, kind == "nand-only")
    d = net.describe()
    for reps in (1, 32, 256, 1024):
        A, B = rng.integers(0, 256, reps), rng.integers(0, 256, reps)
        bits = np.array([[(a >> i) & 1 for i in range(8)] + [(b >> i) & 1 for i in range(8)] for a, b in zip(A, B)], np.uint8)
        cts = R.encrypt_bits(P, key0, bits.reshape(-1), 7).reshape(reps, 16, P.n + 1)
        run = CircuitRunner(eng, net, reps)
        run.set_inputs(cts)
        run.run(); torch.cuda.synchronize()
        t0 = time.perf_counter(); run.run(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        dec = R.decrypt_bits(P, key0, run.outputs().reshape(-1, P.n + 1)).reshape(reps, 9)
        ok = bool(np.array_equal((dec * (1 << np.arange(9))).sum(axis=1), A + B))
        print(json.dumps({"adder": kind, "gates": d["gates"], "depth": d["depth"], "replicas": reps,
                          "ms_per_addition_batch": round(dt * 1e3, 2), "additions_per_s": round(reps / dt, 1),
                          "gates_per_s": round(reps * d["gates"] / dt, 1), "ok": ok}), flush=True)
