#!/usr/bin/env python3
"""The nander front-end on the GPU (BASELINE config 4): evaluates logic expressions in the reference's grammar
(nander/src/lib.rs:90-172: & | ^ $ = NAND, ! prefix, parentheses, left-associative, constants 0 / 1) and adds two encrypted
8-bit numbers with three netlists of the same function -- the NAND-only ripple-carry adder, its parallel-prefix form in NAND
gates, and the parallel-prefix form in AND / OR / XOR gates -- each as one HIP-graph submission.

    python examples/nander_adder.py 200 57 "(1|0)&!(1^1)"
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rustfhe_amd as R
from rustfhe_amd.circuit import CircuitRunner, eval_logic_expr, prefix_adder, ripple_carry_adder

a = int(sys.argv[1]) if len(sys.argv) > 1 else 200
b = int(sys.argv[2]) if len(sys.argv) > 2 else 57
exprs = sys.argv[3:] or ["1$1", "(1|0)&!(1^1)"]
assert 0 <= a < 256 and 0 <= b < 256

P = R.Params()
key0, key1, bk, ksk = R.keygen(P)              # secret keys from the OS CSPRNG, like the reference's thread_rng
eng = R.Engine(P, 0)
eng.load_bk_torus(bk)
eng.load_ksk(ksk)

for text in exprs:
    ct = eval_logic_expr(eng, text)            # constants are trivial ciphertexts, every operator one bootstrapped gate
    print("%-24s = %d" % (text, int(R.decrypt_bits(P, key0, ct[None])[0])))

bits = np.array([(a >> i) & 1 for i in range(8)] + [(b >> i) & 1 for i in range(8)], np.uint8)
cts = R.encrypt_bits(P, key0, bits).reshape(1, 16, P.n + 1)
for name, net in (("ripple-carry, NAND only", ripple_carry_adder(8, True)),
                  ("parallel prefix, NAND only", prefix_adder(8, True)),
                  ("parallel prefix, AND/OR/XOR", prefix_adder(8, False))):
    run = CircuitRunner(eng, net, 1)
    run.set_inputs(cts)
    run.run()                                  # records the levelised netlist into a HIP graph and runs it once
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run.run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    out = R.decrypt_bits(P, key0, run.outputs().reshape(-1, P.n + 1))
    total = int(sum(int(v) << i for i, v in enumerate(out)))
    d = net.describe()
    print("%-28s %3d gates in %2d levels: %d + %d = %d  (%.1f ms)" % (name, d["gates"], d["depth"], a, b, total, ms))
    assert total == a + b
    run.close()
eng.close()
