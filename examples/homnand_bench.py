#!/usr/bin/env python3
"""Python counterpart of the reference's hom_nand/examples/homnand-bench.rs:7-137 (BASELINE config 1): key generation, then
for NAND/AND/OR/XOR (4 input pairs each) and NOT (2) encrypt, time the gate, decrypt, assert the truth table.
Prints the reference's line format: "<gate> <in0> <in1>: <N> micro-seconds"."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustfhe_amd as R

P = R.Params()
key0, key1, bk, ksk = R.keygen(P, int(time.time()))
tfhe = R.Engine(P, 0)
tfhe.load_bk_torus(bk)
tfhe.load_ksk(ksk)
seed = [100]
def enc(bit):
    seed[0] += 1
    return R.encrypt_bits(P, key0, [bit], seed[0])
tables = {"nand": (R.NAND, [1, 1, 1, 0]), "and": (R.AND, [0, 0, 0, 1]), "or": (R.OR, [0, 1, 1, 1]), "xor": (R.XOR, [0, 1, 1, 0])}
for title, (op, expect) in tables.items():
    res = []
    for i in range(4):
        a, b = enc(i & 1), enc((i >> 1) & 1)
        t0 = time.perf_counter()
        out = tfhe.gate_batch(op, a, b)
        print("%s %d %d: %d micro-seconds" % (title, i & 1, (i >> 1) & 1, (time.perf_counter() - t0) * 1e6))
        res.append(int(R.decrypt_bits(P, key0, out)[0]))
    assert res == expect, (title, res, expect)
for i in range(2):
    t0 = time.perf_counter()
    out = tfhe.gate_batch(R.NOT, enc(i))
    print("not %d: %d micro-seconds" % (i, (time.perf_counter() - t0) * 1e6))
    assert int(R.decrypt_bits(P, key0, out)[0]) == 1 - i
print("all truth tables ok")
