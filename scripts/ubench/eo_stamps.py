#!/usr/bin/env python3
"""Diagnostic: per-phase time shares of the N = 2048 kernel k_bootstrap_eo (library built with -DRTFHE_WG_STAMPS:
    python scripts/build_variant.py stamps -DRTFHE_WG_STAMPS   ->  build/ab/stamps.so).
Phases per CMUX step (summed over both polynomials / components): 0 gather | 1 digits, twist, passes 1-2 of three rows with their exchanges |
2 pass 3 + the three row trades | 3 multiply-accumulate (incl. the wait for the last row) | 4 inverse: trade of the sums + table loads |
5 inverse sub-network, untwist, update | 6 loop top."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rustfhe_amd.build as b
b.LIB = os.path.join(ROOT, "build", "ab", os.environ.get("RTFHE_STAMPS_LIB", "stamps.so"))
b.build = lambda *a, **k: b.LIB
import rustfhe_amd as R
P = R.Params(N=2048)
key0, key1, bk, ksk = R.keygen(P, 1)
e = R.Engine(P, 0)
e.load_bk_torus(bk); e.load_ksk(ksk)
c = R.encrypt_bits(P, key0, [1, 0], 3)
for count in (1024,):
    cc = np.repeat(c[:1], count, axis=0)
    e.gate_batch(R.NAND, cc, cc)
    out = (C.c_ulonglong * 128)()
    e.L.rtfhe_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p]
    assert e.L.rtfhe_debug_read_stamps(e.h, out) == 0
    a = np.array(out[:64], np.float64).reshape(8, 8) / 635.0
    np.set_printoptions(linewidth=200, suppress=True)
    print("count", count, "memtime ticks per step by phase (rows = waves 0..7: parity 0 of gates 0..3, then parity 1)")
    print(np.round(a).astype(int))
    print("per-step total (wave 0):", int(a[0].sum()), " (wave 4):", int(a[4].sum()))
    print("share of phases, mean over waves:", np.round(a.mean(0) / a.mean(0).sum(), 3))
