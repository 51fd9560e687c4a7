#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of the two-waves-per-gate kernel (library built with -DRTFHE_WG_STAMPS into
scripts/ubench/librtfhe_stamps.so).  Phases per CMUX step: 0 gather/decompose | 1 2 3 forward transforms | 4 slot P (side 0:
component 0 of rows 0..2) | 5 barrier 1 | 6 slot Q | 7 barrier 2 | 8 slot R (side 1) / pick-up of s0 (side 0) |
9 inverse transform + accumulator update."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["RTFHE_FORCE_WAVES"] = "2"
import rustfhe_amd.build as b
b.LIB = os.path.join(ROOT, "scripts", "ubench", os.environ.get("RTFHE_STAMPS_LIB", "librtfhe_stamps.so"))
b.build = lambda *a, **k: b.LIB
import rustfhe_amd as R
P = R.Params()
key0, key1, bk, ksk = R.keygen(P, 1)
e = R.Engine(P, 0)
e.load_bk_torus(bk); e.load_ksk(ksk)
c = R.encrypt_bits(P, key0, [1, 0], 3)
for count in (4, 1024):
    cc = np.repeat(c[:1], count, axis=0)
    e.gate_batch(R.NAND, cc, cc)
    out = (C.c_ulonglong * 128)()
    e.L.rtfhe_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p]
    assert e.L.rtfhe_debug_read_stamps(e.h, out) == 0
    a = np.array(out[:], np.float64).reshape(8, 16) / 635.0
    print("count", count, "memtime ticks per step by phase (rows = waves 0..7: b sides then a sides)")
    np.set_printoptions(linewidth=200, suppress=True)
    print(np.round(a).astype(int))
    print("per-step total (wave 0):", int(a[0].sum()), " (wave 4):", int(a[4].sum()))
