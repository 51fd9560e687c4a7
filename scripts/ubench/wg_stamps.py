#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of the workgroup-per-gate kernel (library built with -DRTFHE_WG_STAMPS).
Phases per CMUX step: 0 loop head + BK load issue | 1 F (gather, forward transform, spectrum store) | 2 barrier |
3 M (MAC chains) | 4 barrier | 5 I (inverse, update) | 6 barrier."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rustfhe_amd.build as b
b.LIB = os.environ.get("RTFHE_STAMPS_LIB_PATH") or os.path.join(ROOT, "scripts", "ubench", "librtfhe_stamps.so")
b.build = lambda *a, **k: b.LIB
import rustfhe_amd as R
P = R.Params()
key0, key1, bk, ksk = R.keygen(P, 1)
e = R.Engine(P, 0)
e.load_bk_torus(bk); e.load_ksk(ksk)
c = R.encrypt_bits(P, key0, [1, 0], 3)
for count in (1, 256):
    cc = np.repeat(c[:1], count, axis=0)
    e.gate_batch(R.NAND, cc, cc)
    out = (C.c_ulonglong * 128)()
    e.L.rtfhe_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p]
    assert e.L.rtfhe_debug_read_stamps(e.h, out) == 0
    a = np.array(out[:64], np.float64).reshape(8, 8) / 635.0
    print("count", count, "cycles per step by phase (rows = waves 0..7; cols = head, F, bar, M, bar, I, bar, -)")
    np.set_printoptions(linewidth=200, suppress=True)
    print(np.round(a[:, :7]).astype(int))
    print("per-step total (wave 0):", int(a[0, :7].sum()))
