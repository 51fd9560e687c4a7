#!/usr/bin/env python3
"""Phase stamps of k_bootstrap_xpair (workgroup 0; a build with -DRTFHE_WG_STAMPS, e.g. build/ab/x_stamps.so): cycles per step by phase and wave.
usage: xfft_stamps.py lib.so [gates]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RTFHE_LIB"] = os.path.abspath(sys.argv[1])
import rustfhe_amd as R
G = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
P = R.Params(N=int(os.environ.get("RTFHE_N", "1024")))
key0, key1, bk, ksk = R.keygen(P, 20211003)
e = R.Engine(P, 0)
e.load_bk_torus(bk); e.load_ksk(ksk)
e.set_backend(int(os.environ.get("RTFHE_BACKEND_ID", "2")))
rng = np.random.default_rng(0)
b0, b1 = rng.integers(0, 2, G).astype(np.uint8), rng.integers(0, 2, G).astype(np.uint8)
in0, in1 = R.encrypt_bits(P, key0, b0, 1), R.encrypt_bits(P, key0, b1, 2)
import torch
d0 = torch.from_numpy(in0.view(np.int32)).cuda(); d1 = torch.from_numpy(in1.view(np.int32)).cuda(); do = torch.empty_like(d0)
st = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    e.gate_batch_dev(R.NAND, d0, d1, do, G, st)
e.sync(st)
e.timer_begin(st)
for _ in range(3):
    e.gate_batch_dev(R.NAND, d0, d1, do, G, st)
ms, _n = e.timer_end(st)
print("event time per launch: %.3f ms = %.0f cycles@2.4GHz per step" % (ms / 3, ms / 3 * 2.4e6 / P.n))
out = do.cpu().numpy().view(np.uint32)
buf = (C.c_ulonglong * 128)()
e.L.rtfhe_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p]
assert e.L.rtfhe_debug_read_stamps(e.h, buf) == 0
t = np.array(buf[:], dtype=np.float64).reshape(8, 16) / P.n
names = ["gather+cvt", "forward x3", "M1+put", "sync1", "M2+M3+put", "sync2", "M4", "inverse x2", "update"]
if P.N == 2048:       # k_bootstrap_xquad
    names = ["gather+fold", "forward x3", "M1+put", "sync1", "M2+M3+put", "sync2", "M4", "inverse9 x2", "trade write", "sync3", "last stage+upd", "sync4"]
print("cycles per step (s_memtime ticks = 100 MHz? see note) by wave; ok =", bool(np.array_equal(R.decrypt_bits(P, key0, out), 1 - (b0 & b1))))
for k, nm in enumerate(names):
    print("%-14s" % nm + " ".join("%8.0f" % t[w, k] for w in range(8)))
print("%-12s" % "total" + " ".join("%8.0f" % t[w, :len(names)].sum() for w in range(8)))
big = (C.c_ulonglong * 4096)()
e.L.rtfhe_debug_read_wg_times.argtypes = [C.c_void_p, C.c_void_p]
if e.L.rtfhe_debug_read_wg_times(e.h, big) == 0:
    w = np.array(big[:], dtype=np.int64).reshape(1024, 4)[: min(1024, (G + (1 if P.N == 2048 else 3)) // (2 if P.N == 2048 else 4))]
    t0, t1 = w[:, 0], w[:, 1]
    dur = (t1 - t0) / P.n
    print("per-workgroup loop cycles per step: min %.0f  p10 %.0f  median %.0f  p90 %.0f  max %.0f;  start spread %.0f ticks, end spread %.0f ticks, span(first start, last end) %.0f per step"
          % (dur.min(), np.percentile(dur, 10), np.median(dur), np.percentile(dur, 90), dur.max(), t0.max() - t0.min(), t1.max() - t1.min(), (t1.max() - t0.min()) / P.n))
    ghz = (t1 - t0) / w[:, 2] * 0.1
    print("shader clock over the loop (s_memtime / s_memrealtime at 100 MHz): median %.3f GHz, min %.3f, max %.3f; loop wall time median %.3f ms, max %.3f ms"
          % (np.median(ghz), ghz.min(), ghz.max(), np.median(w[:, 2]) * 1e-5, w[:, 2].max() * 1e-5))
    xcc = w[:, 3] & 0xF
    for x in sorted(set(xcc.tolist())):
        m = xcc == x
        print("  xcc %d: %3d workgroups, median %.0f, max %.0f" % (x, m.sum(), np.median(dur[m]), dur[m].max()))
