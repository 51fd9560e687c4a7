#!/usr/bin/env python3
"""The three multiply backends on one batch of HomNAND gates: the fft64 mirror (bit-identical to the reference CPU path), the
exact-integer NTT and the split-FFT exact backend -- the last two must agree word for word (exact products), all three must decrypt alike.
usage: exact_backends.py [gates] [N]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustfhe_amd as R

G = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
P = R.Params(N=int(sys.argv[2]) if len(sys.argv) > 2 else 1024)
key0, key1, bk, ksk = R.keygen(P, int(time.time()))
eng = R.Engine(P, 0)
eng.load_bk_torus(bk)            # the exact backends derive their key forms from the torus form at first use
eng.load_ksk(ksk)
rng = np.random.default_rng(1)
b0, b1 = rng.integers(0, 2, G).astype(np.uint8), rng.integers(0, 2, G).astype(np.uint8)
c0, c1 = R.encrypt_bits(P, key0, b0, 1), R.encrypt_bits(P, key0, b1, 2)
outs = {}
for name, backend in (("fft64-mirror", R._ffi.BACKEND_FFT64_MIRROR), ("ntt-exact", R._ffi.BACKEND_NTT_EXACT), ("split-fft-exact", R._ffi.BACKEND_FFT_SPLIT_EXACT)):
    eng.set_backend(backend)
    eng.gate_batch(R.NAND, c0, c1)                       # first call: key form, launch shapes
    t0 = time.perf_counter()
    outs[name] = eng.gate_batch(R.NAND, c0, c1)
    dt = time.perf_counter() - t0
    ok = bool(np.array_equal(R.decrypt_bits(P, key0, outs[name]), 1 - (b0 & b1)))
    print("%-16s %6d gates in %8.2f ms (host buffers in and out)   decrypts to NAND: %s" % (name, G, dt * 1e3, ok))
    assert ok
assert np.array_equal(outs["ntt-exact"], outs["split-fft-exact"]), "two exact backends must give the same words"
differing = int((outs["fft64-mirror"] != outs["ntt-exact"]).any(axis=1).sum())
print("exact backends agree word for word; the mirror's ciphertexts differ from them in %d of %d gates (the reference's FFT rounds: SURVEY H3) "
      "and decrypt to the same bits" % (differing, G))
