#!/usr/bin/env python3
"""A/B of several builds of librtfhe_hip.so in ONE process, interleaved rounds: prints per-build median/min launch time
of a NAND batch and whether all outputs agree.  usage: ab_libs.py gates rounds lib1.so lib2.so ...  (RTFHE_FORCE_WAVES etc. apply to all;
an argument of the form lib.so:KEY=VAL[:KEY2=VAL2] sets those environment variables around that build's context creation only)"""
import ctypes as C, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rustfhe_amd as R
from rustfhe_amd import _ffi

G, rounds, libs = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3:]
P = R.Params(N=int(os.environ.get('RTFHE_N', '1024')))
key0, key1, bk, ksk = R.keygen(P, 20211003)
rng = np.random.default_rng(0)
b0, b1 = rng.integers(0, 2, G).astype(np.uint8), rng.integers(0, 2, G).astype(np.uint8)
in0, in1 = R.encrypt_bits(P, key0, b0, 1), R.encrypt_bits(P, key0, b1, 2)
d0 = torch.from_numpy(in0.view(np.int32)).cuda(); d1 = torch.from_numpy(in1.view(np.int32)).cuda()
st = torch.cuda.current_stream().cuda_stream
engines, outs = [], []
names = []
for spec in libs:
    path, *envs = spec.split(":")
    names.append(os.path.basename(path) + ("[" + ",".join(envs) + "]" if envs else ""))
    for kv in envs:
        os.environ[kv.split("=")[0]] = kv.split("=", 1)[1]
    L = C.CDLL(os.path.abspath(path))
    for name, (res, args) in _ffi._SIGNATURES.items():
        fn = getattr(L, name); fn.restype = res; fn.argtypes = [C.POINTER(R.Params) if a == "PP" else a for a in args]
    e = R.Engine.__new__(R.Engine)
    e.L, e.p, e.device = L, P, 0
    h = C.c_void_p(); assert L.rtfhe_ctx_create(C.byref(P), 0, C.byref(h)) == 0
    e.h = h
    e.load_bk_torus(bk); e.load_ksk(ksk)
    if os.environ.get("RTFHE_BACKEND") == "ntt":
        e.set_backend(1)
    if os.environ.get("RTFHE_BACKEND") == "xfft":
        e.set_backend(2)
    engines.append(e); outs.append(torch.empty_like(d0))
    for kv in envs:
        del os.environ[kv.split("=")[0]]
times, ks_times = [[] for _ in libs], [[] for _ in libs]
for r in range(rounds + 1):
    for k, e in enumerate(engines):
        e.timer_begin(st)
        for _ in range(3): e.gate_batch_dev(R.NAND, d0, d1, outs[k], G, st)
        ms, ks_ms, n = e.timer_end_detail(st)
        if r: times[k].append(ms / 3); ks_times[k].append(ks_ms / 3)
same = all(bool(torch.equal(outs[0], o)) for o in outs[1:])
for k, path in enumerate(libs):
    t = np.array(times[k])
    print(json.dumps({"lib": names[k], "gates": G, "median_ms": round(float(np.median(t)), 4), "min_ms": round(float(t.min()), 4),
                      "gates_per_s_median": round(G / np.median(t) * 1e3, 1), "key_switch_ms_median": round(float(np.median(ks_times[k])), 4),
                      "outputs_identical": same}), flush=True)
