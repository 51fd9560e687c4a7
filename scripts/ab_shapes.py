#!/usr/bin/env python3
"""A/B of launch shapes (RTFHE_FORCE_WAVES values) of ONE build in ONE process, interleaved rounds: prints per-shape
median/min launch time of a NAND batch and whether all outputs agree.  usage: ab_shapes.py "4,2" [gates] [rounds]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rustfhe_amd as R

shapes = sys.argv[1].split(",")
G = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 7
P = R.Params()
key0, key1, bk, ksk = R.keygen(P, 20211003)
rng = np.random.default_rng(0)
b0, b1 = rng.integers(0, 2, G).astype(np.uint8), rng.integers(0, 2, G).astype(np.uint8)
in0, in1 = R.encrypt_bits(P, key0, b0, 1), R.encrypt_bits(P, key0, b1, 2)
d0 = torch.from_numpy(in0.view(np.int32)).cuda(); d1 = torch.from_numpy(in1.view(np.int32)).cuda()
st = torch.cuda.current_stream().cuda_stream
engines, outs = [], []
for f in shapes:
    os.environ["RTFHE_FORCE_WAVES"] = f
    e = R.Engine(P, 0)
    e.load_bk_torus(bk); e.load_ksk(ksk)
    if os.environ.get("RTFHE_BACKEND") == "ntt":
        e.set_backend(1)
    engines.append(e); outs.append(torch.empty_like(d0))
times = [[] for _ in shapes]
for r in range(rounds + 1):
    for k, e in enumerate(engines):
        e.timer_begin(st)
        for _ in range(3): e.gate_batch_dev(R.NAND, d0, d1, outs[k], G, st)
        ms, n = e.timer_end(st)
        if r: times[k].append(ms / 3)
same = all(bool(torch.equal(outs[0], o)) for o in outs[1:])
for k, f in enumerate(shapes):
    t = np.array(times[k])
    print(json.dumps({"force_waves": f, "gates": G, "median_ms": round(float(np.median(t)), 4), "min_ms": round(float(t.min()), 4),
                      "gates_per_s_median": round(G / np.median(t) * 1e3, 1), "outputs_identical": same}), flush=True)
