#!/usr/bin/env python3
"""Same-process A/B of the fused key switch against the split path (blind rotate + extract, then the batch key switch on the i8
matrix pipe): two engines of ONE library, RTFHE_KS_MM_MIN=0 (fused) and the default; interleaved rounds, outputs compared.
usage: ab_ksmm.py [gates ...]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rustfhe_amd as R

counts = [int(x) for x in sys.argv[1:]] or [1024, 2048, 8192]
P = R.Params(N=int(os.environ.get('RTFHE_N', '1024')))
key0, key1, bk, ksk = R.keygen(P, 20211003)
G = max(counts)
rng = np.random.default_rng(0)
b0, b1 = rng.integers(0, 2, G).astype(np.uint8), rng.integers(0, 2, G).astype(np.uint8)
in0, in1 = R.encrypt_bits(P, key0, b0, 1), R.encrypt_bits(P, key0, b1, 2)
d0 = torch.from_numpy(in0.view(np.int32)).cuda(); d1 = torch.from_numpy(in1.view(np.int32)).cuda()
st = torch.cuda.current_stream().cuda_stream
engines = {}
for name, v in (("fused", "0"), ("split", "1024")):
    os.environ["RTFHE_KS_MM_MIN"] = v
    e = R.Engine(P, 0)
    e.load_bk_torus(bk); e.load_ksk(ksk)
    if os.environ.get("RTFHE_BACKEND") == "ntt":
        e.set_backend(R._ffi.BACKEND_NTT_EXACT)
    engines[name] = (e, torch.empty_like(d0))
for c in counts:
    times = {k: [] for k in engines}
    for r in range(6):
        for name, (e, out) in engines.items():
            e.timer_begin(st)
            for _ in range(3): e.gate_batch_dev(R.NAND, d0, d1, out, c, st)
            ms, n = e.timer_end(st)
            if r: times[name].append(ms / 3)
    same = bool(torch.equal(engines["fused"][1][:c], engines["split"][1][:c]))
    ok = bool(np.array_equal(R.decrypt_bits(P, key0, engines["split"][1][:c].cpu().numpy().view(np.uint32)), (1 - (b0 & b1))[:c]))
    print(json.dumps({"gates": c, **{k + "_ms": round(float(np.median(v)), 4) for k, v in times.items()},
                      "split_gates_per_s": round(c / np.median(times["split"]) * 1e3, 1), "outputs_identical": same, "decrypt_ok": ok}), flush=True)
