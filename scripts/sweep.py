#!/usr/bin/env python3
"""Batch-size sweep + stage kernels once each (so that a rocprofv3 kernel trace of this script prices them)."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rustfhe_amd as R

counts = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "256,512,1024,2048,4096,8192".split(","))]
P = R.Params(N=int(os.environ.get('RTFHE_N', '1024')))
key0, key1, bk, ksk = R.keygen(P, 20211003)
eng = R.Engine(P, 0)
eng.load_bk_torus(bk); eng.load_ksk(ksk)
if os.environ.get('RTFHE_BACKEND') == 'ntt':
    eng.set_backend(R._ffi.BACKEND_NTT_EXACT)
if os.environ.get('RTFHE_BACKEND') == 'xfft':
    eng.set_backend(R._ffi.BACKEND_FFT_SPLIT_EXACT)
G = max(counts)
rng = np.random.default_rng(0)
b0, b1 = rng.integers(0, 2, G).astype(np.uint8), rng.integers(0, 2, G).astype(np.uint8)
in0, in1 = R.encrypt_bits(P, key0, b0, 1), R.encrypt_bits(P, key0, b1, 2)
d0 = torch.from_numpy(in0.view(np.int32)).cuda(); d1 = torch.from_numpy(in1.view(np.int32)).cuda(); do = torch.empty_like(d0)
st = torch.cuda.current_stream().cuda_stream
for c in counts:
    eng.gate_batch_dev(R.NAND, d0, d1, do, c, st); eng.sync(st)
    reps = 3
    eng.timer_begin(st)
    for _ in range(reps): eng.gate_batch_dev(R.NAND, d0, d1, do, c, st)
    ms, n = eng.timer_end(st)
    ok = bool(np.array_equal(R.decrypt_bits(P, key0, do.cpu().numpy().view(np.uint32)[:c]), (1 - (b0 & b1))[:c]))
    print(json.dumps({"gates": c, "ms_per_launch": round(ms / reps, 3), "gates_per_s": round(c * reps / ms * 1e3, 1), "ok": ok}), flush=True)
if os.environ.get('RTFHE_SKIP_STAGES'):
    sys.exit(0)
# stage kernels, 1024 items each
t1 = rng.integers(0, 2 ** 32, (1024, P.N + 1), dtype=np.uint64).astype(np.uint32)
eng.key_switch_batch(t1)
tr = rng.integers(0, 2 ** 32, (1024, 2 * P.N), dtype=np.uint64).astype(np.uint32)
eng.external_product_batch(rng.integers(0, P.n, 1024).astype(np.int32), tr)
eng.external_product_batch(np.zeros(1024, np.int32), tr)
src = rng.integers(-32, 32, (6144, P.N)).astype(np.int32)
f = eng.ifft_i32_batch(src)
eng.fft_u32_batch(f[:2048])
eng.blind_rotate_batch(in0[:1024], 635)
print("stages done")
