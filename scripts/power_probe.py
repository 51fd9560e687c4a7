#!/usr/bin/env python3
"""Socket power of the GPU this process runs on while a backend bootstraps 1,024-gate batches back to back (rocm-smi --showpower, sampled from a
thread twice a second under load; the box shows this container one card), beside the gates/s of the same seconds.
usage: power_probe.py <backend: fft|ntt|xfft> [seconds] [N]      (RTFHE_LIB selects a variant build)"""
import json, os, re, subprocess, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rustfhe_amd as R

backend = sys.argv[1] if len(sys.argv) > 1 else "fft"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
P = R.Params(N=int(sys.argv[3]) if len(sys.argv) > 3 else 1024)
key0, key1, bk, ksk = R.keygen(P, 20211003)
e = R.Engine(P, 0)
e.load_bk_torus(bk); e.load_ksk(ksk)
e.set_backend({"fft": 0, "ntt": 1, "xfft": 2}[backend])
G = 1024
rng = np.random.default_rng(0)
b0, b1 = rng.integers(0, 2, G).astype(np.uint8), rng.integers(0, 2, G).astype(np.uint8)
d0 = torch.from_numpy(R.encrypt_bits(P, key0, b0, 1).view(np.int32)).cuda(); d1 = torch.from_numpy(R.encrypt_bits(P, key0, b1, 2).view(np.int32)).cuda()
do = torch.empty_like(d0)
st = torch.cuda.current_stream().cuda_stream


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10).stdout
        j = json.loads(out)
        card = j[sorted(j)[0]]
        watts = [float(v) for k, v in card.items() if "Power (W)" in k]
        sclk = [int(re.sub(r"[^0-9]", "", v)) for k, v in card.items() if k.startswith("sclk clock speed")]
        return (watts[0] if watts else None), (sclk[0] if sclk else None)
    except Exception:       # noqa: BLE001
        return None, None


idle_w, idle_clk = smi()
samples, stop = [], False


def sampler():
    while not stop:
        samples.append((time.time(),) + smi())
        time.sleep(0.4)


for _ in range(20):
    e.gate_batch_dev(R.NAND, d0, d1, do, G, st)
e.sync(st)
th = threading.Thread(target=sampler); th.start()
t0 = time.time(); n = 0
e.timer_begin(st)
while time.time() - t0 < secs:
    for _ in range(20):
        e.gate_batch_dev(R.NAND, d0, d1, do, G, st)
    e.sync(st); n += 20
ms, _ = e.timer_end(st)
t1 = time.time()
stop = True; th.join()
w = [s[1] for s in samples if s[1] is not None and t0 + 1.0 < s[0] < t1]
c = [s[2] for s in samples if s[2] is not None and t0 + 1.0 < s[0] < t1]
print(json.dumps({"backend": backend, "N": P.N, "lib": os.environ.get("RTFHE_LIB", "shipped"), "launches": n, "ms_per_launch": round(ms / n, 4),
                  "gates_per_s": round(G * n / ms * 1e3, 1), "idle_W": idle_w, "idle_sclk_MHz": idle_clk,
                  "socket_power_W_under_load": {"mean": round(float(np.mean(w)), 1), "min": min(w), "max": max(w), "samples": len(w)} if w else None,
                  "sclk_MHz_reported_under_load": {"mean": round(float(np.mean(c)), 1), "min": min(c), "max": max(c)} if c else None,
                  "note": "sclk is the driver's reported clock level, not the in-kernel clock (MI355X_MICROARCH.md: up to 10 % apart); the in-kernel clock is "
                          "what the stamped builds measure (scripts/xfft_stamps.py)"}))
