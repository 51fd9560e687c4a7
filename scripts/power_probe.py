#!/usr/bin/env python3
"""Board power and shader clock while a backend runs 1024-gate batches back to back for a few seconds (amdsmi / rocm-smi / hwmon, whichever the box offers).
usage: power_probe.py <backend: fft|ntt|xfft> [seconds]      (RTFHE_LIB selects a variant build)"""
import glob, json, os, subprocess, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rustfhe_amd as R

backend = sys.argv[1] if len(sys.argv) > 1 else "fft"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
P = R.Params()
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

def read_power():
    out = {}
    for f in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average") + glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"):
        try: out[f.split("/")[4] + ":" + os.path.basename(f)] = int(open(f).read()) / 1e6
        except Exception: pass
    for f in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
        try:
            cur = [l for l in open(f).read().split("\n") if "*" in l]
            if cur: out[f.split("/")[4] + ":sclk"] = cur[0].strip()
        except Exception: pass
    return out

samples, stop = [], False
def sampler():
    while not stop:
        samples.append((time.time(), read_power()))
        time.sleep(0.05)
idle = read_power()
th = threading.Thread(target=sampler); th.start()
t0 = time.time(); n = 0
e.timer_begin(st)
while time.time() - t0 < secs:
    for _ in range(20): e.gate_batch_dev(R.NAND, d0, d1, do, G, st)
    e.sync(st); n += 20
ms, _ = e.timer_end(st)
stop = True; th.join()
keys = sorted({k for _, s in samples for k in s if "power" in k})
res = {"backend": backend, "lib": os.environ.get("RTFHE_LIB", "shipped"), "launches": n, "ms_per_launch": round(ms / n, 4), "gates_per_s": round(G * n / ms * 1e3, 1), "idle": idle}
for k in keys:
    v = [s[k] for _, s in samples[len(samples) // 4:] if k in s]
    if v: res[k] = {"mean_W": round(float(np.mean(v)), 1), "max_W": round(float(np.max(v)), 1), "n": len(v)}
sclk = [s[k] for _, s in samples[len(samples) // 4:] for k in s if k.endswith(":sclk")]
res["sclk_seen"] = sorted(set(sclk))[:6]
try:
    res["rocm_smi"] = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10).stdout[:600]
except Exception as ex:
    res["rocm_smi"] = repr(ex)
print(json.dumps(res))
