#!/usr/bin/env python3
"""Compiler-reported resources of every kernel in librtfhe_hip.so (hipcc -Rpass-analysis=kernel-resource-usage, the same
flags as the build) -> profiles/<tag>/kernel_resources.json.  Runs on CPU (cross-compile).  Dynamic LDS is what the host
passes at launch (formulas of rtfhe_dispatch_fft.hip restated below for the default parameter set n = 635).

    python scripts/kernel_resources.py r02
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rustfhe_amd import build as b  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
with tempfile.TemporaryDirectory() as td:
    _, err = b.device_asm(td)
    demangle = lambda s: subprocess.run(["c++filt", s], capture_output=True, text=True).stdout.strip()

kernels, cur = {}, None
for line in err.splitlines():
    m = re.search(r"remark: Function Name: (\S+)", line)
    if m:
        cur = demangle(m.group(1))
        kernels[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+)", line)
    if m and cur:
        kernels[cur][m.group(1).strip()] = m.group(2)

# dynamic LDS per workgroup at launch (bytes), n = 635 -> npad = 640
npad = 640
TW10, XS10 = 2 * 1020 * 16, 512 + 64                 # Geo<10>::TW_TOTAL cplx, XSLOTS doubles
lds = {
    "k_bootstrap_pair<3, 6, 8, 2, 3, 4>": TW10 + 4 * (2 * 1024 * 4 + npad * 4 + 2 * (2 * XS10 * 8)),
    "k_bootstrap<10, 3, 6, 8, 2, 3, 4>": TW10 + 4 * (2 * XS10 * 8 + 2 * 1024 * 4 + npad * 4),
    "k_bootstrap<10, 3, 6, 8, 2, 3, 8>": TW10 + 8 * (XS10 * 8 + 2 * 1024 * 4 + npad * 4),
}
for name, info in kernels.items():
    for k, v in lds.items():
        if k in name:
            info["dynamic LDS at launch [bytes/block]"] = v
out_dir = os.path.join(ROOT, "profiles", tag)
os.makedirs(out_dir, exist_ok=True)
with open(os.path.join(out_dir, "kernel_resources.json"), "w") as f:
    json.dump({"flags": b.FLAGS, "kernels": kernels}, f, indent=1)
for name, info in kernels.items():
    if "k_bootstrap" in name:
        print(name, {k: info[k] for k in info if k in ("VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize", "Occupancy", "VGPRs Spill", "dynamic LDS at launch [bytes/block]")})
