#!/bin/bash
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r4m; mkdir -p $O
timeout -k 10 900 python scripts/soak.py 400 2>&1 | grep -v amdgpu.ids | tee $O/soak_repeat_launches.log | tail -20
RTFHE_N2048_KERNEL=eo timeout -k 10 400 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/soak_eo_forced.log
import sys; sys.path.insert(0, "scripts")
import soak
bad = soak.run(300, ((2048, ("fft",), (1024, 768, 512, 100)),), emit=lambda s: print(s, flush=True))
print("soak (parity split forced at every shape):", "clean" if bad == 0 else "%d FAILURES" % bad)
PY
