#!/bin/bash
# k_bootstrap_pair4 at two and three gates per workgroup: the GPU suite and a soak of the shapes that run on it
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r4p4; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest3.log &&
timeout -k 10 600 python - > $O/soak_pair4_3.log 2>&1 <<'PY'
import sys
sys.path.insert(0, "scripts")
import soak
shapes = ((1024, ("fft",), (513, 600, 700, 767, 768, 1600, 1792, 300)),)
bad = soak.run(800, shapes)
print("failures:", bad)
sys.exit(1 if bad else 0)
PY
echo "soak rc=$?"; tail -10 $O/soak_pair4_3.log
