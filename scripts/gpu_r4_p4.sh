#!/bin/bash
# k_bootstrap_pair4: the GPU suite and a soak of the shapes that run on it
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r4p4; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest.log &&
timeout -k 10 600 python - > $O/soak_pair4.log 2>&1 <<'PY'
import sys
sys.path.insert(0, "scripts")
import soak
shapes = ((1024, ("fft",), (257, 300, 400, 511, 512, 1281, 1400, 1536)),)
bad = soak.run(800, shapes)
print("failures:", bad)
sys.exit(1 if bad else 0)
PY
echo "soak rc=$?"; tail -10 $O/soak_pair4.log
