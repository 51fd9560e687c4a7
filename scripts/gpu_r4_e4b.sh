#!/bin/bash
# k_bootstrap_eo4: the N = 2048 GPU tests and a soak of its shapes
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r4e4; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py tests/test_gpu_soak.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest.log &&
timeout -k 10 600 python - > $O/soak_eo4.log 2>&1 <<'PY'
import sys
sys.path.insert(0, "scripts")
import soak
shapes = ((2048, ("fft",), (1, 2, 100, 256, 257, 300, 511, 512, 700, 1280)),)
bad = soak.run(600, shapes)
print("failures:", bad)
sys.exit(1 if bad else 0)
PY
echo "soak rc=$?"; tail -12 $O/soak_eo4.log
