#!/bin/bash
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r4z; mkdir -p $O
LIB=rustfhe_amd/librtfhe_hip.so
for g in 1 256; do
  echo "== N=1024 gates=$g" | tee -a $O/ab_wg_x128.log
  timeout -k 10 300 python scripts/ab_libs.py $g 7 build/ab/wg_late1.so $LIB 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_wg_x128.log
done
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_soak.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.log
