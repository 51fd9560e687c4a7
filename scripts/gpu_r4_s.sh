#!/bin/bash
# round 4, run S: k_bootstrap_eo with the inverse of the two components as two copies of the code (e_uc) against the loop
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r4s; mkdir -p $O
LIB=rustfhe_amd/librtfhe_hip.so
for g in 1024 768 512 1; do
  echo "== N=2048 gates=$g" | tee -a $O/ab_unroll_comp.log
  RTFHE_N=2048 timeout -k 10 400 python scripts/ab_libs.py $g 5 $LIB build/ab/e_uc.so 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_unroll_comp.log
done &&
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_soak.py -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest.log
