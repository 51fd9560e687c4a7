#!/bin/bash
# round 4, run Q: k_bootstrap_eo with rows 0 and 1 in one trade against one trade per row (e_m0), and the staircase forms on top of it
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r4q; mkdir -p $O
LIB=rustfhe_amd/librtfhe_hip.so
for g in 1024 768 512 1; do
  echo "== N=2048 gates=$g" | tee -a $O/ab_merge.log
  RTFHE_N=2048 timeout -k 10 400 python scripts/ab_libs.py $g 5 build/ab/e_m0.so $LIB build/ab/e_m1s1.so build/ab/e_m1s0.so 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_merge.log
done &&
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_soak.py -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest.log
