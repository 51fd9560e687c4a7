#!/bin/bash
# round 4, run P: the half-width trade of k_bootstrap_eo against the full-width one (build/ab/e_full.so), same process; then the N = 2048 tests
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r4p; mkdir -p $O
LIB=rustfhe_amd/librtfhe_hip.so
for g in 1024 768 512 256 1; do
  echo "== N=2048 gates=$g" | tee -a $O/ab_half_trade.log
  RTFHE_N=2048 timeout -k 10 400 python scripts/ab_libs.py $g 5 build/ab/e_full.so $LIB $LIB:RTFHE_N2048_KERNEL=halves 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_half_trade.log
done &&
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_soak.py -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest.log
