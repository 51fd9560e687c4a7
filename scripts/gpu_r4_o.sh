#!/bin/bash
# round 4, run O: the parity split as the N = 2048 default -- same-process A/B against the top-bit split at every launch shape, the N = 2048
# GPU tests, then the PMC passes of k_bootstrap_eo at 1024 gates
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r4o; mkdir -p $O
LIB=rustfhe_amd/librtfhe_hip.so
for g in 1024 768 512 256 2048 1; do
  echo "== N=2048 gates=$g" | tee -a $O/ab_default.log
  RTFHE_N=2048 timeout -k 10 400 python scripts/ab_libs.py $g 5 $LIB:RTFHE_N2048_KERNEL=halves $LIB:RTFHE_N2048_KERNEL=eo $LIB 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_default.log
done &&
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_soak.py -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest.log &&
timeout -k 10 600 bash scripts/profile_n2048.sh eo > $O/profile_eo.log 2>&1; tail -30 $O/profile_eo.log
