#!/bin/bash
# round 4, run E: the even / odd N = 2048 kernel: parity (config5 tests, soak), A/B against the top-bit split
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r4e; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_soak.py tests/test_gpu_multidev.py -m gpu -x -q -k "config5 or soak or 2048" > $O/pytest_config5.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -8 $O/pytest_config5.log
[ $rc -eq 0 ] || exit $rc
for g in 1024 768 512 256; do
  echo "== N=2048 gates=$g" | tee -a $O/ab_eo.log
  RTFHE_N=2048 timeout -k 10 500 python scripts/ab_libs.py $g 5 build/ab/e_base.so:RTFHE_N2048_KERNEL=halves build/ab/e_base.so:RTFHE_N2048_KERNEL=eo build/ab/e_prio1.so:RTFHE_N2048_KERNEL=eo 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_eo.log
done
