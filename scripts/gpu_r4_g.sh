#!/bin/bash
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r4g; mkdir -p $O
for g in 1024 512; do
  echo "== N=2048 gates=$g" | tee -a $O/ab_eo_stairs.log
  RTFHE_N=2048 timeout -k 10 500 python scripts/ab_libs.py $g 5 build/ab/e_base.so:RTFHE_N2048_KERNEL=halves build/ab/e_base.so:RTFHE_N2048_KERNEL=eo build/ab/e_stairs.so:RTFHE_N2048_KERNEL=eo build/ab/e_stairs2.so:RTFHE_N2048_KERNEL=eo 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_eo_stairs.log
done
