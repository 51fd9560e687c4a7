#!/bin/bash
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r4n; mkdir -p $O
for g in 1024 768 512; do
  echo "== N=2048 gates=$g" | tee -a $O/ab_eo2.log
  RTFHE_N=2048 timeout -k 10 500 python scripts/ab_libs.py $g 5 build/ab/h_def.so:RTFHE_N2048_KERNEL=halves build/ab/h_fused.so:RTFHE_N2048_KERNEL=halves build/ab/e_s1.so:RTFHE_N2048_KERNEL=eo build/ab/e_s5.so:RTFHE_N2048_KERNEL=eo build/ab/e_s6.so:RTFHE_N2048_KERNEL=eo build/ab/e_s5g.so:RTFHE_N2048_KERNEL=eo 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_eo2.log
done
