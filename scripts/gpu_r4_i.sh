#!/bin/bash
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r4i; mkdir -p $O
for g in 1024 768; do
  echo "== N=2048 gates=$g (forced top-bit split)" | tee -a $O/ab_halves_stairs2.log
  RTFHE_N=2048 RTFHE_N2048_KERNEL=halves timeout -k 10 500 python scripts/ab_libs.py $g 5 build/ab/h_base.so build/ab/h_def.so build/ab/h_fine.so 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_halves_stairs2.log
done
echo "== N=1024 gates=1024 symmetric staircase" | tee -a $O/ab_pair_stairs.log
timeout -k 10 300 python scripts/ab_libs.py 1024 6 build/ab/p_base.so build/ab/p_stairs.so 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_pair_stairs.log
