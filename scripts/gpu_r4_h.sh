#!/bin/bash
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r4h; mkdir -p $O
for g in 1024 768; do
  echo "== N=2048 gates=$g (forced top-bit split)" | tee -a $O/ab_halves_stairs.log
  RTFHE_N=2048 RTFHE_N2048_KERNEL=halves timeout -k 10 500 python scripts/ab_libs.py $g 5 build/ab/h_base.so build/ab/h_stairs.so build/ab/h_stairs_ow.so build/ab/n_ow.so 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_halves_stairs.log
done
