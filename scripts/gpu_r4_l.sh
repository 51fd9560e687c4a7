#!/bin/bash
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r4l; mkdir -p $O
for g in 1024 768; do
  echo "== NTT N=1024 gates=$g" | tee -a $O/ab_ntt_pair_stairs.log
  RTFHE_BACKEND=ntt timeout -k 10 500 python scripts/ab_libs.py $g 4 build/ab/np_base.so build/ab/np_stairs.so 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_ntt_pair_stairs.log
done
