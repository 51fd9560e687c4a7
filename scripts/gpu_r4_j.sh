#!/bin/bash
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r4j; mkdir -p $O
for g in 1024 768; do
  echo "== NTT N=2048 gates=$g" | tee -a $O/ab_ntt_halves_stairs.log
  RTFHE_N=2048 RTFHE_BACKEND=ntt timeout -k 10 500 python scripts/ab_libs.py $g 3 build/ab/nt_base.so build/ab/nt_stairs.so 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_ntt_halves_stairs.log
done
