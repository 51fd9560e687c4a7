#!/bin/bash
# round 4, run X: k_bootstrap_pair with its wave-private exchanges as 16-byte LDS accesses (the library) against 8-byte ones (p_x0)
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r4x; mkdir -p $O
LIB=rustfhe_amd/librtfhe_hip.so
for g in 1024 768 512; do
  echo "== N=1024 gates=$g" | tee -a $O/ab_x128.log
  timeout -k 10 300 python scripts/ab_libs.py $g 7 build/ab/p_x0.so $LIB 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_x128.log
done
