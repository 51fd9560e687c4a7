#!/bin/bash
# round 4: k_bootstrap_eo4 (N = 2048, four waves per gate) against k_bootstrap_eo at up to two gates per CU, same library, RTFHE_N2048_EO4 per context
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r4e4; mkdir -p $O
LIB=rustfhe_amd/librtfhe_hip.so
for g in 1 256 300 512 768; do
  echo "== N=2048 gates=$g" | tee -a $O/ab_eo4.log
  RTFHE_N=2048 timeout -k 10 120 python scripts/ab_libs.py $g 5 $LIB:RTFHE_N2048_EO4=0 $LIB:RTFHE_N2048_EO4=1 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_eo4.log || exit 1
done
