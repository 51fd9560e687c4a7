#!/bin/bash
# round 4, run V: priority of the latency kernel's half-row waves during the F phase (0 / 1 / 2 = the library / 3)
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r4v; mkdir -p $O
LIB=rustfhe_amd/librtfhe_hip.so
for g in 1 256; do
  echo "== N=1024 gates=$g" | tee -a $O/ab_wg_prio2.log
  timeout -k 10 300 python scripts/ab_libs.py $g 7 $LIB build/ab/wg_late1.so build/ab/wg_late3.so build/ab/wg_early1.so 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_wg_prio2.log
done
