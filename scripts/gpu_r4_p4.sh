#!/bin/bash
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r4p4; mkdir -p $O
LIB=rustfhe_amd/librtfhe_hip.so
for g in 512 300; do
  echo "== N=1024 gates=$g" | tee -a $O/ab_trade_planes.log
  timeout -k 10 120 python scripts/ab_libs.py $g 7 build/ab/p4_tb128.so $LIB 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_trade_planes.log || exit 1
done
