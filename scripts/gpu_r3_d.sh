#!/bin/bash
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3d; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
for cfg in "1024 1024" "1024 1" "1024 256" "1024 512" "2048 1024"; do
  set -- $cfg
  echo "== N=$1 gates=$2" | tee -a $O/ab_triv.log
  RTFHE_N=$1 timeout -k 10 300 python scripts/ab_libs.py $2 4 build/ab/p_0ref.so build/ab/p_1triv.so 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_triv.log
done
