#!/bin/bash
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3c; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
for cfg in "1024 fft" "1024 ntt" "2048 fft" "2048 ntt"; do
  set -- $cfg
  echo "== N=$1 backend=$2" | tee -a $O/ab_ksmm_all.log
  RTFHE_N=$1 RTFHE_BACKEND=$2 timeout -k 10 300 python scripts/ab_ksmm.py 1024 2048 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_ksmm_all.log
done
