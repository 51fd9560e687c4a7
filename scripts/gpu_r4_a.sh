#!/bin/bash
# round 4, run A: N = 2048 ping-pong trade (parity, A/B, short soak) + N = 1024 hand-offs as 8-byte accesses
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r4a; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "config5" > $O/pytest_config5.log 2>&1; rc=$?; echo "pytest config5 rc=$rc"; tail -3 $O/pytest_config5.log
[ $rc -eq 0 ] || exit $rc
for g in 1024 768 512; do
  echo "== N=2048 gates=$g" | tee -a $O/ab_n2048.log
  RTFHE_N=2048 timeout -k 10 400 python scripts/ab_libs.py $g 5 build/ab/pp0.so build/ab/pp1.so build/ab/pp1_late.so 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_n2048.log
done
echo "== N=1024 gates=1024 hand-off forms" | tee -a $O/ab_hand.log
timeout -k 10 300 python scripts/ab_libs.py 1024 6 build/ab/pp0.so build/ab/hand64.so 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_hand.log
