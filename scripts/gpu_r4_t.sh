#!/bin/bash
# round 4, run T: the whole GPU suite, then the restamp (smoke, profile, bench lines) with the parity split as the N = 2048 kernel
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r4t; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; tail -3 $O/pytest_gpu.log; [ $rc -eq 0 ] || exit $rc
bash scripts/gpu_r4_restamp.sh
