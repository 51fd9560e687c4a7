#!/bin/bash
# GPU pass B: A/B of pair-kernel variants (same process, outputs compared), then the full GPU test suite on the default build
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
mkdir -p gpurun_out/r02b
RTFHE_FORCE_WAVES=2 timeout -k 10 400 python scripts/ab_libs.py 1024 7 build/ab/*.so > gpurun_out/r02b/ab_1024.log 2>&1; echo "ab rc=$?"; cat gpurun_out/r02b/ab_1024.log | grep -v amdgpu.ids
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02b/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r02b/pytest_gpu.log
