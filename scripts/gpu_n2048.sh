#!/bin/bash
# N = 2048 (BASELINE config 5): parity tests, then the batch sweep for both kernel shapes
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
mkdir -p gpurun_out/n2048
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "config5" > gpurun_out/n2048/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/n2048/pytest.log
RTFHE_N=2048 RTFHE_SKIP_STAGES=1 timeout -k 10 300 python scripts/sweep.py 256,1024,2048,4096 > gpurun_out/n2048/sweep_halves.log 2>&1; echo "sweep halves rc=$?"; grep -v amdgpu.ids gpurun_out/n2048/sweep_halves.log
RTFHE_FORCE_WAVES=4 RTFHE_N=2048 RTFHE_SKIP_STAGES=1 timeout -k 10 300 python scripts/sweep.py 1024,2048 > gpurun_out/n2048/sweep_onewave.log 2>&1; echo "sweep one-wave rc=$?"; grep -v amdgpu.ids gpurun_out/n2048/sweep_onewave.log
