#!/bin/bash
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
mkdir -p gpurun_out/quick
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/quick/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/quick/pytest.log
RTFHE_BACKEND=ntt RTFHE_SKIP_STAGES=1 timeout -k 10 300 python scripts/sweep.py 1,3,256,512,768,1024,2048 > gpurun_out/quick/sweep_ntt.log 2>&1; echo "sweep rc=$?"; grep -v amdgpu.ids gpurun_out/quick/sweep_ntt.log
