#!/bin/bash
# NTT backend: parity tests, then same-process A/B of build variants at N = 1024 and N = 2048
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/ntt_ab
timeout -k 10 600 python -m pytest tests/test_gpu_ntt.py -m gpu -x -q > gpurun_out/ntt_ab/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/ntt_ab/pytest.log
[ $rc -eq 0 ] || exit 1
RTFHE_BACKEND=ntt timeout -k 10 400 python scripts/ab_libs.py 1024 4 $(ls build/ab/n_*.so) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ntt_ab/ab_n1024.log
RTFHE_N=2048 RTFHE_BACKEND=ntt timeout -k 10 400 python scripts/ab_libs.py 1024 3 $(ls build/ab/n_*.so) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ntt_ab/ab_n2048.log
