#!/bin/bash
# N = 2048 on the FFT mirror: parity tests, then same-process A/B of build variants
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/h
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "config5_gates" > gpurun_out/h/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/h/pytest.log
[ $rc -eq 0 ] || exit 1
for G in 1024 2048; do
RTFHE_N=2048 timeout -k 10 400 python scripts/ab_libs.py $G 4 $(ls build/ab/h_*.so) 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/h/ab.log
done
