#!/bin/bash
# N = 2048 kernel: timing ablations / A/B of build variants h_*.so (same process, interleaved rounds); args: gate counts
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/h
for G in ${@:-1024}; do
RTFHE_N=2048 timeout -k 10 400 python scripts/ab_libs.py $G 4 $(ls build/ab/h_*.so) 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/h/abl.log
done
if [ -n "$NTT_GATES" ]; then
for G in $NTT_GATES; do
RTFHE_BACKEND=ntt RTFHE_N=2048 timeout -k 10 400 python scripts/ab_libs.py $G 3 $(ls build/ab/h_*.so) 2>&1 | grep -v amdgpu.ids | sed 's/^/ntt /' | tee -a gpurun_out/h/abl.log
done
fi
