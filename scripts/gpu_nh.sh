#!/bin/bash
# N = 2048 on the NTT backend: same-process A/B of build variants
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/nh
for G in 1024 2048; do
RTFHE_N=2048 RTFHE_BACKEND=ntt timeout -k 10 400 python scripts/ab_libs.py $G 3 $(ls build/ab/nh_*.so) 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/nh/ab.log
done
