#!/bin/bash
# key-switching key layout: rows of two adjacent levels pre-summed (k_pairs, the default) vs the reference's per-level rows (k_single)
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/ks_ab
timeout -k 10 400 python scripts/ab_libs.py 1024 4 $(ls build/ab/k_*.so) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ks_ab/ab_n1024.log
RTFHE_N=2048 timeout -k 10 400 python scripts/ab_libs.py 1024 4 $(ls build/ab/k_*.so) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ks_ab/ab_n2048.log
