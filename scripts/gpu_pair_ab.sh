#!/bin/bash
# headline kernel (k_bootstrap_pair): same-process A/B of build variants at 1024 gates
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/pair_ab
timeout -k 10 400 python scripts/ab_libs.py 1024 5 $(ls build/ab/p_*.so) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/pair_ab/ab.log
