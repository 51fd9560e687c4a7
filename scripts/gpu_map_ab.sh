#!/bin/bash
# placement of a gate's two waves on the SIMDs: same-process A/B of build variants v_*.so, N = 1024 (k_bootstrap_pair) and N = 2048 (k_bootstrap_halves)
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/map
timeout -k 10 400 python scripts/ab_libs.py 1024 5 $(ls build/ab/v_*.so) 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/map/ab.log
RTFHE_N=2048 timeout -k 10 400 python scripts/ab_libs.py 1024 4 $(ls build/ab/v_*.so) 2>&1 | grep -v amdgpu.ids | sed 's/^/n2048 /' | tee -a gpurun_out/map/ab.log
