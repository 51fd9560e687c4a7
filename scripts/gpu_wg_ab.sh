#!/bin/bash
# latency shape (k_bootstrap_wg): same-process A/B of build variants at 1 and 256 gates
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/wg_ab
for G in 1 256; do
timeout -k 10 300 python scripts/ab_libs.py $G 6 $(ls build/ab/w_*.so) 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/wg_ab/ab.log
done
