#!/bin/bash
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r4k; mkdir -p $O
timeout -k 10 500 python scripts/tune_prio.py 1024 4 2>&1 | grep -v amdgpu.ids | tee $O/pair_priority_search.log
