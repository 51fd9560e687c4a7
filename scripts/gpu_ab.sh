#!/bin/bash
# same-process A/B of every library under build/ab/ (RTFHE_FORCE_WAVES=2: the pair kernel); args: gates rounds tag
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
G=${1:-1024}; R=${2:-5}; TAG=${3:-ab}
mkdir -p gpurun_out/ab
RTFHE_FORCE_WAVES=${RTFHE_FORCE_WAVES:-2} timeout -k 10 500 python scripts/ab_libs.py $G $R build/ab/*.so > gpurun_out/ab/$TAG.log 2>&1; echo "ab rc=$?"; grep -v amdgpu.ids gpurun_out/ab/$TAG.log
