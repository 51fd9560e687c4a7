#!/bin/bash
# rocprofv3 kernel trace of a short script: per-kernel durations (usage: gpu_prof_quick.sh <tag> <python script> [args])
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$REPO/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $REPO/"$@" > $OUT/run.log 2>&1 || { tail -20 $OUT/run.log; exit 1; }
F=$(find $OUT/kt -name "*kernel_stats.csv" | head -1)
cp $F $OUT/kernel_stats.csv; head -12 $F | cut -c1-220
