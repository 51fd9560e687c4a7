#!/bin/bash
# rocprofv3 kernel-trace stats of the other kernels (N = 2048 mirror, NTT at both N, latency shapes): per-kernel average durations
# to set beside the sweep logs.  Outputs: gpurun_out/prof_extra/*_kernel_stats.csv
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_extra
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {   # name, env assignments..., -- sweep counts
  local name=$1; shift
  ( export RTFHE_SKIP_STAGES=1 "$@"; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- python3 $REPO/scripts/sweep.py $COUNTS > $OUT/$name.log 2>&1 ) || { echo "$name failed"; tail -5 $OUT/$name.log; return 1; }
  f=$(find $OUT/$name -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/${name}_kernel_stats.csv; head -4 $OUT/${name}_kernel_stats.csv | cut -c1-220
}
COUNTS=1024,1024,1024 run n2048_mirror RTFHE_N=2048 &&
COUNTS=1024,1024,1024 run n1024_ntt RTFHE_BACKEND=ntt &&
COUNTS=1024,1024,1024 run n2048_ntt RTFHE_N=2048 RTFHE_BACKEND=ntt &&
COUNTS=1,1,1 run n1024_single_gate RTFHE_N=1024 &&
COUNTS=1,1,1 run n1024_ntt_single_gate RTFHE_BACKEND=ntt
