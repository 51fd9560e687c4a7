#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + separate PMC passes of the default bench.
# Usage: bash scripts/profile_gpu.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-r01}; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# the DRIVER's command (--steps 20 --warmup 5) unless the caller gives its own flags: the steady launches of this trace are the launches the
# bench line's ms_per_step is made of (scripts/summarize_prof.py reports them apart from the clock ramp of the first launches)
[ $# -eq 0 ] && set -- --steps 20 --warmup 5
export RTFHE_PROF_WARMUP=$(echo "$@" | sed -n 's/.*--warmup \([0-9]*\).*/\1/p'); RTFHE_PROF_WARMUP=${RTFHE_PROF_WARMUP:-2}
ARGS="--no-cpu-baseline --no-secondary $@"      # the headline launches of the bench line only
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $REPO/bench.py $ARGS > $OUT/kt.log 2>&1 || { tail -20 $OUT/kt.log; exit 1; }
# the same command with the secondary measurements (every BASELINE config + the NTT backend): one stats row per kernel family
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_all -- python3 $REPO/bench.py --no-cpu-baseline "$@" > $OUT/kt_all.log 2>&1 || { echo "kt_all failed"; tail -5 $OUT/kt_all.log; }
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$N -- python3 $REPO/bench.py $ARGS > $OUT/pmc_$N.log 2>&1 || { echo "pmc $C failed"; tail -5 $OUT/pmc_$N.log; }
done
find $OUT -name "*.csv" | head -50
python3 $REPO/scripts/summarize_prof.py $OUT $TAG
