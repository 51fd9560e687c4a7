#!/bin/bash
# rocprofv3 PMC passes (separate runs, no tracing beside them) of ONE kernel family on a batch of <gates> gates: per-launch means of the counters
# -> gpurun_out/pmc_kernel/pmc_<name>_N<N>_g<gates>.json
# usage: profile_kernel.sh <kernel name substring> <gates> [N] [matrix]     ("matrix" adds the MFMA / barrier counters, "cache" replaces the set by the vector-memory / L1 / L2 counters)
#        (RTFHE_LIB=build/ab/<variant>.so in the environment profiles a variant build instead of the shipped library)
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
KERN=$1; GATES=${2:-1024}; export RTFHE_N=${3:-1024} RTFHE_SKIP_STAGES=1
[ -n "$RTFHE_LIB" ] && export RTFHE_LIB=$(realpath "$RTFHE_LIB")      # the passes run from /tmp: a relative path would not be found there
OUT=$REPO/gpurun_out/pmc_kernel/${KERN}_N${RTFHE_N}_g$GATES
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE" FETCH_SIZE WRITE_SIZE)
[ "$4" = matrix ] && SETS+=("SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_I8" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_WAVES SQ_INSTS_WAVE32_LDS SQ_INSTS_SMEM SQ_WAIT_INST_ANY" "TCC_HIT_sum TCC_MISS_sum")
[ "$4" = cache ] && SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_ANY" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "TA_BUSY_avr TA_TA_BUSY_sum" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCC_READ_REQ_LATENCY_sum" FETCH_SIZE)
for C in "${SETS[@]}"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$N -- python3 $REPO/scripts/sweep.py $GATES > $OUT/pmc_$N.log 2>&1 || { echo "pmc $C failed"; tail -3 $OUT/pmc_$N.log; }
done
python3 - <<PY
import csv, glob, json
acc = {}
for f in glob.glob("$OUT/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$KERN" in r.get("Kernel_Name", ""):
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
out = {k: {"per_launch_mean": sum(v) / len(v), "launches": len(v)} for k, v in sorted(acc.items())}
out["_what"] = "$KERN, $GATES gates per launch, N = $RTFHE_N (scripts/profile_kernel.sh); FETCH_SIZE in KiB (x2 for bytes on gfx950, see profiles/pmc_traffic.json)"
json.dump(out, open("$REPO/gpurun_out/pmc_kernel/pmc_${KERN}_N${RTFHE_N}_g$GATES.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
