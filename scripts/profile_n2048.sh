#!/bin/bash
# rocprofv3 PMC passes of the N = 2048 kernel on a 1024-gate batch: counters per launch -> gpurun_out/profiles_n2048/pmc_n2048_<kernel>.json
# usage: profile_n2048.sh [eo]   (RTFHE_LIB=build/ab/<variant>.so in the environment profiles a variant build instead of the shipped library)
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_n2048
mkdir -p $OUT $REPO/gpurun_out/profiles_n2048
cd /tmp && export TMPDIR=/tmp
export RTFHE_N=2048 RTFHE_SKIP_STAGES=1
KERN=${1:-eo}
OUT=$OUT/$KERN; mkdir -p $OUT
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE" FETCH_SIZE WRITE_SIZE; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$N -- python3 $REPO/scripts/sweep.py 1024 > $OUT/pmc_$N.log 2>&1 || { echo "pmc $C failed"; tail -5 $OUT/pmc_$N.log; }
done
python3 - <<PY
import csv, glob, json, os
acc = {}
for f in glob.glob("$OUT/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_bootstrap_$KERN" in r.get("Kernel_Name", ""):
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
out = {k: {"per_launch_mean": sum(v) / len(v), "launches": len(v)} for k, v in acc.items()}
out["_what"] = "k_bootstrap_$KERN<...,4>, 1024 gates per launch, N = 2048 (scripts/profile_n2048.sh); FETCH_SIZE in KiB (x2 for bytes on gfx950, see profiles/pmc_traffic.json)"
json.dump(out, open("$REPO/gpurun_out/profiles_n2048/pmc_n2048_$KERN.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])
PY
