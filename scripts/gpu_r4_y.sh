#!/bin/bash
# LDS bank-conflict counters of k_bootstrap_pair with 16-byte exchanges (the library as built)
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r4y; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc -- python3 $REPO/scripts/sweep.py 1024 > $O/pmc.log 2>&1
python3 - <<PY
import csv, glob
acc = {}
for f in glob.glob("$O/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_bootstrap_pair" in r.get("Kernel_Name", ""):
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, v in acc.items(): print(k, sum(v) / len(v), len(v))
PY
