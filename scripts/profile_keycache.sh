#!/bin/bash
# Does a mixed N = 1024 batch pay for a second copy of the key in the caches?  A 1536-gate batch is a full round of k_bootstrap_pair (reads the
# canonical key layout, 62 MB) followed by a 512-gate tail of k_bootstrap_pair4 (reads its own layout of the same key, another 62 MB).  This
# collects FETCH_SIZE (HBM-side reads, KiB; x2 for bytes on gfx950) and the duration of the tail launch when it runs ALONE (512-gate batches)
# and BEHIND a full round (1536-gate batches) -> gpurun_out/keycache/keycache.json
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/keycache
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export RTFHE_SKIP_STAGES=1
for CASE in 512 1536; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_$CASE -- python3 $REPO/scripts/sweep.py $CASE,$CASE,$CASE > $OUT/pmc_$CASE.log 2>&1 || { echo "pmc $CASE failed"; tail -5 $OUT/pmc_$CASE.log; }
  rocprofv3 --kernel-trace --output-format csv -d $OUT/kt_$CASE -- python3 $REPO/scripts/sweep.py $CASE,$CASE,$CASE > $OUT/kt_$CASE.log 2>&1 || { echo "kt $CASE failed"; tail -5 $OUT/kt_$CASE.log; }
done
python3 - <<PY
import csv, glob, json
out = {}
for case in ("512", "1536"):
    fetch, dur = {}, {}
    for f in glob.glob("$OUT/pmc_%s/**/*counter_collection.csv" % case, recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == "FETCH_SIZE":
                k = r["Kernel_Name"].split("(")[0][:60]
                fetch.setdefault(k, []).append(float(r["Counter_Value"]))
    for f in glob.glob("$OUT/kt_%s/**/*kernel_trace.csv" % case, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            dur.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
    out[case] = {k: {"launches": len(v), "FETCH_SIZE_KiB_per_launch_last3": [round(x) for x in v[-3:]],
                     "ms_per_launch_last3": [round(x, 3) for x in dur.get(k, [])[-3:]]} for k, v in fetch.items() if "bootstrap" in k or "key_switch" in k}
out["_what"] = "N = 1024; 512-gate batches = k_bootstrap_pair4<2> alone; 1536-gate batches = k_bootstrap_pair<4> (1024 gates) then k_bootstrap_pair4<2> (512 gates) on another copy of the key"
json.dump(out, open("$OUT/keycache.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
