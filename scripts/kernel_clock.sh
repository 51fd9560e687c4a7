#!/bin/bash
# The shader clock a binary's kernels actually ran at: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / the launch's duration, launch by launch.
# Two rocprofv3 runs of the same binary (counters in their own run, tracing in the other), joined by dispatch order.
# usage: kernel_clock.sh <binary or "python3 script args">   ->  gpurun_out/clock/<name>.json
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
NAME=$(basename $1)
OUT=$REPO/gpurun_out/clock/$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- "$@" > $OUT/kt.log 2>&1 || { tail -5 $OUT/kt.log; exit 1; }
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc -- "$@" > $OUT/pmc.log 2>&1 || { tail -5 $OUT/pmc.log; exit 1; }
python3 - <<PY
import csv, glob, json, collections
kt = sorted((r for f in glob.glob("$OUT/kt/**/*kernel_trace.csv", recursive=True) for r in csv.DictReader(open(f))), key=lambda r: int(r["Start_Timestamp"]))
pm = [r for f in glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f)) if r["Counter_Name"] == "GRBM_GUI_ACTIVE"]
pm.sort(key=lambda r: int(r["Dispatch_Id"]))
rows = []
for a, b in zip(kt, pm):
    ns = int(a["End_Timestamp"]) - int(a["Start_Timestamp"])
    cyc = float(b["Counter_Value"]) / 8
    rows.append({"kernel": a["Kernel_Name"][:60], "us_traced": ns / 1e3, "cycles_counted_run": cyc})
# (the two runs are different executions: compare per kernel NAME the mean duration with the mean cycle count)
agg = collections.OrderedDict()
for i, r in enumerate(rows):
    k = agg.setdefault((r["kernel"], i if len(rows) < 80 else 0), [0, 0.0, 0.0]); k[0] += 1; k[1] += r["us_traced"]; k[2] += r["cycles_counted_run"]
out = [{"kernel": k[0], "launch": k[1], "us": v[1] / v[0], "cycles": v[2] / v[0], "GHz": v[2] / v[1] / 1e3} for k, v in agg.items()]
json.dump(out, open("$REPO/gpurun_out/clock/$NAME.json", "w"), indent=1)
for o in out[:80]: print("%-62s %10.1f us %12.0f cycles  %.3f GHz" % (o["kernel"], o["us"], o["cycles"], o["GHz"]))
PY
