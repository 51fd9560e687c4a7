#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output (kernel trace stats + PMC passes) into profiles/<tag>_summary.json/.txt"""
import csv
import glob
import json
import os
import sys

out_dir, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(os.environ.get("GRAFT_REPO_ROOT", root), "gpurun_out", "profiles_" + tag)
os.makedirs(dst, exist_ok=True)
summary = {"tag": tag, "kernels": {}, "pmc": {}}

for f in glob.glob(os.path.join(out_dir, "kt", "**", "*kernel_stats.csv"), recursive=True):
    with open(f) as fh:
        rows = list(csv.DictReader(fh))
    with open(os.path.join(dst, "kernel_stats.csv"), "w") as o:
        o.write(open(f).read())
    for r in rows:
        summary["kernels"][r["Name"][:120]] = {k: r[k] for k in r if k != "Name"}

for f in glob.glob(os.path.join(out_dir, "kt", "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        rows = list(csv.DictReader(fh))
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows if "k_bootstrap" in r["Kernel_Name"]]
    if d:
        summary["k_bootstrap_launch_ns"] = d
        r0 = [r for r in rows if "k_bootstrap" in r["Kernel_Name"]][0]
        summary["k_bootstrap_resources"] = {k: r0.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size")}

for f in glob.glob(os.path.join(out_dir, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        rows = list(csv.DictReader(fh))
    acc = {}
    for r in rows:
        if "k_bootstrap" not in r.get("Kernel_Name", ""):
            continue
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in acc.items():
        summary["pmc"][k] = {"per_launch_mean": sum(v) / len(v), "launches": len(v)}

with open(os.path.join(dst, "summary.json"), "w") as o:
    json.dump(summary, o, indent=1)
print(json.dumps({k: summary[k] for k in summary if k != "kernels"}, indent=1)[:3000])
