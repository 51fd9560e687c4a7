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
summary = {"tag": tag, "kernels": {}, "pmc": {}, "pmc_key_switch_mm": {}}
DOM = os.environ.get("RTFHE_PROF_KERNEL", "k_bootstrap_pair")     # the dominant kernel the roofline is about

for f in glob.glob(os.path.join(out_dir, "kt", "**", "*kernel_stats.csv"), recursive=True):
    with open(f) as fh:
        rows = list(csv.DictReader(fh))
    with open(os.path.join(dst, "kernel_stats.csv"), "w") as o:
        o.write(open(f).read())
    for r in rows:
        summary["kernels"][r["Name"][:120]] = {k: r[k] for k in r if k != "Name"}

for f in glob.glob(os.path.join(out_dir, "kt_all", "**", "*kernel_stats.csv"), recursive=True):      # bench.py with its secondary measurements
    with open(os.path.join(dst, "kernel_stats_all_configs.csv"), "w") as o:
        o.write(open(f).read())

for f in glob.glob(os.path.join(out_dir, "kt", "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        rows = list(csv.DictReader(fh))
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows if DOM in r["Kernel_Name"]]
    k2 = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows if "k_key_switch_mm" in r["Kernel_Name"]]
    if k2:
        summary["k_key_switch_mm_launch_ns"] = k2
    if d:
        summary["k_bootstrap_launch_ns"] = d
        # The figure to read is the STEADY one.  rocprofv3's AverageNs (kernel_stats.csv) is the mean over every launch of the run, the first ones
        # of which run while the clock is still ramping up from idle (8.1 -> 6.4 ms over the first five launches): that mean lies above the
        # bench's own ms_per_step and is not what the roofline is priced with.  Steady = the launches behind the command's
        # warm-up launches (RTFHE_PROF_WARMUP, the --warmup the profiled command ran with); median and minimum over all launches beside it.
        warm = int(os.environ.get("RTFHE_PROF_WARMUP", "5"))
        sd = sorted(d)
        steady = d[warm:] if len(d) > warm else d
        summary["k_bootstrap_launch"] = {
            "launches": len(d), "mean_ms_all_launches": round(sum(d) / len(d) / 1e6, 4), "median_ms": round(sd[len(sd) // 2] / 1e6, 4), "min_ms": round(sd[0] / 1e6, 4),
            "steady_launches": len(steady), "steady_mean_ms": round(sum(steady) / len(steady) / 1e6, 4), "steady_max_ms": round(max(steady) / 1e6, 4),
            "first_launches_ms": [round(x / 1e6, 3) for x in d[:warm]],
            "read": "steady_mean_ms (launches after the first %d, the command's warm-up steps); mean_ms_all_launches = rocprofv3's AverageNs includes the clock ramp" % warm}
        # ... and the same table rocprofv3 --stats writes, over the steady launches only (per kernel: its launches after the first `warm`)
        per = {}
        for r in rows:
            per.setdefault(r["Kernel_Name"], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        tot = sum(sum(v[warm:]) for v in per.values() if len(v) > warm) or 1
        with open(os.path.join(dst, "kernel_stats_steady.csv"), "w") as o:
            o.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev","SkippedFirstCalls"\n')
            for name, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
                sv = v[warm:] if len(v) > warm else []
                if not sv:
                    continue
                mean = sum(sv) / len(sv)
                sdv = (sum((x - mean) ** 2 for x in sv) / len(sv)) ** 0.5
                o.write('"%s",%d,%d,%.1f,%.4f,%d,%d,%.1f,%d\n' % (name.replace('"', "'"), len(sv), sum(sv), mean, 100.0 * sum(sv) / tot, min(sv), max(sv), sdv, warm))
        r0 = [r for r in rows if DOM in r["Kernel_Name"]][0]
        summary["k_bootstrap_resources"] = {k: r0.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size")}

for f in glob.glob(os.path.join(out_dir, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        rows = list(csv.DictReader(fh))
    acc, acc2 = {}, {}
    for r in rows:
        if DOM in r.get("Kernel_Name", ""):
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        elif "k_key_switch_mm" in r.get("Kernel_Name", ""):
            acc2.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in acc.items():
        summary["pmc"][k] = {"per_launch_mean": sum(v) / len(v), "launches": len(v)}
    for k, v in acc2.items():
        summary["pmc_key_switch_mm"][k] = {"per_launch_mean": sum(v) / len(v), "launches": len(v)}
    # effective clock of the kernel (MI355X_MICROARCH.md, DVFS section): GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / launch duration,
    # per launch with the counter pass's own timestamps when the CSV carries them
    g = [r for r in rows if DOM in r.get("Kernel_Name", "") and r.get("Counter_Name") == "GRBM_GUI_ACTIVE"]
    if g and "Start_Timestamp" in g[0] and "End_Timestamp" in g[0]:
        clk = sorted(float(r["Counter_Value"]) / 8.0 / max(1, int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in g)
        summary["effective_clock_GHz"] = {"median": round(clk[len(clk) // 2], 3), "min": round(clk[0], 3), "max": round(clk[-1], 3),
                                          "from": "GRBM_GUI_ACTIVE / 8 / (End - Start) of each launch in the counter pass"}

# Measured HBM-side bytes per launch of the bootstrap kernel, corrected as MI355X_MICROARCH.md's HBM section prescribes:
# FETCH_SIZE (KiB-unit counter) doubled on gfx950 for 16-B-per-lane coalesced reads, WRITE_SIZE as is; separate --pmc
# passes.  Stamped with the kernel name and a hash of the device sources so that bench.py only quotes it for the code it
# was measured on (copy it to profiles/pmc_traffic.json together with the summary).
sys.path.insert(0, root)
import bench  # noqa: E402  (launcher half only: no torch, no HIP)
names = [k for k in summary["kernels"] if DOM in k]
if "FETCH_SIZE" in summary["pmc"] and "WRITE_SIZE" in summary["pmc"] and names:
    fetch_kb, write_kb = summary["pmc"]["FETCH_SIZE"]["per_launch_mean"], summary["pmc"]["WRITE_SIZE"]["per_launch_mean"]
    gates = int(os.environ.get("RTFHE_PROF_GATES", "1024"))
    traffic = {
        "gates_per_launch": gates, "kernel": names[0].split("<")[0].split("::")[-1],
        "hbm_bytes_per_launch": int(2 * fetch_kb * 1024 + write_kb * 1024),
        "fetch_size_kb_raw": fetch_kb, "write_size_kb_raw": write_kb, "src_hash": bench.kernel_src_hash(),
        "key_switch_mm_hbm_bytes_per_launch": (int(2 * summary["pmc_key_switch_mm"]["FETCH_SIZE"]["per_launch_mean"] * 1024 +
                                                   summary["pmc_key_switch_mm"]["WRITE_SIZE"]["per_launch_mean"] * 1024)
                                               if "FETCH_SIZE" in summary["pmc_key_switch_mm"] and "WRITE_SIZE" in summary["pmc_key_switch_mm"] else None),
        "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section (gfx950 reports 1/2 of 16B/lane coalesced reads); WRITE_SIZE exact; "
                "separate rocprofv3 --pmc passes of `bench.py --no-cpu-baseline`, mean over all its launches",
        "source": "profiles/%s/summary.json" % tag,
    }
    summary["pmc_traffic"] = traffic
    with open(os.path.join(dst, "pmc_traffic.json"), "w") as o:
        json.dump(traffic, o, indent=1)

if "effective_clock_GHz" not in summary and "GRBM_GUI_ACTIVE" in summary["pmc"] and summary.get("k_bootstrap_launch_ns"):
    d = summary["k_bootstrap_launch_ns"]
    summary["effective_clock_GHz"] = {"mean": round(summary["pmc"]["GRBM_GUI_ACTIVE"]["per_launch_mean"] / 8.0 / (sum(d) / len(d)), 3),
                                      "from": "mean GRBM_GUI_ACTIVE / 8 / mean launch duration of the kernel-trace pass (another run of the same command)"}

with open(os.path.join(dst, "summary.json"), "w") as o:
    json.dump(summary, o, indent=1)
print(json.dumps({k: summary[k] for k in summary if k != "kernels"}, indent=1)[:3000])
