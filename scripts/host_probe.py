#!/usr/bin/env python3
"""What the host of a GPU box looks like to an ordinary user: CPUs, cgroup quota, memory nodes, toolchains (one JSON object)."""
import json, os, shutil, subprocess, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import orc
t = orc.host_topology()
out = {"cpu_model": t["cpu_model"], "hw_threads": len(t["hw_threads"]), "physical_cores": len(t["one_thread_per_core"]), "memory_nodes": t["nodes"],
       "cgroup_cpu_quota": t["cgroup_cpu_quota"], "nproc": os.cpu_count()}
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpuset.cpus.effective", "/proc/meminfo"):
    try:
        out[f] = open(f).read().strip().split("\n")[0]
    except OSError as e:
        out[f] = "unreadable: %s" % e.strerror
for tool in ("cargo", "rustc", "gcc", "hipcc", "numactl", "lscpu"):
    out["which_" + tool] = shutil.which(tool)
try:
    out["lscpu"] = [ln for ln in subprocess.run(["lscpu"], capture_output=True, text=True, timeout=20).stdout.split("\n")
                    if any(k in ln for k in ("Model name", "Socket", "NUMA", "Thread", "Core(s)", "L3", "Flags"))][:12]
    out["lscpu"] = [ln[:200] for ln in out["lscpu"]]
except Exception as e:
    out["lscpu"] = str(e)
print(json.dumps(out, indent=1))
