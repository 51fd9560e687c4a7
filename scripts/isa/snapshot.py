#!/usr/bin/env python3
"""Per-kernel fingerprint of the default device build: a hash of each kernel's instruction stream (labels normalised),
its instruction-class histogram and the compiler-reported registers / scratch / static LDS.  Two trees whose snapshots are
equal ship the same device code -- the check a source clean-up has to pass (CPU only, cross-compile).

    python scripts/isa/snapshot.py out.json [-DX=1 ...]      # compile every device unit under rustfhe_amd/csrc/ and fingerprint the kernels
    python scripts/isa/snapshot.py --diff a.json b.json      # kernels that differ / appeared / disappeared
"""
import collections
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math"]


def cls(op):
    if op.startswith("v_") and "_f64" in op and not op.startswith("v_cmp"):
        return "dp"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "ds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    return "other"


def fingerprint(asm_path, remarks):
    lines = open(asm_path).read().split("\n")
    out = {}
    i = 0
    while i < len(lines):
        l = lines[i]
        m = re.match(r"^(_Z\w+):", l)
        if not m:
            i += 1
            continue
        name = m.group(1)
        j = i + 1
        h = hashlib.sha256()
        hist = collections.Counter()
        n = 0
        while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
            s = lines[j].split(";")[0].strip()
            j += 1
            if not s or s.startswith("."):
                if re.match(r"^\.LBB\d+_\d+:", s):
                    h.update(b"L\n")
                continue
            s = re.sub(r"\.LBB\d+_(\d+)", r".LBB_\1", s)
            s = re.sub(r"\.L\w+\$\w+", ".Lsym", s)
            h.update(s.encode() + b"\n")
            hist[cls(s.split()[0])] += 1
            n += 1
        out[name] = {"sha": h.hexdigest()[:16], "instructions": n, "mix": dict(sorted(hist.items()))}
        i = j
    cur = None
    for line in remarks.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+)", line)
        if m and cur in out:
            k = m.group(1).strip()
            if k in ("VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize", "Occupancy", "LDS Size"):
                out[cur][k] = m.group(2)
    names = subprocess.run(["c++filt"], input="\n".join(out), capture_output=True, text=True).stdout.split("\n")
    return {d: out[k] for k, d in zip(out, names)}


def diff(a, b):
    a, b = json.load(open(a)), json.load(open(b))
    bad = 0
    for k in sorted(set(a) | set(b)):
        if k not in a:
            print("ONLY IN SECOND:", k)
            bad += 1
        elif k not in b:
            print("ONLY IN FIRST: ", k)
            bad += 1
        elif a[k] != b[k]:
            print("DIFFERS:", k, "\n   ", a[k], "\n   ", b[k])
            bad += 1
    print("%d kernels compared, %d differ" % (len(set(a) & set(b)), bad))
    return bad


if __name__ == "__main__":
    if sys.argv[1] == "--diff":
        sys.exit(1 if diff(sys.argv[2], sys.argv[3]) else 0)
    dst, extra = sys.argv[1], sys.argv[2:]
    sys.path.insert(0, ROOT)
    from rustfhe_amd import build as b
    with tempfile.TemporaryDirectory() as td:
        asm, remarks = b.device_asm(td, extra)
        fp = fingerprint(asm, remarks)
    with open(dst, "w") as f:
        json.dump(fp, f, indent=1, sort_keys=True)
    print(len(fp), "kernels ->", dst)
