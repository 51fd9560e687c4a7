#!/usr/bin/env python3
"""Instruction histogram of one kernel of a device assembly file (hipcc -S --cuda-device-only), split at its basic-block
labels: shows which blocks belong to the step loop and how many VALU / DP / DS / VMEM / SALU instructions each holds.
usage: hist.py api.s <mangled-name-substring> [first_label last_label]"""
import collections, re, sys
path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().endswith(":") or (l.startswith("_Z") and key in l and ": ;" in l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
blocks, cur = collections.OrderedDict(), "entry"
blocks[cur] = []
for l in lines[start + 1:end]:
    s = l.strip()
    m = re.match(r"^(\.LBB\d+_\d+):", s)
    if m:
        cur = m.group(1); blocks[cur] = []; continue
    if not s or s.startswith(";") or s.startswith("."):
        continue
    blocks[cur].append(s.split()[0])
def cls(op):
    if op.startswith("v_") and ("_f64" in op) and not op.startswith("v_cmp"): return "dp"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "ds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")): return "vmem"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_"): return "salu"
    return "other"
sel = None
if len(sys.argv) > 4:
    names = list(blocks); a, b = names.index(sys.argv[3]), names.index(sys.argv[4]); sel = names[a:b + 1]
tot = collections.Counter(); ops = collections.Counter()
for name, ins in blocks.items():
    c = collections.Counter(cls(o) for o in ins)
    if sel is None or name in sel:
        tot.update(c); ops.update(ins)
    print(f"{name:12s} n={len(ins):5d} " + " ".join(f"{k}={v}" for k, v in sorted(c.items())))
print("TOTAL", dict(tot))
for o, n in ops.most_common(60):
    print(f"  {o:28s} {n}")
