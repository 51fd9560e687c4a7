#!/usr/bin/env python3
"""Builds a VARIANT of librtfhe_hip.so into build/ab/<name>.so (git-ignored, travels to the GPU box) for same-process A/B runs with
scripts/ab_libs.py.  The shipped headers carry no A/B switches: a variant is the shipped source plus a patch (scripts/variants/*.patch,
applied to a scratch copy of rustfhe_amd/csrc/ under build/variant_src/<name>/) and / or extra compiler flags.

    build_variant.py name [--patch scripts/variants/x.patch ...] [-DX=1 ...]
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rustfhe_amd import build as b  # noqa: E402

name, args = sys.argv[1], sys.argv[2:]
patches, extra = [], []
while args:
    a = args.pop(0)
    if a == "--patch":
        patches.append(os.path.abspath(args.pop(0)))
    else:
        extra.append(a)
src = os.path.join(ROOT, "build", "variant_src", name, "rustfhe_amd", "csrc")
shutil.rmtree(os.path.join(ROOT, "build", "variant_src", name), ignore_errors=True)
shutil.copytree(b.CSRC, src)
for p in patches:
    subprocess.check_call(["patch", "-p1", "-d", os.path.join(ROOT, "build", "variant_src", name), "-i", p])
out = os.path.join(ROOT, "build", "ab")
os.makedirs(out, exist_ok=True)
# the sources include "../../include/rtfhe.h" relative to csrc/: give the scratch copy the same neighbourhood
inc = os.path.join(ROOT, "build", "variant_src", name, "include")
if not os.path.exists(inc):
    os.symlink(os.path.join(ROOT, "include"), inc)
lib = b.compile_all(os.path.join(out, name + ".so"), extra, obj_dir=os.path.join(ROOT, "build", "variant_obj", name), csrc=src)
print(lib)
