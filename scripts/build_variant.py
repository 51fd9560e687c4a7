#!/usr/bin/env python3
"""Builds a variant of librtfhe_hip.so with extra -D flags into build/ab/<name>.so (git-ignored, travels to the GPU box) for
same-process A/B runs with scripts/ab_libs.py.   usage: build_variant.py name [-DX=1 ...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rustfhe_amd import build as b  # noqa: E402

name, extra = sys.argv[1], sys.argv[2:]
out = os.path.join(ROOT, "build", "ab")
os.makedirs(out, exist_ok=True)
lib = os.path.join(out, name + ".so")
cmd = ["/opt/rocm/bin/hipcc"] + b.FLAGS + extra + ["-x", "hip"] + [os.path.join(b.CSRC, s) for s in b.SOURCES] + ["-o", lib]
subprocess.check_call(cmd)
print(lib)
