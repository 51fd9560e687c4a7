"""Builds rustfhe_amd/librtfhe_hip.so in-tree with hipcc for gfx950.

-ffp-contract=off is REQUIRED for parity: the reference's AVX FFT rounds every product and sum
separately (no FMA); hipcc fuses to v_fma_f64 by default.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librtfhe_hip.so")
SOURCES = ["rtfhe_api.hip", "rtfhe_keygen.cpp", "rtfhe_wire.cpp", "rtfhe_spqlios.cpp"]


def _deps():
    """Every source and header under csrc/ plus the public header: a changed header must rebuild the library."""
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".hpp", ".cpp", ".h"))]
    return files + [os.path.join(HERE, "..", "include", "rtfhe.h"), os.path.join(HERE, "..", "include", "rtfhe_spqlios.h")]


FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function", "-pthread"]


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in _deps())


def build(force=False, verbose=False, extra=()):
    if not (force or stale()):
        return LIB
    # several ranks of one job may get here together: one of them builds (into a temporary name, renamed when complete), the others
    # wait on the lock and find the library fresh
    import fcntl
    with open(LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if force or stale():
            hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
            tmp = LIB + ".tmp.%d" % os.getpid()
            cmd = [hipcc] + FLAGS + list(extra) + ["-x", "hip"] + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", tmp]
            if verbose:
                print(" ".join(cmd).replace(tmp, LIB), flush=True)
            try:
                subprocess.check_call(cmd)
                os.replace(tmp, LIB)
            finally:
                if os.path.exists(tmp):
                    os.remove(tmp)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True,
          extra=[a for a in sys.argv[1:] if a.startswith("-R") or a.startswith("-save")])
