"""Builds rustfhe_amd/librtfhe_hip.so in-tree with hipcc for gfx950: every translation unit under csrc/ to an object of its own (in
parallel), then one link.

-ffp-contract=off is REQUIRED for parity: the reference's AVX FFT rounds every product and sum separately (no FMA); hipcc fuses to
v_fma_f64 by default.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librtfhe_hip.so")
OBJ = os.path.join(HERE, "..", "build", "obj")
# device code lives in the .hip units (every kernel is instantiated in exactly one of them); the .cpp units are host-only
HIP_SOURCES = ["rtfhe_dispatch_fft.hip", "rtfhe_dispatch_ntt.hip", "rtfhe_dispatch_xfft.hip", "rtfhe_stages.hip", "rtfhe_context.hip", "rtfhe_twiddles.hip",
               "rtfhe_batch.hip", "rtfhe_circuit.hip", "rtfhe_multi.hip"]
CPP_SOURCES = ["rtfhe_keygen.cpp", "rtfhe_wire.cpp", "rtfhe_spqlios.cpp"]
SOURCES = HIP_SOURCES + CPP_SOURCES


def _deps():
    """Every source and header under csrc/ plus the public headers: a changed header must rebuild the library."""
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".hpp", ".cpp", ".h"))]
    return files + [os.path.join(HERE, "..", "include", "rtfhe.h"), os.path.join(HERE, "..", "include", "rtfhe_spqlios.h")]


CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function", "-pthread"]
FLAGS = CFLAGS + ["-shared"]      # (what a one-step build of all sources would take; kept for scripts that print it)


def stale(lib=LIB):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(d) > t for d in _deps())


def compile_all(out_lib, extra=(), obj_dir=OBJ, verbose=False, jobs=None, csrc=CSRC):
    """Compiles every unit of `csrc` with CFLAGS + extra into obj_dir and links out_lib (atomically: temporary name, then rename)."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(obj_dir, exist_ok=True)

    def one(src):
        obj = os.path.join(obj_dir, os.path.splitext(src)[0] + ".o")
        cmd = [hipcc] + CFLAGS + list(extra) + ["-I", os.path.join(HERE, "..", "include"), "-c", "-x", "hip", os.path.join(csrc, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=jobs or min(8, os.cpu_count() or 1)) as pool:
        objs = list(pool.map(one, SOURCES))
    tmp = out_lib + ".tmp.%d" % os.getpid()
    try:
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread"] + objs + ["-o", tmp])
        os.replace(tmp, out_lib)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return out_lib


def device_asm(out_dir, extra=()):
    """Device assembly of every .hip unit (hipcc -S --cuda-device-only), concatenated: what the ISA checks of the tests and
    scripts/isa/snapshot.py read.  Returns (path of the concatenated file, compiler remarks on kernel resources)."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(out_dir, exist_ok=True)
    flags = [f for f in CFLAGS if f not in ("-fPIC", "-pthread")]

    def one(src):
        s = os.path.join(out_dir, os.path.splitext(src)[0] + ".s")
        r = subprocess.run([hipcc] + flags + list(extra) + ["--cuda-device-only", "-S", "-Rpass-analysis=kernel-resource-usage", "-x", "hip",
                            os.path.join(CSRC, src), "-o", s], capture_output=True, text=True, check=True)
        return s, r.stderr

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        res = list(pool.map(one, HIP_SOURCES))
    path = os.path.join(out_dir, "device.s")
    with open(path, "w") as f:
        for s, _ in res:
            f.write(open(s).read())
    return path, "".join(r for _, r in res)


def build(force=False, verbose=False, extra=()):
    if not (force or stale()):
        return LIB
    # several ranks of one job may get here together: one of them builds, the others wait on the lock and find the library fresh
    import fcntl
    with open(LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if force or stale():
            compile_all(LIB, extra, verbose=verbose)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True,
          extra=[a for a in sys.argv[1:] if a.startswith("-R") or a.startswith("-save") or a.startswith("-D")])
