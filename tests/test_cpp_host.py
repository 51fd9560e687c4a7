"""The C++ host mirror of the hom_nand crate surface (rustfhe_amd/host/hom_nand.hpp): compiles against the C ABI on
CPU; on the GPU box the homnand-bench counterpart runs the reference example's truth-table checks."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "homnand_bench.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "homnand_bench")


def _build():
    import rustfhe_amd as R
    R.load()
    libdir = os.path.join(ROOT, "rustfhe_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", SRC, "-o", EXE, "-L" + libdir, "-lrtfhe_hip",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])


def test_cpp_host_mirror_compiles_and_fails_loudly_without_gpu():
    _build()
    import rustfhe_amd as R
    if R.load().rtfhe_device_count() > 0:
        pytest.skip("GPU present: covered by the gpu test")
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=600)
    assert out.returncode == 2 and "no HIP device" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_cpp_homnand_bench_truth_tables():
    _build()
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "all truth tables ok" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]
