"""CPU: the oracle (test infrastructure) built once under AddressSanitizer + UndefinedBehaviorSanitizer and walked end to end
(SURVEY 5).  GPU sanitizers are not available on this pool; the device code shares no source with the oracle."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not shutil.which("gcc"), reason="gcc not available")
def test_oracle_under_asan_and_ubsan(tmp_path):
    exe = tmp_path / "oracle_sanitize"
    cmd = ["gcc", "-std=gnu11", "-O1", "-g", "-fno-omit-frame-pointer", "-ffp-contract=off", "-fsanitize=address,undefined",
           "-fno-sanitize-recover=undefined", "-Wall", "-Wextra", "-pthread", "-I", os.path.join(ROOT, "oracle"),
           os.path.join(ROOT, "oracle", "tfhe_oracle.c"), os.path.join(ROOT, "tests", "c", "oracle_sanitize_main.c"),
           "-o", str(exe), "-lm", "-lpthread"]
    subprocess.check_call(cmd)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0 and "oracle sanitizer walk ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr[-4000:]


@pytest.mark.skipif(not (shutil.which("gcc") and shutil.which("g++")), reason="gcc/g++ not available")
def test_product_host_sources_under_asan_and_ubsan(tmp_path):
    """The product's host-only sources (key generation / encryption, wire format) compiled by g++ under ASan + UBSan and driven
    from a C host: secure and deterministic keygen, encrypt -> decrypt, file round trips, corrupted and truncated files."""
    csrc = os.path.join(ROOT, "rustfhe_amd", "csrc")
    exe = tmp_path / "host_sanitize"
    san = ["-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-Wall", "-Wextra"]
    objs = []
    for src in ("rtfhe_keygen.cpp", "rtfhe_wire.cpp"):
        o = tmp_path / (src + ".o")
        subprocess.check_call(["g++", "-std=c++17"] + san + ["-I", os.path.join(ROOT, "include"), "-c", os.path.join(csrc, src), "-o", str(o)])
        objs.append(str(o))
    o = tmp_path / "main.o"
    subprocess.check_call(["gcc", "-std=gnu11"] + san + ["-I", os.path.join(ROOT, "include"), "-c",
                           os.path.join(ROOT, "tests", "c", "host_sanitize_main.c"), "-o", str(o)])
    subprocess.check_call(["g++", "-fsanitize=address,undefined", str(o)] + objs + ["-o", str(exe), "-pthread"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)
    out = subprocess.run([str(exe), str(tmp_path)], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "host sanitizer walk ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr[-4000:]
