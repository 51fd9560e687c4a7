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
