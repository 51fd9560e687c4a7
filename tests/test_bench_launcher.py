"""CPU: bench.py's own rank launcher (`python bench.py --gpus N` must start N ranks itself) and the bookkeeping behind
the roofline line (FP64 wave-instruction count derived from the transform structure vs the ISA of the built kernel)."""
import json
import os
import re
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHILD = r'''
import json, os, sys
import torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["LOCAL_RANK"] == os.environ["RANK"] and os.environ["MASTER_ADDR"] == "127.0.0.1"
dist.init_process_group("gloo")
t = torch.tensor([rank + 1.0])
dist.all_reduce(t)
if rank == 0:
    print(json.dumps({"n_gpus": dist.get_world_size(), "sum": float(t[0])}), flush=True)
dist.barrier()
dist.destroy_process_group()
'''


def test_spawn_ranks_sets_env_and_relays_rank0(tmp_path, capsys):
    import bench
    child = tmp_path / "child.py"
    child.write_text(CHILD)
    rc, lines = bench.spawn_ranks(2, [sys.executable, str(child)], timeout=240)
    assert rc == 0
    got = [json.loads(ln) for ln in lines if ln.startswith("{")]
    assert got == [{"n_gpus": 2, "sum": 3.0}]
    assert '"n_gpus": 2' in capsys.readouterr().out          # relayed to our own stdout


def test_spawn_ranks_fails_when_any_rank_fails(tmp_path):
    import bench
    child = tmp_path / "child.py"
    child.write_text("import os, sys, time\nif os.environ['RANK'] == '1':\n    sys.exit(3)\ntime.sleep(600)\n")
    t0 = time.time()
    rc, lines = bench.spawn_ranks(2, [sys.executable, str(child)], timeout=240)
    assert rc == 3 and time.time() - t0 < 60                  # the surviving rank was terminated, not waited for


def test_bare_bench_gpus2_launches_children_and_reports_their_failure():
    """No GPU here: both children must come up as ranks (own env) and refuse loudly; the parent exits non-zero without
    printing a JSON line of its own -- and without importing torch."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0
    assert "rank" in r.stderr and "needs a GPU" in r.stderr
    assert "{" not in r.stdout


def test_parent_does_not_import_torch_before_spawning():
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[:src.index("def run_rank")]
    assert not re.search(r"^\s*import torch|^\s*from torch|import rustfhe_amd", head.replace("    import torch, torch", ""), re.M), \
        "the launcher half of bench.py must not import torch or the HIP library"


def test_dp_opcount_formula():
    import bench
    ops = bench.dp_wave_instr_per_cmux(1024, 3)
    assert ops["transform"] == 360 and ops["mac_row"] == 64
    assert ops["reference"] == 3808 and ops["cvt_trunc"] == 128
    assert ops["total"] == 3808 - 8 * 6 - 16 and ops["add_mul"] == 3680 - 8 * 6 - 16      # unit-twiddle butterflies, first fold row
    assert bench.dp_wave_instr_per_cmux(2048, 3)["transform"] == 800
    assert bench.dp_wave_instr_per_cmux(2048, 3)["reference"] == 8256
    assert bench.dp_wave_instr_per_cmux(2048, 3)["total"] == 8256 - 8 * 18
    assert abs(bench.FP64_VALU_PEAK - 39.3216e12) < 1e6
    assert bench.ALG_BYTES_PER_GATE == 78061008


@pytest.fixture(scope="module")
def device_asm(tmp_path_factory):
    """device assembly of every unit of the library, compiled once for the ISA checks below (hipcc -S --cuda-device-only, the build's flags)"""
    import pathlib
    from rustfhe_amd import build as b
    path, _ = b.device_asm(str(tmp_path_factory.mktemp("isa")))
    return pathlib.Path(path)


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
def test_dp_opcount_matches_the_isa_of_the_built_kernel(device_asm):
    """The count behind roofline.achieved is not a literal: the formula must agree with the v_add/v_mul_f64 (+ cvt/trunc)
    instructions hipcc emits for k_bootstrap_pair's step loop.  The loop body holds slot P (side 0 only), slot Q (both)
    and slot R (side 1 only): a wave executes all of it but one three-row slot, so static = per-wave + 3 MAC rows."""
    import bench
    from rustfhe_amd import build as b
    asm = device_asm
    text = asm.read_text()
    m = re.search(r"\n(_ZN5rtfhe16k_bootstrap_pair\w+):[^\n]*\n(.*?)\n\.Lfunc_end", text, re.S)
    assert m, "k_bootstrap_pair not found in the device assembly"
    body = m.group(2)
    arith = len(re.findall(r"^\s*v_(?:add|mul)_f64", body, re.M))
    fused = len(re.findall(r"^\s*v_fma_f64", body, re.M))
    cvt = len(re.findall(r"^\s*v_(?:cvt_f64_i32|trunc_f64)", body, re.M))
    assert fused == 0, "-ffp-contract=off is what parity rests on"
    ops = bench.dp_wave_instr_per_cmux(1024, 3)
    per_wave_cvt = ops["cvt_trunc"] // 2
    # static step-loop body of one wave's code: three forward transforms + one inverse (each without its unit-twiddle butterfly's 6), nine
    # multiply-accumulate rows (slots P, Q, R) of which slot P's first lacks its 16 fold sums, the 16 magic-constant adds of the truncation
    tr = ops["transform"] - ops["unit_twiddle_saved_per_transform"]
    static = 4 * tr + 9 * ops["mac_row"] - ops["first_row_saved"] + 16
    assert arith == static, (arith, static)
    # ... and the per-gate count the roofline uses is what the two waves of a gate execute of it: P + Q on side 0, Q + R on side 1
    assert ops["add_mul"] == 2 * (4 * tr + 6 * ops["mac_row"] + 16) - ops["first_row_saved"]
    assert cvt == per_wave_cvt, (cvt, per_wave_cvt)


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
def test_time_sliced_kernel_executes_the_pair_kernels_operation_list(device_asm):
    """k_bootstrap_pair_rr (five or six gates time-sliced over a CU's four wave pairs) claims k_bootstrap_pair's arithmetic per CMUX: the same
    static count of FP64 instructions in its code, none of them fused."""
    import bench
    text = device_asm.read_text()
    counts = {}
    for name in ("16k_bootstrap_pair", "19k_bootstrap_pair_rr"):
        m = re.search(r"\n(_ZN5rtfhe%s\w+):[^\n]*\n(.*?)\n\.Lfunc_end" % name, text, re.S)
        assert m, name
        body = m.group(2)
        assert not re.findall(r"^\s*v_fma_f64", body, re.M), name
        counts[name] = (len(re.findall(r"^\s*v_(?:add|mul)_f64", body, re.M)), len(re.findall(r"^\s*v_(?:cvt_f64_i32|trunc_f64)", body, re.M)))
    assert counts["19k_bootstrap_pair_rr"] == counts["16k_bootstrap_pair"], counts
    assert counts["16k_bootstrap_pair"][1] == bench.dp_wave_instr_per_cmux(1024, 3)["cvt_trunc"] // 2


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
def test_n2048_parity_split_kernel_executes_the_reference_operation_list(device_asm):
    """k_bootstrap_eo holds its step loop twice (one copy per parity, each with its polynomial loop and its component loop not unrolled): the
    FP64-rate instructions of the two copies together are what the two waves of a gate execute per (polynomial + component) -- the reference's
    operation list for N = 2048 (bench.dp_wave_instr_per_cmux: 8,256 per CMUX = 2 x 4,128) minus the unit-twiddle butterflies of the
    even-point network (3 per transform x 6 instructions: one in the halfnn = 8 stage, two in the halfnn = 4 stage), and no v_fma_f64."""
    import bench
    from rustfhe_amd import build as b
    asm = device_asm
    m = re.search(r"\n(_ZN5rtfhe14k_bootstrap_eoILi3ELi6ELi8ELi2ELi3ELi4E\w+):[^\n]*\n(.*?)\n\.Lfunc_end", asm.read_text(), re.S)
    assert m, "k_bootstrap_eo not found in the device assembly"
    f64 = len(re.findall(r"^\s*v_(?:add|mul|cvt_f64_i32|trunc)_f64|^\s*v_cvt_f64_i32|^\s*v_trunc_f64", m.group(2), re.M))
    assert len(re.findall(r"^\s*v_fma_f64", m.group(2), re.M)) == 0
    ref = bench.dp_wave_instr_per_cmux(2048, 3)["reference"]
    # static = (one polynomial + one component) per parity = half a CMUX step per parity; the even parity saves 18 per transform, 4 transforms in it
    assert f64 == ref // 2 - 4 * 18, (f64, ref)
    assert 2 * f64 == bench.dp_wave_instr_per_cmux(2048, 3)["total"]          # the count bench.py prices secondary.config5 with


def test_ntt_opcount_formula():
    import bench
    o = bench.ntt_dp_wave_instr_per_cmux(1024, 3)
    assert o["forward"] == 596 and o["inverse"] == 832 and o["mac_row"] == 224        # round 4: 688 + 8 conversions / 880 (two table stages instead of one,
    assert o["per_wave"] == 3324 and o["total"] == 6648 and o["loop_static"] == 1684  # one renormalisation less per transform); round 4: 7,344 per CMUX
    assert bench.ntt_dp_wave_instr_per_cmux(2048, 3)["total"] == 19968


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
def test_ntt_opcount_matches_the_isa_of_the_built_kernel(device_asm):
    """k_bootstrap_ntt_pair's step loop holds the row loop's body once (not unrolled) and the step's tail: the static count of
    FP64-rate instructions must be the formula's one-row count."""
    import bench
    from rustfhe_amd import build as b
    asm = device_asm
    text = asm.read_text()
    m = re.search(r"\n(_ZN5rtfhe20k_bootstrap_ntt_pairILi3ELi6ELi8ELi2ELi3ELi4E\w+):[^\n]*\n(.*?)\n\.Lfunc_end", text, re.S)
    assert m, "k_bootstrap_ntt_pair not found in the device assembly"
    f64 = len(re.findall(r"^\s*v_\w+_f64", m.group(2), re.M))
    assert f64 == bench.ntt_dp_wave_instr_per_cmux(1024, 3)["loop_static"], f64


def test_xfft_opcount_formula_is_the_models():
    """bench.py prices the split-FFT exact backend with the count scripts/xfft/model.py derives (3,072 per CMUX; beside the mirror's 3,744 and
    the NTT backend's 6,648)."""
    import importlib.util
    import bench
    spec = importlib.util.spec_from_file_location("xfft_model", os.path.join(ROOT, "scripts", "xfft", "model.py"))
    model = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(model)
    per_wave, per_cmux = model.instruction_counts(1024)
    o = bench.xfft_dp_wave_instr_per_cmux(1024, 3)
    assert o["total"] == per_cmux == 3072 and o["per_wave"] == sum(per_wave.values()) == 1536
    per_wave2, per_cmux2 = model.instruction_counts(2048)
    o2 = bench.xfft_dp_wave_instr_per_cmux(2048, 3)
    assert o2["total"] == per_cmux2 == 6912 and o2["per_wave"] == sum(per_wave2.values()) == 1728


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
def test_xfft_opcount_matches_the_isa_of_the_built_kernel(device_asm):
    """k_bootstrap_xpair's step loop is one wave's straight-line step (three forward transforms, twelve row multiply-accumulates, two inverse
    transforms, nothing unrolled twice): its FP64-rate instructions are the formula's per-wave count, and all of them but the digit conversions
    and the additions of the first inverse pass are fused multiply-adds."""
    import bench
    m = re.search(r"\n(_ZN5rtfhe17k_bootstrap_xpairILi3ELi6ELi8ELi2ELi3ELi4E\w+):[^\n]*\n(.*?)\n\.Lfunc_end", device_asm.read_text(), re.S)
    assert m, "k_bootstrap_xpair not found in the device assembly"
    body = m.group(2)
    f64 = len(re.findall(r"^\s*v_\w+_f64", body, re.M))
    fma = len(re.findall(r"^\s*v_fma(?:c)?_f64", body, re.M))
    o = bench.xfft_dp_wave_instr_per_cmux(1024, 3)
    assert f64 == o["per_wave"], (f64, o)
    # not fused: the digit conversions, the 44 plain sums of each inverse transform's first pass (twiddles 1, -i, (+-1 - i)/sqrt 2) and the 32
    # products that open a partial sum (first row of phases M1 and M3: 8 points x 2 products x 2 phases)
    assert f64 - fma == o["cvt"] + 2 * 44 + 32, (f64, fma)
    assert "scratch_" not in body or len(re.findall(r"^\s*scratch_", body, re.M)) <= 8, "the step must not spill"
    # the time-sliced launch of the same backend (five or six gates on a CU's four wave pairs) holds the same step
    mr = re.search(r"\n(_ZN5rtfhe20k_bootstrap_xpair_rr\w+):[^\n]*\n(.*?)\n\.Lfunc_end", device_asm.read_text(), re.S)
    assert mr, "k_bootstrap_xpair_rr not found in the device assembly"
    assert len(re.findall(r"^\s*v_\w+_f64", mr.group(2), re.M)) == f64 and len(re.findall(r"^\s*v_fma(?:c)?_f64", mr.group(2), re.M)) == fma
    assert not re.findall(r"^\s*scratch_", mr.group(2), re.M)


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
def test_xfft_n2048_opcount_matches_the_isa_of_the_built_kernel(device_asm):
    """k_bootstrap_xquad (N = 2048, four waves per gate): one wave's straight-line step holds the formula's 1,728 FP64-rate instructions."""
    import bench
    m = re.search(r"\n(_ZN5rtfhe17k_bootstrap_xquadILi3ELi6ELi8ELi2ELi3ELi2E\w+):[^\n]*\n(.*?)\n\.Lfunc_end", device_asm.read_text(), re.S)
    assert m, "k_bootstrap_xquad not found in the device assembly"
    f64 = len(re.findall(r"^\s*v_\w+_f64", m.group(2), re.M))
    assert f64 == bench.xfft_dp_wave_instr_per_cmux(2048, 3)["per_wave"] == 1728, f64


# ---- round 4: the headline must survive the death of a side leg -------------------------------------------------
FAKE_INNER = r'''
import json, os, sys
assert os.environ.get("RTFHE_BENCH_INNER") == "1"
print("RCCL banner or other stray text")
print(json.dumps({"metric": "m", "value": 1.0, "provisional": ["cpu_baseline", "secondary"]}), flush=True)
print(json.dumps({"metric": "m", "value": 1.0, "cpu_baseline": {"value": 2}, "provisional": ["secondary"]}), flush=True)
MODE
'''


@pytest.mark.parametrize("mode, crash", [("os.abort()", True), ("print(json.dumps({'metric': 'm', 'value': 1.0, 'cpu_baseline': {'value': 2}, 'secondary': {}}), flush=True)", False)])
def test_supervisor_prints_the_last_complete_line_even_if_the_measuring_process_dies(tmp_path, capfd, monkeypatch, mode, crash):
    import bench
    child = tmp_path / "inner.py"
    child.write_text(FAKE_INNER.replace("MODE", mode))
    monkeypatch.delenv("RTFHE_BENCH_STRICT", raising=False)
    rc = bench.supervise([sys.executable, str(child)])
    out = capfd.readouterr().out.strip().split("\n")
    assert rc == 0 and len(out) == 1, out                       # ONE line on stdout; without RTFHE_BENCH_STRICT the exit code stays 0
    line = json.loads(out[0])
    assert line["value"] == 1.0 and line["cpu_baseline"] == {"value": 2} and "provisional" not in line
    if crash:
        assert line["ok"] is False                              # ... but the line itself says that the run is not whole
        assert line["side_leg_crash"]["signal"] == "SIGABRT" and line["side_leg_crash"]["legs_not_completed"] == ["secondary"]
    else:
        assert line["ok"] is True and "side_leg_crash" not in line and line["secondary"] == {}
    # the evidence runs (scripts/gpu_run.sh) set RTFHE_BENCH_STRICT=1: the same line, and a distinct non-zero exit code for the crash
    monkeypatch.setenv("RTFHE_BENCH_STRICT", "1")
    rc = bench.supervise([sys.executable, str(child)])
    out = capfd.readouterr().out.strip().split("\n")
    assert len(out) == 1 and json.loads(out[0])["ok"] is (not crash)
    assert rc == (bench.SIDE_LEG_CRASH_RC if crash else 0) and bench.SIDE_LEG_CRASH_RC not in (0, 1, 2)
    assert "RTFHE_BENCH_STRICT=1" in open(os.path.join(ROOT, "scripts", "gpu_run.sh")).read()


def test_supervisor_without_any_line_reports_the_exit_code(tmp_path, capfd):
    import bench
    child = tmp_path / "inner.py"
    child.write_text("import sys\nsys.exit(7)\n")
    assert bench.supervise([sys.executable, str(child)]) == 7
    assert capfd.readouterr().out == ""


@pytest.fixture(scope="module")
def baseline_inputs():
    """keys, 8 NAND inputs and the oracle's outputs for them (standing in for the GPU's: the child only compares)"""
    import numpy as np
    import ctypes as C
    import rustfhe_amd as R
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    P = R.Params()
    key0, key1, bk, ksk = R.keygen(P, 424242)
    b0 = np.array([0, 0, 1, 1, 1, 0, 1, 0], np.uint8)
    b1 = np.array([0, 1, 0, 1, 1, 1, 0, 0], np.uint8)
    in0, in1 = R.encrypt_bits(P, key0, b0, 1), R.encrypt_bits(P, key0, b1, 2)
    p = orc.Params()
    pl = orc.Plan(p.N)
    bk_f = np.empty(bk.size, np.float64)
    orc.lib().orc_trgsw_to_fft(pl.h, bk.ctypes.data_as(C.POINTER(C.c_uint32)), bk_f.ctypes.data_as(C.POINTER(C.c_double)), bk.size // p.N)
    exp, _ = orc.gate_batch_mt(p, orc.NAND, bk_f, None, ksk, in0, in1, 4)
    return np, P, bk, ksk, in0, in1, exp


def test_cpu_baseline_child_measures_tiled_batches_on_pinned_threads(baseline_inputs, monkeypatch):
    import bench
    np, P, bk, ksk, in0, in1, exp = baseline_inputs
    monkeypatch.setenv("RTFHE_BENCH_CPU_TARGET_S", "0.2")
    child = bench.CpuBaselineChild(np, P, bk, ksk, in0, in1, 6)
    res = child.collect(exp, timeout=600)
    assert "error" not in res, res
    assert res["matches_gpu_bit_exact"] is True and res["value"] > 0 and res["unit"] == "gates/s"
    for leg in (res, res.get("port", res)):
        for run in leg["runs"]:
            assert run["gates_per_thread"] >= 6 and run["gates"] == run["gates_per_thread"] * run["threads_used"] and run["matches_gpu_bit_exact"]
        assert 0 < leg["scaling_efficiency"] <= 1.5
    assert res["host"]["physical_cores"] >= 1
    assert not os.path.exists(child.dir)                        # the scratch directory is gone


def test_cpu_baseline_child_respects_a_cgroup_cpu_quota(baseline_inputs, monkeypatch):
    """The GPU boxes of this pool show 256 hardware threads and grant 16 cores of CPU time (cpu.max): the run of record then uses
    ceil(quota) pinned threads, and its efficiency is measured against the cores it was entitled to."""
    import bench
    np, P, bk, ksk, in0, in1, exp = baseline_inputs
    if len(os.sched_getaffinity(0)) < 4:
        pytest.skip("needs >= 4 CPUs")
    monkeypatch.setenv("RTFHE_BENCH_CPU_TARGET_S", "0.2")
    monkeypatch.setenv("RTFHE_TEST_CPU_QUOTA", "2")
    res = bench.CpuBaselineChild(np, P, bk, ksk, in0, in1, 3).collect(exp, timeout=600)
    assert "error" not in res, res
    leg = res.get("port", res)
    assert [r["threads"] for r in leg["runs"]] == ["cgroup_cpu_quota_cores_one_thread_each", "one_thread_per_physical_core"]
    assert leg["runs"][0]["threads_used"] == 2 and leg["runs"][0]["cores_entitled"] == 2 and leg["runs"][1]["cores_entitled"] == 2
    assert res["host"]["cgroup_cpu_quota_cores"] == 2.0 and res["matches_gpu_bit_exact"] is True


def test_cpu_baseline_child_abort_is_an_error_entry_not_a_lost_line(baseline_inputs, monkeypatch):
    import bench
    np, P, bk, ksk, in0, in1, exp = baseline_inputs
    monkeypatch.setenv("RTFHE_BENCH_TEST_ABORT", "cpu_child")
    res = bench.CpuBaselineChild(np, P, bk, ksk, in0, in1, 6).collect(exp, timeout=600)
    assert res["rc"] == -6 and "SIGABRT" in res["error"]


def test_a_wrong_gpu_output_is_reported_as_a_mismatch(baseline_inputs, monkeypatch):
    import bench
    np, P, bk, ksk, in0, in1, exp = baseline_inputs
    monkeypatch.setenv("RTFHE_BENCH_CPU_TARGET_S", "0.05")
    bad = exp.copy()
    bad[3, 17] ^= 1
    res = bench.CpuBaselineChild(np, P, bk, ksk, in0, in1, 1).collect(bad, timeout=600)
    assert res["matches_gpu_bit_exact"] is False


@pytest.mark.gpu
@pytest.mark.parametrize("where", ["cpu_child", "secondary"])
def test_bench_headline_survives_an_abort_in_a_side_leg(where):
    """... and says so: a CPU-baseline child that dies is an error inside its own leg of a whole line ("ok": true); the measuring process itself
    dying in a side leg is "ok": false + side_leg_crash and, under RTFHE_BENCH_STRICT=1 (what scripts/gpu_run.sh exports), exit code 86."""
    import bench
    env = dict(os.environ, RTFHE_BENCH_TEST_ABORT=where, RTFHE_BENCH_STRICT="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--cpu-gates-per-thread", "1"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == (0 if where == "cpu_child" else bench.SIDE_LEG_CRASH_RC), r.stderr[-2000:]
    lines = [ln for ln in r.stdout.split("\n") if ln.strip()]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["value"] > 0 and line["outputs_decrypt_correctly"] and line["roofline"]["frac"] > 0
    if where == "cpu_child":
        assert "SIGABRT" in line["cpu_baseline"]["error"] and "secondary" in line and line["ok"] is True
    else:
        assert line["ok"] is False
        assert line["side_leg_crash"]["signal"] == "SIGABRT" and line["side_leg_crash"]["legs_not_completed"] == ["secondary"]
        assert "value" in line["cpu_baseline"] or "error" in line["cpu_baseline"]
