#!/usr/bin/env python3
"""bench.py -- HomNAND gates/sec on N MI355X (BASELINE.json metric).

A "step" is one pass of the hot path (pre-step, blind rotate, sample extract, identity key switch) over one batch of
independent NAND gates, inputs and keys already resident in HBM.  One process per GPU; rank 0 prints ONE JSON line.

    python bench.py                          # 1 GPU, BASELINE configs[1]: 1024 gates, one kernel launch per step
    python bench.py --gpus 8                 # spawns 8 ranks itself (RCCL), 1024 gates per GPU per step, weak scaling
    python bench.py --gpus 8 --workload config3   # BASELINE configs[2]: 8192 gates per GPU, the whole batch starts on
                                                  # rank 0's GPU: scatter over xGMI -> bootstrap -> gather, all timed
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W      # the same ranks under an external launcher

With --gpus N > 1 and no WORLD_SIZE in the environment this process only LAUNCHES: before torch is imported or any
HIP call is made it starts N fresh children (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set), relays rank 0's JSON line
and exits non-zero if any child did.  RTFHE_BENCH_BACKEND=gloo rehearses the N > 1 path on a box with fewer GPUs
(ranks share the cards; collectives run on the CPU).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# ---- SURVEY 8(d) / BASELINE.md 4: algorithmic bytes one gate must consume at N=1024, n=635, l=3 (the streaming model) ----
BK_BYTES_PER_GATE = 635 * 2 * 6 * 1024 * 8            # 62,423,040  whole bootstrapping key once
KSK_BYTES_PER_GATE = 1024 * 8 * 3 // 4 * 2544          # 15,630,336  expected touched key-switch rows
IO_BYTES_PER_GATE = 3 * 2544                           #      7,632  two inputs + one output
ALG_BYTES_PER_GATE = BK_BYTES_PER_GATE + KSK_BYTES_PER_GATE + IO_BYTES_PER_GATE   # 78,061,008
HBM_PEAK = 8.0e12                                      # MI355X_MICROARCH.md: 8 TB/s spec

# ---- the ceiling that binds: FP64 vector issue without FMA (MI355X_MICROARCH.md chip table) ----
CUS, SIMDS_PER_CU, DP_LANES_PER_CLK, CLOCK_HZ = 256, 4, 16, 2.4e9
FP64_VALU_PEAK = CUS * SIMDS_PER_CU * DP_LANES_PER_CLK * CLOCK_HZ     # 39.32e12 lane-ops/s: one v_add/v_mul_f64 lane result each


def dp_wave_instr_per_cmux(N=1024, l=3):
    """FP64-rate VALU wave-instructions one CMUX step costs, derived from the transform structure of
    rustfhe_amd/csrc/rtfhe_device.hpp (checked against the built kernel's ISA by tests/test_bench_launcher.py):
    the mirror arithmetic may not fuse, so every product and every sum is one v_mul_f64 / v_add_f64.
    `reference` is the reference's operation list; the kernels execute `total` = reference minus the instructions that cannot change a
    torus word: the 6 multiplies / sums of the one butterfly per transform whose twiddle is exactly (1, 0) (N = 1024: the halfnn = 4
    stage of every 512-point transform; N = 2048 (k_bootstrap_eo, transforms split over two waves by the parity of the point index):
    three in the even-point network -- one in its halfnn = 8 stage, two in halfnn = 4) and, in
    the two-waves-per-gate kernel at N = 1024, the "+0.0 +" of the first row of component 0's fold."""
    P = N // 2
    R = P // 64                               # points per lane
    LR = R.bit_length() - 1
    low = (P.bit_length() - 1) - 2 * LR       # radix-2 stages left for pass 3
    twist = 6 * R                             # 4 products + 2 sums per point
    pass12 = 2 * LR * (R // 2) * 10           # two passes of LR twiddled stages: 4 sums + 4 products + 2 sums per butterfly
    pass3_tw = max(low - 2, 0) * (R // 2) * 10
    size4_size2 = 2 * R + 2 * R               # halfnn = 2 and 1: sums only
    transform = twist + pass12 + pass3_tw + size4_size2           # 360 at N = 1024, forward and inverse alike
    mac = 8 * R                               # hadamard 4 products + 2 sums, fold 2 sums, per point
    trunc_add = 2 * R                         # the magic-constant add of trunc_to_torus, per inverse transform
    arith = 2 * l * transform + 2 * transform + 2 * 2 * l * mac + 2 * trunc_add
    cvt = 2 * l * 2 * R + 2 * 2 * R           # v_cvt_f64_i32 per digit, v_trunc_f64 per output word
    transforms = 2 * l + 2
    unit_per_transform = 6 * (1 if N == 1024 else 3)      # per unit-twiddle butterfly: 4 products + 2 sums
    unit = unit_per_transform * transforms
    first_row = 2 * R if N == 1024 else 0                         # slot P of side 0 only
    return {"add_mul": arith - unit - first_row, "cvt_trunc": cvt, "total": arith - unit - first_row + cvt, "reference": arith + cvt,
            "transform": transform, "unit_twiddle_saved_per_transform": unit_per_transform, "first_row_saved": first_row, "mac_row": mac}


def ntt_dp_wave_instr_per_cmux(N=1024, l=3):
    """FP64-rate VALU wave-instructions of one CMUX step on the exact-integer NTT backend (rustfhe_amd/csrc/rtfhe_ntt.hpp,
    rtfhe_kernels_ntt.hpp, rtfhe_kernels_ntt_halves.hpp; checked against the built kernels' ISA by tests/test_bench_launcher.py).
    A modular product is 6 instructions (mul, fma, mul, rndne, fma, add), a renormalisation 3 (mul, rndne, fma); one wave
    transforms 1024 points, 16 per lane: 10 stages x 8 butterflies x (product + 2 sums)."""
    R, stages = 16, 10
    modmul, norm = 6, 3
    lean = N == 1024                                                    # round 5: one renormalisation per forward transform, four per inverse (rtfhe_ntt.hpp)
    fwd = stages * (R // 2) * (modmul + 2) + (1 if lean else 2) * R * norm      # renormalised after stage 7 (N = 2048: after stages 5 and 10)
    inv = stages * (R // 2) * (modmul + 2) + (4 if lean else 5) * R * norm      # on entry and after stages 4, 8, 10 (N = 2048: 3, 6, 9, 10)
    mac = 2 * R * (modmul + 1)                                          # both components of a key row
    if N == 1024:
        # the first TWO stages act on decomposition digits and read every digit x twiddle product from 64-entry tables in LDS: per four points
        # one conversion and eight sums (rtfhe_ntt.hpp, first_two_stages_digits); stages 3..10 are butterflies, one renormalisation
        fwd_digits = (R // 4) * (1 + 8) + (stages - 2) * (R // 2) * (modmul + 2) + R * norm
        # two waves per gate, each: l rows (two table stages, transform, products), the swapped component's add, one inverse, magic add
        row = fwd_digits + mac
        wave = l * row + R + inv + R
        return {"total": 2 * wave, "per_wave": wave, "loop_static": row + R + inv + R, "forward": fwd_digits, "inverse": inv, "mac_row": mac}
    if N == 2048:
        # two waves per transform: per polynomial and row both halves' digits (2R cvt), the stage across the halves (R products +
        # R sums), the half's transform and products; four inverse transforms per step (b-rows and a-rows separately), each with the
        # last stage across the halves: wave 0 R sums + R renormalisations, wave 1 R differences + R products + R renormalisations
        row = R + R + fwd + mac                                         # R conversions, R sums of the stage across the halves (table)
        cross0, cross1 = R + R * norm + R, R + R * modmul + R * norm + R
        total = 2 * (2 * l * row) + 4 * (2 * inv + cross0 + cross1)
        return {"total": total, "row": row, "forward": fwd, "inverse": inv, "mac_row": mac}
    raise ValueError("N")


def kernel_src_hash():
    """Identifies the device code a PMC profile was taken with (profiles/pmc_traffic.json is refused when it differs): every kernel lives
    in a header under rustfhe_amd/csrc/ (*.hpp); the .hip / .cpp files there are host code (C ABI, key generation, files)."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "rustfhe_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith(".hpp"):
            h.update(f.encode())
            with open(os.path.join(d, f), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


# ------------------------------------------------------------------------------------------------------------------
# launcher (parent process: never imports torch, never touches HIP)
# ------------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(nranks, child_cmd, timeout=None, env_extra=None):
    """Starts `nranks` fresh processes of child_cmd with the torch.distributed environment set, relays rank 0's stdout
    to ours, and returns (exit code, rank-0 stdout lines).  Exit code is non-zero if ANY rank failed; on the first
    failure the other ranks are terminated (each is its own process, killed by exact PID)."""
    port = _free_port()
    procs = []
    for r in range(nranks):
        env = dict(os.environ)
        # HSA_ENABLE_IPC_MODE_LEGACY=0: the host driver of this pool only supports dmabuf IPC; with the legacy mode RCCL's (and
        # torch's) cross-process device-memory sharing fails with "hipIpcGetMemHandle: invalid argument".  The image exports it
        # already; it is passed on explicitly so that a launcher with a scrubbed environment still gets working xGMI P2P.
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nranks), LOCAL_WORLD_SIZE=str(nranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        env.update(env_extra or {})
        procs.append(subprocess.Popen(list(child_cmd), env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=(r == 0)))
    import threading
    lines, rc, t0 = [], 0, time.time()

    def relay():                               # rank 0 prints the one JSON line; everything it prints is relayed
        for line in procs[0].stdout:
            lines.append(line)
            sys.stdout.write(line)
            sys.stdout.flush()
    reader = threading.Thread(target=relay, daemon=True)
    reader.start()
    try:
        pending = list(range(nranks))
        while pending:
            for r in list(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.remove(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    sys.stderr.write("bench.py: rank %d exited with %d\n" % (r, code))
                    for q in pending:
                        procs[q].terminate()
            if timeout and time.time() - t0 > timeout and pending:
                rc = rc or 124
                for q in pending:
                    procs[q].terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        reader.join(timeout=10)
    return rc, lines


def launch(args):
    rc, lines = spawn_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:])
    if rc == 0:
        ok = False
        for ln in lines:
            try:
                ok = ok or json.loads(ln).get("n_gpus") == args.gpus
            except ValueError:
                pass
        if not ok:
            sys.stderr.write("bench.py: no JSON line with n_gpus == %d came back from rank 0\n" % args.gpus)
            rc = 1
    return rc


# ------------------------------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N = 1 only): the oracle / the reference's own FFT on the host cores.  Checker, never product.
# ------------------------------------------------------------------------------------------------------------------
def under_profiler():
    """rocprofv3 preloads its tool library into the program it starts, and that library may initialise the GPU before main(): a process
    in that state must not start other programs (this pool refuses the exec).  Under the profiler bench.py therefore measures in THIS process:
    no supervisor, no CPU-baseline child (the baseline is not what a profile is taken for)."""
    return "rocprofiler" in os.environ.get("LD_PRELOAD", "") or bool(os.environ.get("ROCPROFILER_LIBRARY_CTOR"))


def _signal_name(rc):
    import signal
    if rc is None or rc >= 0:
        return None
    try:
        return signal.Signals(-rc).name
    except ValueError:
        return "signal %d" % -rc


def cpu_baseline_child(tmpdir, per_thread):
    """The CPU baseline leg, run as a FRESH PROCESS that never loads the HIP library or torch: a native crash in the oracle / the
    reference-built spqlios (an instruction this host lacks, a fault) costs this process, not the bench line.  Protocol: the parent
    (rank 0, before its first GPU call) writes params / keys / inputs to `tmpdir` and starts this; it sets itself up (loads the
    oracle, transforms the key) and blocks on stdin; the parent, once its timed region is over, writes gpu_out.npy and sends "go";
    this measures and prints ONE JSON object on stdout.  EOF instead of "go" = exit quietly.

    What is measured: first the single-thread rate on an otherwise idle host (16 gates, thread pinned); then
    multi-thread runs in which EVERY thread gets `per_thread` gates or more (the batch's 1024 inputs are tiled: gate g takes input
    g % 1024 and is compared with GPU output g % 1024), threads pinned, the two 62 MB keys replicated per memory node and
    first-touched by a thread of that node (oracle/tfhe_oracle.c: orc_gate_batch_mt_numa).  The run of record uses as many threads
    as the host lets this process have: one per physical core -- or, when the cgroup's CPU quota (cpu.max) is fewer cores than
    that, ceil(quota) threads spread over the machine; a short second run with the other thread set stands beside it.  The best is
    `value`; scaling_efficiency = value / (cores the run was entitled to x the single-thread rate); `cores` = cores of that run."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C
    import math
    import numpy as np
    import orc
    z = np.load(os.path.join(tmpdir, "inputs.npz"))
    pr = [int(v) for v in z["params"]]
    p = orc.Params(n=pr[0], N=pr[1], l=pr[2], bgbit=pr[3], ks_t=pr[4], ks_basebit=pr[5])
    bk, ksk, in0, in1 = z["bk"], z["ksk"], z["in0"], z["in1"]
    pl = orc.Plan(p.N)
    bk_f = np.empty(bk.size, np.float64)
    orc.lib().orc_trgsw_to_fft(pl.h, bk.ctypes.data_as(C.POINTER(C.c_uint32)), bk_f.ctypes.data_as(C.POINTER(C.c_double)), bk.size // p.N)
    del bk
    topo = orc.host_topology()
    have_ref = orc.have_ref()
    if have_ref:
        orc.use_reference_fft_in_mt()
    if not sys.stdin.readline().startswith("go"):
        return 0
    if os.environ.get("RTFHE_BENCH_TEST_ABORT") == "cpu_child":       # tests: the headline must survive this
        os.abort()
    gpu_out = np.load(os.path.join(tmpdir, "gpu_out.npy"))
    in_count = in0.shape[0]
    per_core, per_hw = topo["one_thread_per_core"], topo["hw_threads"]
    quota = topo["cgroup_cpu_quota"]
    single = min(in_count, 16)
    target_s = float(os.environ.get("RTFHE_BENCH_CPU_TARGET_S", "2.5"))      # seconds an all-core run lasts at perfect scaling (tests shorten it)

    def leg(backend):
        # single thread first, on an otherwise idle host (after an all-core run the package is still clocked down)
        # (twice, the better of the two: 16 gates take ~0.2 s and one disturbance by another tenant of the host would otherwise set the yardstick)
        out1, s1 = orc.gate_batch_mt_numa(p, orc.NAND, bk_f, ksk, in0[:single], in1[:single], single, per_core[:1], topo["node_of"], backend=backend)
        s1 = min(s1, orc.gate_batch_mt_numa(p, orc.NAND, bk_f, ksk, in0[:single], in1[:single], single, per_core[:1], topo["node_of"], backend=backend)[1])
        rate1 = single / s1
        ok = bool(np.array_equal(out1, gpu_out[:single]))
        # Which thread sets: a container may see every CPU of the machine and still be entitled to a few cores' worth of CPU time (cgroup
        # cpu.max; the GPU boxes of this pool: 256 hardware threads, quota 16 cores).  More busy threads than the quota only get throttled,
        # so there the run of record uses ceil(quota) threads, pinned to cores spread evenly over the machine (one per L3 / memory node where
        # possible), and the one-thread-per-core run is kept short, as the witness of the throttling.  Without a quota: one thread per
        # physical core is the run of record, one per hardware thread the short second opinion.
        limited = bool(quota) and quota < len(per_core)
        if limited:
            q = max(1, int(math.ceil(quota)))
            spread = [per_core[(j * len(per_core)) // q] for j in range(q)]
            sets = [("cgroup_cpu_quota_cores_one_thread_each", spread, True, True), ("one_thread_per_physical_core", per_core, True, False)]
        else:
            sets = [("one_thread_per_physical_core", per_core, True, True)]
            if len(per_hw) > len(per_core):
                sets.append(("one_thread_per_hardware_thread", per_hw, True, False))
        runs = []
        for name, cpus, pin, primary in sets:
            # every thread gets at least per_thread gates; the run of record enough for ~target_s seconds at perfect scaling
            k = int(min(512, max(per_thread, math.ceil(target_s * rate1)))) if primary else int(per_thread)
            count = k * len(cpus)
            out, secs = orc.gate_batch_mt_numa(p, orc.NAND, bk_f, ksk, in0, in1, count, cpus, topo["node_of"], backend=backend, pin=pin)
            same = bool(np.array_equal(out, gpu_out[np.arange(count) % in_count]))
            ok = ok and same
            ncores = len(set(topo["core_of"][c] for c in cpus))       # physical cores the threads occupy
            entitled = min(ncores, quota) if quota else ncores         # ... and the CPU time the cgroup lets them have
            runs.append({"threads": name, "threads_used": len(cpus), "pinned": pin, "gates": count, "gates_per_thread": k, "seconds": round(secs, 3),
                         "gates_per_s": round(count / secs, 1), "cores_busy": ncores, "cores_entitled": entitled,
                         "scaling_efficiency": round(count / secs / (entitled * rate1), 3), "matches_gpu_bit_exact": same})
            del out
        best = max(runs, key=lambda r: r["gates_per_s"])
        return rate1, runs, best, ok

    timing = "timed from the moment every thread holds its FFT plan and every memory node its key replica to the last thread's last gate"
    nodes = len(topo["nodes"])
    host = {"cpu_model": topo["cpu_model"], "physical_cores": len(per_core), "hw_threads": len(per_hw), "memory_nodes": nodes,
            "cgroup_cpu_quota_cores": quota}
    rate1, runs, best, ok = leg(orc.BACKEND_MIRROR)
    port = {"value": best["gates_per_s"], "unit": "gates/s", "cores": best["cores_busy"], "hw_threads": len(per_hw), "threads_used": best["threads_used"],
            "kind": "port", "single_thread_ms_per_gate": round(1e3 / rate1, 2), "single_thread_sample_gates": single,
            "scaling_efficiency": best["scaling_efficiency"], "runs": runs, "matches_gpu_bit_exact": ok,
            "sample": "%d NAND gates (%d per thread on %d pinned threads, the batch's %d inputs tiled, every output compared with the GPU's), "
                      "oracle/tfhe_oracle.c FP64 mirror of the reference spqlios FFT, gcc -O3 -march=x86-64-v3 -ffp-contract=off, keys replicated "
                      "on each of %d memory node(s); %s" % (best["gates"], best["gates_per_thread"], best["threads_used"], in_count, nodes, timing)}
    res = port
    if have_ref:
        # the reference's OWN compiled native FFT (oracle/_ref: utils/src/spqlios/*.cpp + AVX .s) under the restated Rust glue
        # (the Rust half cannot be built: no toolchain); one Spqlios handle per thread, as its thread_local FFT_MAP would give
        rate1r, runs_r, best_r, ok_r = leg(orc.BACKEND_HOOK)
        res = {"value": best_r["gates_per_s"], "unit": "gates/s", "cores": best_r["cores_busy"], "hw_threads": len(per_hw), "threads_used": best_r["threads_used"],
               "kind": "reference", "single_thread_ms_per_gate": round(1e3 / rate1r, 2), "single_thread_sample_gates": single,
               "scaling_efficiency": best_r["scaling_efficiency"], "runs": runs_r, "matches_gpu_bit_exact": ok_r,
               "sample": "%d NAND gates (%d per thread on %d pinned threads, the batch's %d inputs tiled, every output compared with the GPU's): the "
                         "reference's own compiled spqlios AVX FFT (oracle/_ref, flags of utils/build.rs with -march pinned to x86-64-v3), one handle "
                         "per thread, under the C restatement of its Rust glue, keys replicated on each of %d memory node(s); %s"
                         % (best_r["gates"], best_r["gates_per_thread"], best_r["threads_used"], in_count, nodes, timing),
               "port": port}
        rate1, best = rate1r, best_r
    res["host"] = host
    print(json.dumps(res), flush=True)
    return 0


class CpuBaselineChild:
    """Parent side of cpu_baseline_child: start() BEFORE the first GPU call, collect() after the timed region."""

    def __init__(self, np, params, bk, ksk, in0, in1, per_thread):
        import tempfile
        self.np = np
        self.dir = tempfile.mkdtemp(prefix="rtfhe_bench_")
        np.savez(os.path.join(self.dir, "inputs.npz"), params=np.array([params.n, params.N, params.l, params.bgbit, params.ks_t, params.ks_basebit], np.int64),
                 bk=bk, ksk=ksk, in0=in0, in1=in1)
        self.proc = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", self.dir, "--cpu-gates-per-thread", str(per_thread)],
                                     stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)

    def collect(self, gpu_out, timeout=420):
        try:
            self.np.save(os.path.join(self.dir, "gpu_out.npy"), gpu_out)
            try:
                out, _ = self.proc.communicate("go\n", timeout=timeout)
            except subprocess.TimeoutExpired:
                self.proc.kill()
                self.proc.communicate()
                return {"error": "the CPU baseline child did not finish within %d s and was killed" % timeout}
            rc = self.proc.returncode
            if rc != 0:
                return {"error": "the CPU baseline child exited with %s%s" % (rc, " (%s)" % _signal_name(rc) if rc < 0 else ""), "rc": rc}
            for ln in reversed(out.strip().split("\n")):
                try:
                    return json.loads(ln)
                except ValueError:
                    continue
            return {"error": "the CPU baseline child printed no JSON"}
        finally:
            self.close()

    def close(self):
        import shutil
        if self.proc.poll() is None:
            try:
                self.proc.stdin.close()         # EOF instead of "go": the child exits quietly
                self.proc.wait(timeout=20)
            except Exception:                   # noqa: BLE001
                self.proc.kill()
        shutil.rmtree(self.dir, ignore_errors=True)


# ------------------------------------------------------------------------------------------------------------------
# supervisor (N = 1, plain `python bench.py`): the measuring process runs as a child; the headline survives its death
# ------------------------------------------------------------------------------------------------------------------
SIDE_LEG_CRASH_RC = 86


def supervise(child_cmd, env=None):
    """Runs the measuring process as a child with RTFHE_BENCH_INNER=1 and prints exactly ONE JSON line: the LAST complete line the
    child produced.  The child prints its line as soon as the headline is complete (marked "provisional") and again after each side
    leg (CPU baseline, secondary measurements); if it then dies -- a native crash in a side kernel, an abort() in a library -- the
    headline measured before is printed with the crash recorded in it: top-level "ok": false and "side_leg_crash" (a complete run carries
    "ok": true).  The exit code then is SIDE_LEG_CRASH_RC (86) when RTFHE_BENCH_STRICT=1 is set -- scripts/gpu_run.sh sets it, so a GPU fault in
    a side kernel fails the evidence run -- and 0 otherwise, so that a driver that gates on the exit code alone does not lose the headline it
    came for (the line says what happened either way).  No line at all = the child's exit code."""
    e = dict(os.environ if env is None else env)
    e["RTFHE_BENCH_INNER"] = "1"
    proc = subprocess.Popen(list(child_cmd), stdout=subprocess.PIPE, text=True, env=e)
    last = None
    for ln in proc.stdout:
        try:
            j = json.loads(ln)
        except ValueError:
            sys.stderr.write(ln)             # stray text never reaches our stdout
            continue
        if isinstance(j, dict) and "metric" in j:
            last = j
    rc = proc.wait()
    if last is None:
        sys.stderr.write("bench.py: the measuring process exited with %s and printed no line\n" % rc)
        return rc if rc > 0 else 1
    pending = last.pop("provisional", None)
    if rc != 0:
        strict = os.environ.get("RTFHE_BENCH_STRICT") == "1"
        sys.stderr.write("bench.py: THE MEASURING PROCESS DIED (rc %s, %s) after the headline was complete; legs not completed: %s -- the line "
                         "below carries \"ok\": false and side_leg_crash; exit code %d (RTFHE_BENCH_STRICT=1 makes it %d)\n"
                         % (rc, _signal_name(rc), pending, SIDE_LEG_CRASH_RC if strict else 0, SIDE_LEG_CRASH_RC))
        last["ok"] = False
        last["side_leg_crash"] = {"rc": rc, "signal": _signal_name(rc), "legs_not_completed": pending,
                                  "note": "the measuring process died AFTER the headline above was complete; the value, roofline and every leg "
                                          "present in this line were measured before that"}
        print(json.dumps(last), flush=True)
        return SIDE_LEG_CRASH_RC if strict else 0
    last["ok"] = True
    print(json.dumps(last), flush=True)
    return 0


def xfft_dp_wave_instr_per_cmux(N=1024, l=3):
    """FP64-rate VALU wave-instructions of one CMUX step on the split-FFT exact backend (rustfhe_amd/csrc/rtfhe_xfft.hpp,
    rtfhe_kernels_xfft.hpp; the same count as scripts/xfft/model.py, checked against the built kernel's ISA by tests/test_bench_launcher.py).
    Every one of them but the conversions is a v_fma_f64 / v_fmac_f64 (or an addition of the first inverse pass): two flops each against the
    78.6 TFLOP/s FMA peak is the same fraction as one instruction each against the 39.3 T lane-instructions/s issue ceiling."""
    assert N in (1024, 2048) and l == 3
    R = 8
    fwd = 3 * 12 * 6                                               # three passes of 12 radix-2 butterflies, 6 FMAs each; the twist is in the twiddles
    inv = (4 * 4 + 4 * 4 + 2 * 4 + 2 * 6) + 2 * 12 * 6             # first pass on twiddles 1, -i, (+-1 - i)/sqrt 2; two full passes
    if N == 2048:
        # four waves per gate (polynomial x half of the spectrum, rtfhe_kernels_xfft2.hpp): the digits and their stage-1 sums / differences
        # converted (4 per point), stage 1 across the halves (2 FMAs per point), the half's 512-point transforms, the same twelve row
        # multiply-accumulates, nine inverse stages, the last stage fused with untwist and rounding (20 per pair of outputs, 4 pairs, hi and lo)
        per_wave = {"cvt": l * R * 4, "stage1": l * R * 2, "forward": l * fwd, "mac": 4 * l * R * 4, "inverse": 2 * inv, "last_stage_round": 2 * 4 * 20}
        return {**per_wave, "per_wave": sum(per_wave.values()), "total": 4 * sum(per_wave.values())}
    per_wave = {"cvt": l * 2 * R, "forward": l * fwd, "mac": 4 * l * R * 4, "inverse": 2 * inv, "untwist_round": 2 * 2 * R * 2}
    return {**per_wave, "per_wave": sum(per_wave.values()), "total": 2 * sum(per_wave.values())}


def fp64_frac(dp_per_cmux, n, gates, seconds):
    """fraction of the FP64 vector-issue ceiling a launch of `gates` gates taking `seconds` reaches"""
    return round(dp_per_cmux * n * 64 * gates / seconds / FP64_VALU_PEAK, 4)


def secondary_measurements(R, params, eng, key0, bk, ksk, gpu, stream, torch, np):
    """The other BASELINE configs and the NTT backend, measured in the same run on the same GPU (rank 0, N = 1): each entry times
    its launches with HIP events on the launch stream, names the kernel that dominates it and that kernel's fraction of the FP64
    vector-issue ceiling.  Correctness of every entry is checked by decryption here and bit for bit by tests/ (-m gpu)."""
    from rustfhe_amd.circuit import CircuitRunner, prefix_adder, ripple_carry_adder
    from rustfhe_amd.shard import ShardedGates, engine_compute
    sec = {}
    ops = dp_wave_instr_per_cmux(params.N, params.l)["total"]
    rng = np.random.default_rng(77)

    def timed(e, fn, reps):
        fn(); fn(); e.sync(stream)      # two untimed launches: the first ones after a change of kernel run at a lower clock
        e.timer_begin(stream)
        for _ in range(reps):
            fn()
        ms, ks_ms, n = e.timer_end_detail(stream)
        return ms / reps, ks_ms / reps

    def guard(name, fn):             # one failing entry is recorded as such and does not cost the others
        try:
            fn()
        except Exception as e:       # noqa: BLE001
            sec[name] = {"error": "%s: %s" % (type(e).__name__, e)}
            try:
                eng.set_backend(R._ffi.BACKEND_FFT64_MIRROR)
            except Exception:        # noqa: BLE001
                pass

    # -- BASELINE configs[0]: a single HomNAND gate (latency shape: one gate per 8-wave workgroup)
    def _config1_single_gate():
        b = rng.integers(0, 2, (2, 8)).astype(np.uint8)
        c0 = torch.from_numpy(R.encrypt_bits(params, key0, b[0], 11).view(np.int32)).to(gpu)
        c1 = torch.from_numpy(R.encrypt_bits(params, key0, b[1], 12).view(np.int32)).to(gpu)
        o = torch.empty_like(c0)
        ms, _ = timed(eng, lambda: eng.gate_batch_dev(R.NAND, c0, c1, o, 1, stream), 5)
        sec["config1_single_gate"] = {"ms_per_gate": round(ms, 4), "kernel": "k_bootstrap_wg", "roofline_frac_fp64": fp64_frac(ops, params.n, 1, ms * 1e-3),
                                      "note": "one gate on one CU: the fraction is of the WHOLE chip's ceiling (1/256 of it is reachable)"}
    guard("config1_single_gate", _config1_single_gate)
    # -- BASELINE configs[2] on one GPU: 8192 gates held on the device, scatter -> bootstrap -> gather inside the timed step
    def _config3_8192_gates_one_gpu():
        G3 = 8192
        bb = rng.integers(0, 2, (2, G3)).astype(np.uint8)
        f0 = torch.from_numpy(R.encrypt_bits(params, key0, bb[0], 21).view(np.int32)).to(gpu)
        f1 = torch.from_numpy(R.encrypt_bits(params, key0, bb[1], 22).view(np.int32)).to(gpu)
        sg = ShardedGates(engine_compute(eng, gpu), params.n + 1, gpu)
        sg.run(R.NAND, f0, f1, G3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.timer_begin(stream)
        res = sg.run(R.NAND, f0, f1, G3, sync=torch.cuda.synchronize)
        ms3, ks3, n3 = eng.timer_end_detail(stream)
        torch.cuda.synchronize()
        wall3 = time.perf_counter() - t0
        ok3 = bool(np.array_equal(R.decrypt_bits(params, key0, res.cpu().numpy().view(np.uint32)), 1 - (bb[0] & bb[1])))
        sec["config3_8192_gates_one_gpu"] = {"gates_per_s": round(G3 / wall3, 1), "ms_per_step": round(wall3 * 1e3, 3),
                                             "phase_ms": {k: round(1e3 * v, 3) for k, v in sg.last_timing.items()},
                                             "kernel": "k_bootstrap_pair", "kernel_ms": round(ms3 - ks3, 3), "key_switch_kernel_ms": round(ks3, 3),
                                             "roofline_frac_fp64": fp64_frac(ops, params.n, G3, (ms3 - ks3) * 1e-3), "ok": ok3}
        del f0, f1, res
    guard("config3_8192_gates_one_gpu", _config3_8192_gates_one_gpu)
    # -- BASELINE configs[3]: the 8-bit adder as a NAND netlist through the front-end, one replica, whole netlist = one HIP graph
    def _config4_adder_8bit_one_replica():
        adders = {}
        for name, net in (("ripple_carry_nand_only", ripple_carry_adder(8, True)), ("ripple_carry_xor_and_or", ripple_carry_adder(8, False)),
                          ("parallel_prefix_nand_only", prefix_adder(8, True)), ("parallel_prefix_xor_and_or", prefix_adder(8, False))):
            A, B = int(rng.integers(0, 256)), int(rng.integers(0, 256))
            bits = np.array([(A >> i) & 1 for i in range(8)] + [(B >> i) & 1 for i in range(8)], np.uint8)
            run = CircuitRunner(eng, net, 1)
            run.set_inputs(R.encrypt_bits(params, key0, bits, 31).reshape(1, 16, params.n + 1))
            run.run()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run.run()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            dec = R.decrypt_bits(params, key0, run.outputs().reshape(-1, params.n + 1))
            d = net.describe()
            adders[name] = {"ms_per_addition": round(dt * 1e3, 3), "gates": d["gates"], "levels": d["depth"], "kernel": "k_bootstrap_wg",
                            "roofline_frac_fp64": fp64_frac(ops, params.n, d["gates"], dt), "ok": int((dec * (1 << np.arange(9))).sum()) == A + B}
            run.close()
        sec["config4_adder_8bit_one_replica"] = adders
    guard("config4_adder_8bit_one_replica", _config4_adder_8bit_one_replica)
    # -- ... and its throughput form: 1024 replicas of each netlist side by side (a wave of the netlist = 1024 x its gates of that level).
    #    "config 4 as named" (BASELINE configs[3]) is the ripple-carry NAND-only netlist: 72 gates, 20 levels.
    def _config4_adder_8bit_1024_replicas():
        reps, adders = 1024, {}
        for name, net in (("ripple_carry_nand_only", ripple_carry_adder(8, True)), ("ripple_carry_xor_and_or", ripple_carry_adder(8, False)),
                          ("parallel_prefix_nand_only", prefix_adder(8, True)), ("parallel_prefix_xor_and_or", prefix_adder(8, False))):
            A, B = rng.integers(0, 256, reps), rng.integers(0, 256, reps)
            bits = np.array([[(a >> i) & 1 for i in range(8)] + [(b >> i) & 1 for i in range(8)] for a, b in zip(A, B)], np.uint8)
            run = CircuitRunner(eng, net, reps)
            run.set_inputs(R.encrypt_bits(params, key0, bits.reshape(-1), 33).reshape(reps, 16, params.n + 1))
            run.run()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run.run()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            dec = R.decrypt_bits(params, key0, run.outputs().reshape(-1, params.n + 1)).reshape(reps, 9)
            d = net.describe()
            adders[name] = {"ms_per_1024_additions": round(dt * 1e3, 2), "additions_per_s": round(reps / dt, 1), "gates_per_s": round(reps * d["gates"] / dt, 1),
                            "gates": d["gates"], "levels": d["depth"], "kernel": "k_bootstrap_pair",
                            "roofline_frac_fp64": fp64_frac(ops, params.n, reps * d["gates"], dt),
                            "ok": bool(np.array_equal((dec * (1 << np.arange(9))).sum(axis=1), A + B))}
            run.close()
        sec["config4_adder_8bit_1024_replicas"] = {"config4_as_named": "ripple_carry_nand_only", **adders}
    guard("config4_adder_8bit_1024_replicas", _config4_adder_8bit_1024_replicas)
    # -- BASELINE configs[2] behind the C ABI: a batch resident on one device sharded by a multi-device CONTEXT (rtfhe_gate_batch_dev on
    #    rtfhe_ctx_create_multi; no Python, no process per GPU in the data path).  On this one-GPU run the context names the GPU twice: the
    #    number prices the sharding machinery (peer copies, events, two streams sharing the card), not a second GPU.
    def _config3_c_abi_two_contexts_one_gpu():
        G3 = 8192
        bb = rng.integers(0, 2, (2, G3)).astype(np.uint8)
        f0 = torch.from_numpy(R.encrypt_bits(params, key0, bb[0], 23).view(np.int32)).to(gpu)
        f1 = torch.from_numpy(R.encrypt_bits(params, key0, bb[1], 24).view(np.int32)).to(gpu)
        fo = torch.empty_like(f0)
        m = R.Engine(params, devices=[gpu.index, gpu.index])
        try:
            m.load_bk_torus(bk)
            m.load_ksk(ksk)
            m.gate_batch_dev(R.NAND, f0, f1, fo, G3, stream); m.sync(stream)
            t0 = time.perf_counter()
            m.gate_batch_dev(R.NAND, f0, f1, fo, G3, stream); m.sync(stream)
            wall = time.perf_counter() - t0
            okm = bool(np.array_equal(R.decrypt_bits(params, key0, fo.cpu().numpy().view(np.uint32)), 1 - (bb[0] & bb[1])))
            sec["config3_c_abi_two_contexts_one_gpu"] = {"gates_per_s": round(G3 / wall, 1), "ms_per_step": round(wall * 1e3, 3), "entries": 2,
                                                        "device_bytes_per_entry": [m.memory_bytes(0), m.memory_bytes(1)], "ok": okm}
        finally:
            m.close()
    guard("config3_c_abi_two_contexts_one_gpu", _config3_c_abi_two_contexts_one_gpu)
    # -- the NTT backend north_star names, 1024 gates (exact-integer arithmetic; decrypt-level parity with the reference)
    def _ntt_exact_1024_gates():
        G = 1024
        bb = rng.integers(0, 2, (2, G)).astype(np.uint8)
        d0 = torch.from_numpy(R.encrypt_bits(params, key0, bb[0], 41).view(np.int32)).to(gpu)
        d1 = torch.from_numpy(R.encrypt_bits(params, key0, bb[1], 42).view(np.int32)).to(gpu)
        do = torch.empty_like(d0)
        eng.set_backend(R._ffi.BACKEND_NTT_EXACT)
        ms, _ = timed(eng, lambda: eng.gate_batch_dev(R.NAND, d0, d1, do, G, stream), 5)
        okn = bool(np.array_equal(R.decrypt_bits(params, key0, do.cpu().numpy().view(np.uint32)), 1 - (bb[0] & bb[1])))
        eng.set_backend(R._ffi.BACKEND_FFT64_MIRROR)
        nops = ntt_dp_wave_instr_per_cmux(params.N, params.l)["total"]
        sec["ntt_exact_1024_gates"] = {"gates_per_s": round(G / ms * 1e3, 1), "ms_per_launch": round(ms, 3), "kernel": "k_bootstrap_ntt_pair",
                                       "roofline_frac_fp64": fp64_frac(nops, params.n, G, ms * 1e-3), "ok": okn}
    guard("ntt_exact_1024_gates", _ntt_exact_1024_gates)
    # -- the same exact products through the split FFT (RTFHE_BACKEND_FFT_SPLIT_EXACT): 1024 gates, every output word equal to the NTT backend's
    def _xfft_exact_1024_gates():
        G = 1024
        bb = rng.integers(0, 2, (2, G)).astype(np.uint8)
        d0 = torch.from_numpy(R.encrypt_bits(params, key0, bb[0], 41).view(np.int32)).to(gpu)
        d1 = torch.from_numpy(R.encrypt_bits(params, key0, bb[1], 42).view(np.int32)).to(gpu)
        do, dn = torch.empty_like(d0), torch.empty_like(d0)
        eng.set_backend(R._ffi.BACKEND_NTT_EXACT)
        eng.gate_batch_dev(R.NAND, d0, d1, dn, G, stream); eng.sync(stream)
        eng.set_backend(R._ffi.BACKEND_FFT_SPLIT_EXACT)
        ms, _ = timed(eng, lambda: eng.gate_batch_dev(R.NAND, d0, d1, do, G, stream), 5)
        okx = bool(np.array_equal(R.decrypt_bits(params, key0, do.cpu().numpy().view(np.uint32)), 1 - (bb[0] & bb[1])))
        same = bool(torch.equal(do, dn))
        eng.set_backend(R._ffi.BACKEND_FFT64_MIRROR)
        xops = xfft_dp_wave_instr_per_cmux(params.N, params.l)["total"]
        sec["xfft_exact_1024_gates"] = {"gates_per_s": round(G / ms * 1e3, 1), "ms_per_launch": round(ms, 3), "kernel": "k_bootstrap_xpair",
                                        "dp_wave_instr_per_cmux": xops, "roofline_frac_fp64_fma": fp64_frac(xops, params.n, G, ms * 1e-3),
                                        "frac_is": "v_fma_f64 issue: 2 flops each against 78.6 TFLOP/s = 1 instruction each against 39.3 T lane-instructions/s",
                                        "equals_ntt_exact_bit_for_bit": same, "ok": okx}
    guard("xfft_exact_1024_gates", _xfft_exact_1024_gates)
    # -- tail behaviour on the driver's clock: batches that do not fill whole rounds of 4 gates per CU (default dispatch: a remainder of
    #    <= 1 gate per CU on the latency shape, <= 2 / <= 3 per CU on 2 / 3 gates per workgroup; 1,280 and 1,536 gates: one time-sliced launch of
    #    five / six gates per CU, k_bootstrap_pair_rr); two untimed launches, then three timed
    def _batch_sweep():
        sizes = (256, 512, 768, 1280, 1536)
        Gm = max(sizes)
        bb = rng.integers(0, 2, (2, Gm)).astype(np.uint8)
        s0 = torch.from_numpy(R.encrypt_bits(params, key0, bb[0], 61).view(np.int32)).to(gpu)
        s1 = torch.from_numpy(R.encrypt_bits(params, key0, bb[1], 62).view(np.int32)).to(gpu)
        so = torch.empty_like(s0)
        sweep = {}
        for Gs in sizes:
            ms, ks = timed(eng, lambda: eng.gate_batch_dev(R.NAND, s0, s1, so, Gs, stream), 3)
            oks = bool(np.array_equal(R.decrypt_bits(params, key0, so[:Gs].cpu().numpy().view(np.uint32)), 1 - (bb[0][:Gs] & bb[1][:Gs])))
            sweep[str(Gs)] = {"gates_per_s": round(Gs / ms * 1e3, 1), "ms_per_batch": round(ms, 3), "key_switch_ms": round(ks, 3), "ok": oks}
        sec["batch_sweep"] = sweep
        # the same two sizes between rounds on the split-FFT exact backend (k_bootstrap_xpair_rr), words compared with the NTT backend's
        xs = {}
        for Gs in (1280, 1536):
            sn = torch.empty_like(so)
            eng.set_backend(R._ffi.BACKEND_NTT_EXACT)
            eng.gate_batch_dev(R.NAND, s0, s1, sn, Gs, stream); eng.sync(stream)
            eng.set_backend(R._ffi.BACKEND_FFT_SPLIT_EXACT)
            try:
                ms, ks = timed(eng, lambda: eng.gate_batch_dev(R.NAND, s0, s1, so, Gs, stream), 3)
                xs[str(Gs)] = {"gates_per_s": round(Gs / ms * 1e3, 1), "ms_per_batch": round(ms, 3), "equals_ntt_exact_bit_for_bit": bool(torch.equal(so[:Gs], sn[:Gs]))}
            finally:
                eng.set_backend(R._ffi.BACKEND_FFT64_MIRROR)
        sec["batch_sweep_split_fft_exact"] = xs
    guard("batch_sweep", _batch_sweep)
    # -- BASELINE configs[4]: N = 2048, 1024 gates (its own key set and context)
    def _config5_n2048_1024_gates():
        G = 1024
        bb = rng.integers(0, 2, (2, G)).astype(np.uint8)
        p5 = R.Params(N=2048)
        k0, k1, bk5, ksk5 = R.keygen(p5, 20482048)
        e5 = R.Engine(p5, gpu.index)
        try:
            e5.load_bk_torus(bk5)
            e5.load_ksk(ksk5)
            del bk5, ksk5
            x0 = torch.from_numpy(R.encrypt_bits(p5, k0, bb[0], 51).view(np.int32)).to(gpu)
            x1 = torch.from_numpy(R.encrypt_bits(p5, k0, bb[1], 52).view(np.int32)).to(gpu)
            xo = torch.empty_like(x0)
            ms, _ = timed(e5, lambda: e5.gate_batch_dev(R.NAND, x0, x1, xo, G, stream), 5)
            ok5 = bool(np.array_equal(R.decrypt_bits(p5, k0, xo.cpu().numpy().view(np.uint32)), 1 - (bb[0] & bb[1])))
            ops5 = dp_wave_instr_per_cmux(2048, p5.l)["total"]
            sec["config5_n2048_1024_gates"] = {"gates_per_s": round(G / ms * 1e3, 1), "ms_per_launch": round(ms, 3),
                                               "kernel": "k_bootstrap_eo",
                                               "dp_wave_instr_per_cmux": ops5, "roofline_frac_fp64": fp64_frac(ops5, p5.n, G, ms * 1e-3), "ok": ok5}
            # below a full round (the library's default: four waves per gate, k_bootstrap_eo4, at up to two gates per CU): 3 launches each
            small = {}
            for gs in (1, 256, 512):
                ms_s, _ = timed(e5, lambda: e5.gate_batch_dev(R.NAND, x0, x1, xo, gs, stream), 3)
                ok_s = bool(np.array_equal(R.decrypt_bits(p5, k0, xo.cpu().numpy().view(np.uint32)[:gs]), (1 - (bb[0] & bb[1]))[:gs]))
                small[str(gs)] = {"ms_per_launch": round(ms_s, 3), "gates_per_s": round(gs / ms_s * 1e3, 1), "ok": ok_s}
            sec["config5_n2048_small_batches"] = small
            # the two exact backends at N = 2048: the NTT (two waves per transform) and the split FFT (k_bootstrap_xquad, four waves per gate),
            # the latter compared word for word with the former
            xn = torch.empty_like(x0)
            e5.set_backend(R._ffi.BACKEND_NTT_EXACT)
            ms_n, _ = timed(e5, lambda: e5.gate_batch_dev(R.NAND, x0, x1, xn, G, stream), 3)
            e5.set_backend(R._ffi.BACKEND_FFT_SPLIT_EXACT)
            ms_x, _ = timed(e5, lambda: e5.gate_batch_dev(R.NAND, x0, x1, xo, G, stream), 5)
            ok_x = bool(np.array_equal(R.decrypt_bits(p5, k0, xo.cpu().numpy().view(np.uint32)), 1 - (bb[0] & bb[1])))
            xops5 = xfft_dp_wave_instr_per_cmux(2048, p5.l)["total"]
            sec["config5_n2048_exact_backends_1024_gates"] = {
                "ntt_exact": {"gates_per_s": round(G / ms_n * 1e3, 1), "ms_per_launch": round(ms_n, 3), "kernel": "k_bootstrap_ntt_halves"},
                "split_fft_exact": {"gates_per_s": round(G / ms_x * 1e3, 1), "ms_per_launch": round(ms_x, 3), "kernel": "k_bootstrap_xquad",
                                    "dp_wave_instr_per_cmux": xops5, "roofline_frac_fp64_fma": fp64_frac(xops5, p5.n, G, ms_x * 1e-3),
                                    "equals_ntt_exact_bit_for_bit": bool(torch.equal(xo, xn)), "ok": ok_x}}
            e5.set_backend(R._ffi.BACKEND_FFT64_MIRROR)
        finally:
            e5.close()
    guard("config5_n2048_1024_gates", _config5_n2048_1024_gates)
    return sec


def cargo_probe():
    """"if cargo exists on the GPU box, additionally run the reference's homnand-bench".  The reference checkout never travels
    to the GPU box, so what a toolchain there can do is build the shipped shim crates (bindings/rust) against this library; absent
    toolchain = "absent" in the line.  Never fatal."""
    import shutil
    cargo = shutil.which("cargo")
    if not cargo:
        return "absent"
    res = {"cargo": cargo}
    if under_profiler():
        return res
    try:
        res["version"] = subprocess.run([cargo, "--version"], capture_output=True, text=True, timeout=20).stdout.strip()
        crate = os.path.join(ROOT, "bindings", "rust", "rtfhe-sys")
        r = subprocess.run([cargo, "build", "--release", "--offline"], cwd=crate, capture_output=True, text=True, timeout=180,
                           env=dict(os.environ, RTFHE_LIB_DIR=os.path.join(ROOT, "rustfhe_amd")))
        res["rtfhe_sys_build_rc"] = r.returncode
        if r.returncode != 0:
            res["rtfhe_sys_build_stderr_tail"] = r.stderr[-400:]
    except Exception as e:          # noqa: BLE001
        res["error"] = "%s: %s" % (type(e).__name__, e)
    return res


# ------------------------------------------------------------------------------------------------------------------
# one rank
# ------------------------------------------------------------------------------------------------------------------
def run_rank(args):
    import numpy as np
    import torch
    import rustfhe_amd as R
    from rustfhe_amd.shard import ShardedGates, engine_compute

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    inner = bool(os.environ.get("RTFHE_BENCH_INNER"))       # our stdout is read by supervise(): intermediate lines are welcome
    # stdout carries ONE JSON line and nothing else: RCCL prints a version banner to stdout when its first communicator comes up
    # (seen on ROCm 7: "RCCL version : ...", 5 lines).  Until the line is printed, file descriptor 1 points at stderr.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(obj), flush=True)
        os.dup2(2, 1)

    config3 = args.workload == "config3"
    mirror = args.backend == "fft64-mirror"
    G = args.gates if args.gates else (8192 if config3 else 1024)      # gates per GPU per step
    params = R.Params()
    # ---- everything that needs no GPU comes first: keys (rank 0), this rank's inputs, and -- N = 1 only -- the CPU baseline's own
    # process, started BEFORE this process makes its first GPU call and left waiting on its stdin until the timed region is over
    t_key = time.perf_counter()
    if rank == 0:
        key0, key1, bk, ksk = R.keygen(params, 20211003)
    else:
        key0, bk, ksk = np.empty(params.n, np.int32), np.empty(params.bk_words, np.uint32), np.empty(params.ksk_words, np.uint32)
    cargo_info = cargo_probe() if rank == 0 else None      # (may start `cargo`: before this process touches the GPU, like the child below)
    cpu_child = None
    want_cpu = world == 1 and not args.no_cpu_baseline and mirror and not config3 and not under_profiler()
    if want_cpu:
        rng = np.random.default_rng(1000 + rank)
        b0, b1 = rng.integers(0, 2, G).astype(np.uint8), rng.integers(0, 2, G).astype(np.uint8)
        in0 = R.encrypt_bits(params, key0, b0, 5000 + 2 * rank)
        in1 = R.encrypt_bits(params, key0, b1, 5001 + 2 * rank)
        try:
            cpu_child = CpuBaselineChild(np, params, bk, ksk, in0, in1, args.cpu_gates_per_thread)
        except Exception as e:          # noqa: BLE001
            cpu_child = {"error": "could not start the CPU baseline child: %s: %s" % (type(e).__name__, e)}

    if not torch.cuda.is_available():
        if isinstance(cpu_child, CpuBaselineChild):
            cpu_child.close()
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # RTFHE_BENCH_BACKEND=gloo lets several ranks share one GPU (rehearsal of the N > 1 path on a 1-GPU box)
    backend = os.environ.get("RTFHE_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and world > ndev:
        raise SystemExit("bench.py: %d ranks but %d GPUs (RCCL needs one GPU per rank)" % (world, ndev))
    dev = local if backend == "nccl" else local % ndev
    torch.cuda.set_device(dev)
    gpu = torch.device("cuda", dev)
    comm = gpu if backend == "nccl" else torch.device("cpu")      # where collectives run
    dist = None
    # RTFHE_BENCH_FORCE_PG=1: build the process group even for one rank (a 1-GPU box then still runs RCCL's communicator set-up,
    # the key broadcast, the barriers and the max-reduction exactly as the N > 1 run does)
    if world > 1 or os.environ.get("RTFHE_BENCH_FORCE_PG"):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=gpu)
        else:
            dist.init_process_group(backend)
        assert dist.get_world_size() == world
    n_gpus = dist.get_world_size() if dist is not None else 1     # the ranks the communicator really has
    if world > 1 or os.environ.get("RTFHE_BENCH_FORCE_PG"):
        # one line per rank on stderr: enough to read a first multi-GPU run from its log (which card, which peers it can reach over
        # xGMI / P2P, what the communicator looks like)
        peers = [d for d in range(ndev) if d != dev and torch.cuda.can_device_access_peer(dev, d)]
        props = torch.cuda.get_device_properties(dev)
        ident = "%s/%s" % (getattr(props, "uuid", ""), getattr(props, "pci_bus_id", dev))
        # ... and the link the runtime reports from this rank's card to every other card of the node (queries only)
        links = {}
        for d in range(ndev):
            if d != dev:
                try:
                    lk = R.device_link(dev, d)
                    links[d] = "%s/%d hop%s%s" % (lk["link"], lk["hops"], "" if lk["hops"] == 1 else "s", "" if lk["can_access"] else " (no peer access)")
                except Exception as e:      # noqa: BLE001
                    links[d] = "query failed: %s" % e
        sys.stderr.write("bench.py rank %d/%d: device %d of %d (%s, %s), backend %s, communicator world size %d, P2P-reachable peers %s, links %s, "
                         "HSA_ENABLE_IPC_MODE_LEGACY=%s\n" % (rank, world, dev, ndev, torch.cuda.get_device_name(dev), ident, backend, n_gpus, peers, links,
                                                              os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")))
        # a whole-node number is only a whole-node number if every rank drives a card of its own: under RCCL the ranks compare device
        # identities and the job refuses to print a line when two of them share one
        idents = [None] * n_gpus
        dist.all_gather_object(idents, ident)
        if backend == "nccl" and len(set(idents)) != n_gpus:
            raise SystemExit("bench.py: %d ranks but only %d distinct GPUs (%s): no line is printed for such a run" % (n_gpus, len(set(idents)), sorted(set(idents))))

    # ---- keys: generated ONCE (rank 0, above) and broadcast to the other ranks (RCCL), then loaded into each rank's engine ----
    if dist is not None:
        for arr in (key0, bk, ksk):
            t = torch.from_numpy(arr.view(np.int32)).to(comm)
            dist.broadcast(t, src=0)
            if rank != 0:
                arr.view(np.int32)[:] = t.cpu().numpy()
            del t
    eng = R.Engine(params, dev)
    eng.load_bk_torus(bk)
    eng.load_ksk(ksk)
    key_s = time.perf_counter() - t_key
    if args.backend == "ntt-exact":
        eng.set_backend(R._ffi.BACKEND_NTT_EXACT)
    if args.backend == "split-fft-exact":
        eng.set_backend(R._ffi.BACKEND_FFT_SPLIT_EXACT)

    stream = torch.cuda.current_stream().cuda_stream

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    phase = None
    settle = 0
    if config3:
        # the WHOLE batch (G gates per GPU) starts as ciphertexts in rank 0's HBM and the results end there
        total = G * world
        sg = ShardedGates(engine_compute(eng, gpu), params.n + 1, comm)
        full0 = full1 = None
        if rank == 0:
            rng = np.random.default_rng(1000)
            b0, b1 = rng.integers(0, 2, total).astype(np.uint8), rng.integers(0, 2, total).astype(np.uint8)
            in0 = R.encrypt_bits(params, key0, b0, 5000)
            in1 = R.encrypt_bits(params, key0, b1, 5001)
            full0 = torch.from_numpy(in0.view(np.int32)).to(comm)
            full1 = torch.from_numpy(in1.view(np.int32)).to(comm)
        # self-check before anything is timed: the batch as scattered, bootstrapped on every rank and gathered must be, word
        # for word, what rank 0's card computes alone on the first 2 x 8192 gates -- on distinct devices this is the first time the P2P
        # sends / receives carry real data, and a job that fails here prints no line
        check = sg.run(R.NAND, full0, full1, total)
        if world > 1:
            verdict = torch.zeros(1, dtype=torch.int32, device=comm)
            if rank == 0:
                Gc = min(total, 2 * 8192)
                ref = torch.empty((Gc, params.n + 1), dtype=torch.int32, device=gpu)
                eng.gate_batch_dev(R.NAND, full0.to(gpu), full1.to(gpu), ref, Gc, stream); eng.sync(stream)
                differing = int((check[:Gc].to(gpu) != ref).any(dim=1).sum().item())
                verdict[0] = differing
                del ref
            dist.broadcast(verdict, src=0)
            if int(verdict.item()):
                raise SystemExit("bench.py: SELF-CHECK FAILED before timing: %d gates of the sharded batch differ from rank 0's single-GPU batch; no line is printed"
                                 % int(verdict.item()))
        del check
        for _ in range(args.warmup):
            sg.run(R.NAND, full0, full1, total)
        barrier()
        t0 = time.perf_counter()
        eng.timer_begin(stream)
        acc = {"scatter_s": 0.0, "compute_s": 0.0, "gather_s": 0.0}
        for _ in range(args.steps):
            res = sg.run(R.NAND, full0, full1, total, sync=torch.cuda.synchronize)
            for k in acc:
                acc[k] += sg.last_timing[k]
        kern_ms, ks_ms, launches = eng.timer_end_detail(stream)
        barrier()
        elapsed = time.perf_counter() - t0
        phase = {k.replace("_s", "_ms_per_step"): round(1e3 * v / args.steps, 3) for k, v in acc.items()}
        ok = True
        if rank == 0:
            out = res.cpu().numpy().view(np.uint32)
            ok = bool(np.array_equal(R.decrypt_bits(params, key0, out), 1 - (b0 & b1)))
    else:
        if not want_cpu:
            rng = np.random.default_rng(1000 + rank)
            b0, b1 = rng.integers(0, 2, G).astype(np.uint8), rng.integers(0, 2, G).astype(np.uint8)
            in0 = R.encrypt_bits(params, key0, b0, 5000 + 2 * rank)
            in1 = R.encrypt_bits(params, key0, b1, 5001 + 2 * rank)
        d_in0 = torch.from_numpy(in0.view(np.int32)).to(gpu)
        d_in1 = torch.from_numpy(in1.view(np.int32)).to(gpu)
        d_out = torch.empty_like(d_in0)
        # Optional clock settle (--settle-max N > 0; default OFF: the driver's contract is exactly W untimed warm-up steps): the first launches
        # after idle run at a lower clock (rocprofv3 of round 3: 8.0 -> 6.4 ms over five launches).  Untimed launches until three in a row agree
        # within 1 %, at most N; then the W warm-up steps, then the K timed ones.  The count is reported in config.clock_settle_launches.
        settle, recent = 0, []
        while settle < args.settle_max:
            eng.timer_begin(stream)
            eng.gate_batch_dev(R.NAND, d_in0, d_in1, d_out, G, stream)
            ms1, _ = eng.timer_end(stream)
            settle += 1
            recent = (recent + [ms1])[-3:]
            if len(recent) == 3 and max(recent) <= 1.01 * min(recent):
                break
        for _ in range(args.warmup):
            eng.gate_batch_dev(R.NAND, d_in0, d_in1, d_out, G, stream)
        barrier()
        t0 = time.perf_counter()
        eng.timer_begin(stream)
        for _ in range(args.steps):
            eng.gate_batch_dev(R.NAND, d_in0, d_in1, d_out, G, stream)
        kern_ms, ks_ms, launches = eng.timer_end_detail(stream)
        barrier()
        elapsed = time.perf_counter() - t0
        out = d_out.cpu().numpy().view(np.uint32)
        ok = bool(np.array_equal(R.decrypt_bits(params, key0, out), 1 - (b0 & b1)))

    t = torch.tensor([elapsed, 0.0 if ok else 1.0, kern_ms], dtype=torch.float64, device=comm)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, bad, kern_ms_max = float(t[0]), float(t[1]), float(t[2])

    if rank == 0:
        value = n_gpus * G * args.steps / elapsed
        step_s = kern_ms * 1e-3 / args.steps            # device time of one batch on this rank (HIP events on the launch stream)
        ks_s = ks_ms * 1e-3 / args.steps                # of it: the batch key switch of the split path (its own launch; 0 when fused)
        launch_s = step_s - ks_s                        # the dominant kernel: blind rotation (+ fused key switch when not split)
        mirror = args.backend == "fft64-mirror"
        kernel = "k_bootstrap_pair" if mirror else ("k_bootstrap_xpair" if args.backend == "split-fft-exact" else "k_bootstrap_ntt_pair")
        ops = dp_wave_instr_per_cmux(params.N, params.l)
        line = {
            "metric": "HomNAND gates/sec (whole node), N=1024", "value": round(value, 1), "unit": "gates/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("batch of %d HomNAND gates (%d per GPU) held on rank 0's GPU: scatter over xGMI -> bootstrap -> gather, "
                                    "all inside the timed step (BASELINE configs[2])" % (G * n_gpus, G)) if config3 else
                                   ("batch of %d independent HomNAND gates per GPU per step (BASELINE configs[1])" % G),
                       "params": "N=1024, n=635, l=3, Bgbit=6, ks t=8 basebit=2", "gates_per_gpu": G, "backend": args.backend,
                       "sharding": "contiguous gate ranges, replicated keys (generated on rank 0, broadcast); " +
                                   ("P2P scatter/gather of ciphertexts only" if config3 else "no data-path collective"),
                       "comm_backend": (backend if dist is not None else None), "key_setup_s": round(key_s, 2),
                       "clock_settle_launches": settle},
            "outputs_decrypt_correctly": ok and bad == 0.0,
        }
        if phase:
            line["phase_ms_per_step_rank0"] = phase
            line["scatter_ms"], line["gather_ms"] = phase["scatter_ms_per_step"], phase["gather_ms_per_step"]      # per step, on rank 0's clock
        if mirror:
            # The ceiling that binds: FP64 vector issue with no FMA.  achieved = FP64-rate VALU lane-results per
            # second of the dominant kernel = counted wave-instructions per gate (structure of the transform, checked against
            # the ISA of the built kernel) x 64 lanes x gates per launch / measured launch time.
            dp_gate = ops["total"] * params.n
            achieved = dp_gate * 64 * G / launch_s
            line["roofline"] = {
                "bound": "fp64-valu-nofma", "achieved": round(achieved / 1e12, 3), "peak": round(FP64_VALU_PEAK / 1e12, 3), "unit": "Tops/s",
                "frac": round(achieved / FP64_VALU_PEAK, 4), "traffic": None, "kernel": kernel, "avg_launch_ms": round(1e3 * launch_s, 3),
                "peak_is": "%d CU x %d SIMD x %d DP lanes/clk x %.1f GHz (MI355X_MICROARCH.md max clock)" % (CUS, SIMDS_PER_CU, DP_LANES_PER_CLK, CLOCK_HZ / 1e9),
                "dp_wave_instr_per_gate": dp_gate, "dp_wave_instr_per_cmux": ops, "gates_per_launch": G,
                "launches_per_batch": round(launches / args.steps, 2), "device_ms_per_batch": round(1e3 * step_s, 3),
                "timing": "HIP events on the launch stream over the timed steps; avg_launch_ms = (whole batch - the key-switch launches, "
                          "which are bracketed by events of their own) / steps",
                # SURVEY 8(d)'s streaming model, kept for the record: it is NOT a bound for a batch that shares bk_i through L2
                "hbm_algorithmic": {"bytes_per_gate": ALG_BYTES_PER_GATE, "GBps": round(ALG_BYTES_PER_GATE * G / launch_s / 1e9, 2),
                                    "vs_hbm_peak": round(ALG_BYTES_PER_GATE * G / launch_s / HBM_PEAK, 4), "hbm_peak_GBps": HBM_PEAK / 1e9},
            }
            if ks_s > 0:
                # the second launch of a batch: the key switch of all G gates as one exact i8 contraction on the matrix pipe
                # (one-hot digits x signed byte limbs; rtfhe_kernels_ksmm.hpp).  M = gates, N = 16 ceil((n+1)/16) x 4 limbs, K = N_ring 8 x 4.
                mm_ops = 2.0 * G * (16 * ((params.n + 1 + 15) // 16) * 4) * (params.N * params.ks_t * 4)
                line["roofline"]["key_switch_kernel"] = {
                    "kernel": "k_key_switch_mm", "avg_launch_ms": round(1e3 * ks_s, 4), "bound": "mfma", "unit": "Pops/s (i8)",
                    "achieved": round(mm_ops / ks_s / 1e15, 4), "peak": 5.0, "frac": round(mm_ops / ks_s / 5.0e15, 4),
                    "peak_is": "dense i8 MFMA = 2 x the ~2.5 PF bf16 peak (MI355X_MICROARCH.md, Matrix cores)",
                    "share_of_batch": round(ks_s / step_s, 4),
                    "note": "includes the memset of the output; 25 % of the contraction's K are the zero rows of digit 0"}
        else:
            # same ceiling (FP64-rate vector issue; the NTT's v_fma_f64 count as one instruction each)
            nops = xfft_dp_wave_instr_per_cmux(params.N, params.l) if args.backend == "split-fft-exact" else ntt_dp_wave_instr_per_cmux(params.N, params.l)
            dp_gate = nops["total"] * params.n
            achieved = dp_gate * 64 * G / launch_s
            line["roofline"] = {
                "bound": "fp64-valu", "achieved": round(achieved / 1e12, 3), "peak": round(FP64_VALU_PEAK / 1e12, 3), "unit": "Tops/s",
                "frac": round(achieved / FP64_VALU_PEAK, 4), "traffic": None, "kernel": kernel, "avg_launch_ms": round(1e3 * launch_s, 3),
                "peak_is": "%d CU x %d SIMD x %d DP lanes/clk x %.1f GHz (MI355X_MICROARCH.md max clock)" % (CUS, SIMDS_PER_CU, DP_LANES_PER_CLK, CLOCK_HZ / 1e9),
                "dp_wave_instr_per_gate": dp_gate, "dp_wave_instr_per_cmux": nops, "gates_per_launch": G,
                "launches_per_batch": round(launches / args.steps, 2),
                "hbm_algorithmic": {"bytes_per_gate": ALG_BYTES_PER_GATE, "GBps": round(ALG_BYTES_PER_GATE * G / launch_s / 1e9, 2),
                                    "vs_hbm_peak": round(ALG_BYTES_PER_GATE * G / launch_s / HBM_PEAK, 4), "hbm_peak_GBps": HBM_PEAK / 1e9},
            }
        # measured HBM-side bytes per launch (rocprofv3 --pmc passes): only from a profile of THIS device code and launch shape
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            with open(pmc) as f:
                j = json.load(f)
            if j.get("gates_per_launch") == G and j.get("kernel") == kernel and j.get("src_hash") == kernel_src_hash():
                line["roofline"]["traffic"] = j.get("hbm_bytes_per_launch")
                line["roofline"]["traffic_source"] = j.get("source")
                line["roofline"]["traffic_measured_in_this_run"] = False     # a builder-side rocprofv3 --pmc profile of the same device code
            else:
                line["roofline"]["traffic_note"] = "profiles/pmc_traffic.json is from other device code or another launch shape: not quoted"
        # The headline above is complete at this point, and nothing below may cost it.  Under supervise() (plain `python bench.py`, N = 1) the
        # line goes out NOW, marked provisional, and again after every side leg: if this process then dies of a native crash in a side
        # kernel or library, the supervisor prints the last line it saw and exits 0.  The CPU baseline runs in a process of its own.
        line["cargo"] = cargo_info
        pending = []
        if isinstance(cpu_child, (CpuBaselineChild, dict)):
            pending.append("cpu_baseline")
        want_secondary = n_gpus == 1 and not args.no_secondary and mirror and not config3 and not args.gates
        if want_secondary:
            pending.append("secondary")

        def progress():
            if inner and pending:
                emit(dict(line, provisional=list(pending)))
        progress()
        if isinstance(cpu_child, dict):
            line["cpu_baseline"] = cpu_child
            pending.remove("cpu_baseline")
        elif cpu_child is not None:
            line["cpu_baseline"] = cpu_child.collect(out)
            pending.remove("cpu_baseline")
            progress()
        if want_secondary:
            # the other BASELINE configs + the NTT backend + the batch sweep on the same clock (headline fields above are not touched by this)
            if os.environ.get("RTFHE_BENCH_TEST_ABORT") == "secondary":       # tests: the headline must survive this
                os.abort()
            try:
                line["secondary"] = secondary_measurements(R, params, eng, key0, bk, ksk, gpu, stream, torch, np)
            except Exception as e:          # noqa: BLE001
                line["secondary"] = {"error": "%s: %s" % (type(e).__name__, e)}
            pending.remove("secondary")
        emit(line)
    elif isinstance(cpu_child, CpuBaselineChild):
        cpu_child.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


def run_config3_c_abi(args):
    """BASELINE configs[2] in ONE process behind the C ABI (builder-side tool; the driver's multi-GPU runs are one process per GPU): a
    multi-device context over --devices, 8192 gates per device resident on devices[0], every step = rtfhe_gate_batch_dev on that context
    (the library scatters over xGMI, bootstraps on every device, gathers; rustfhe_amd/csrc/rtfhe_multi.hip).  Prints one JSON line."""
    import numpy as np
    import torch
    import rustfhe_amd as R
    devs = [int(d) for d in args.devices.split(",")]
    params = R.Params()
    key0, key1, bk, ksk = R.keygen(params, 20211003)
    G = (args.gates if args.gates else 8192) * len(devs)
    gpu = torch.device("cuda", devs[0])
    torch.cuda.set_device(gpu)
    eng = R.Engine(params, devices=devs)
    try:
        eng.load_bk_torus(bk)
        eng.load_ksk(ksk)
        if args.backend == "ntt-exact":
            eng.set_backend(R._ffi.BACKEND_NTT_EXACT)
        if args.backend == "split-fft-exact":
            eng.set_backend(R._ffi.BACKEND_FFT_SPLIT_EXACT)
        # first contact between the devices happened in rtfhe_ctx_create_multi: say what the runtime reported, per entry
        peers = [eng.peer_info(d) for d in range(1, len(devs))]
        for d, i in enumerate(peers, 1):
            sys.stderr.write("bench.py config3-c-abi: entry %d = device %d: %s; peer access primary->entry can %d enabled %d, entry->primary can %d enabled %d; "
                             "link %s, %d hop(s)\n" % (d, i["device"], "SAME device as the primary (rehearsal on one card)" if i["same_device"] else "distinct device",
                                                       i["can_access_from_primary"], i["enabled_from_primary"], i["can_access_to_primary"], i["enabled_to_primary"],
                                                       i["link"], i["hops"]))
        rng = np.random.default_rng(3)
        b0, b1 = rng.integers(0, 2, G).astype(np.uint8), rng.integers(0, 2, G).astype(np.uint8)
        d0 = torch.from_numpy(R.encrypt_bits(params, key0, b0, 1).view(np.int32)).to(gpu)
        d1 = torch.from_numpy(R.encrypt_bits(params, key0, b1, 2).view(np.int32)).to(gpu)
        do = torch.empty_like(d0)
        st = torch.cuda.current_stream().cuda_stream
        # self-check before anything is timed: the sharded batch must be the single-device batch, word for word, on 2 x 8192 gates (the context's
        # own primary as a one-device engine computes the reference; on distinct devices this is the first time peer copies and cross-device
        # events carry real data)
        selfcheck = None
        if len(devs) > 1:
            Gc = min(G, 2 * 8192)
            one = R.Engine(params, devs[0])
            try:
                one.load_bk_torus(bk)
                one.load_ksk(ksk)
                one.set_backend(eng.backend())
                ref = torch.empty_like(d0[:Gc])
                one.gate_batch_dev(R.NAND, d0, d1, ref, Gc, st); one.sync(st)
            finally:
                one.close()
            eng.gate_batch_dev(R.NAND, d0, d1, do, Gc, st); eng.sync(st)
            differing = int((do[:Gc] != ref).any(dim=1).sum().item())
            selfcheck = {"gates": Gc, "differing_gates": differing}
            if differing:
                first = int((do[:Gc] != ref).any(dim=1).nonzero()[0].item())
                lo_hi = [R.shard_range(Gc, d, len(devs)) for d in range(len(devs))]
                sys.stderr.write("bench.py config3-c-abi: SELF-CHECK FAILED before timing: %d of %d gates of the sharded batch differ from the single-device batch "
                                 "(first at gate %d; shard ranges %s; peers %s) -- nothing was timed\n" % (differing, Gc, first, lo_hi, peers))
                print(json.dumps({"metric": "HomNAND gates/sec (whole node), N=1024", "value": None, "ok": False, "selfcheck": selfcheck, "peers": peers}), flush=True)
                return 3
            del ref
        for _ in range(args.warmup):
            eng.gate_batch_dev(R.NAND, d0, d1, do, G, st)
        eng.sync(st)
        phases = [[] for _ in peers]
        t0 = time.perf_counter()
        for _ in range(args.steps):
            eng.gate_batch_dev(R.NAND, d0, d1, do, G, st)
        eng.sync(st)
        elapsed = time.perf_counter() - t0
        # the entries' phases of one more (untimed) step: pull from the primary, bootstrap, push back -- events on each entry's own stream
        eng.gate_batch_dev(R.NAND, d0, d1, do, G, st); eng.sync(st)
        for d in range(1, len(devs)):
            i = eng.peer_info(d)
            phases[d - 1] = {"entry": d, "device": i["device"], "scatter_ms": i["scatter_ms"], "compute_ms": i["compute_ms"], "gather_ms": i["gather_ms"]}
        ok = bool(np.array_equal(R.decrypt_bits(params, key0, do.cpu().numpy().view(np.uint32)), 1 - (b0 & b1)))
        for ph in phases:
            sys.stderr.write("bench.py config3-c-abi: entry %(entry)d (device %(device)d) per step: scatter %(scatter_ms)s ms, bootstrap %(compute_ms)s ms, gather %(gather_ms)s ms\n" % ph)
        scat = [ph["scatter_ms"] for ph in phases if ph["scatter_ms"] is not None]
        gath = [ph["gather_ms"] for ph in phases if ph["gather_ms"] is not None]
        print(json.dumps({"metric": "HomNAND gates/sec (whole node), N=1024", "value": round(G * args.steps / elapsed, 1), "unit": "gates/s",
                          "n_gpus": len(set(devs)), "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                          "config": {"workload": "batch of %d HomNAND gates resident on device %d, sharded over a multi-device CONTEXT of %d entries by "
                                                 "rtfhe_gate_batch_dev (C ABI; BASELINE configs[2])" % (G, devs[0], len(devs)),
                                     "devices": devs, "backend": args.backend, "device_bytes_per_entry": [eng.memory_bytes(d) for d in range(len(devs))]},
                          "scatter_ms": round(max(scat), 4) if scat else None, "gather_ms": round(max(gath), 4) if gath else None,
                          "scatter_gather_is": "the slowest entry's pull of its inputs from / push of its outputs to the primary in one step (HIP events on the entry's own stream)",
                          "entries": phases, "peers": peers, "selfcheck_before_timing": selfcheck,
                          "outputs_decrypt_correctly": ok, "ok": bool(ok)}), flush=True)
        return 0 if ok else 3
    finally:
        eng.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=["config2", "config3", "config3-c-abi"], default="config2",
                    help="config2 (default, BASELINE configs[1]): independent 1024-gate batches per GPU; config3 (BASELINE configs[2]): "
                         "8192 gates per GPU scattered from / gathered to rank 0 inside the timed step; config3-c-abi: the same batch sharded "
                         "inside the library by a multi-device context over --devices, one process")
    ap.add_argument("--devices", default="0", help="config3-c-abi: comma-separated device ids of the multi-device context (an id may repeat)")
    ap.add_argument("--gates", type=int, default=0, help="gates per GPU per step (default 1024, config3: 8192)")
    ap.add_argument("--backend", choices=["fft64-mirror", "ntt-exact", "split-fft-exact"], default="fft64-mirror",
                    help="fft64-mirror (default): bit-identical to the reference CPU path; ntt-exact: exact-integer NTT; split-fft-exact: the same "
                         "exact products through an FMA FP64 FFT with the key in 16-bit halves")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the other BASELINE configs / NTT backend measured after the headline")
    ap.add_argument("--cpu-gates-per-thread", type=int, default=24, help="least number of gates every thread of the all-core CPU baseline runs")
    ap.add_argument("--settle-max", type=int, default=0, help="untimed clock-settle launches BEFORE the caller's warm-up steps, at most this many "
                    "(default 0: exactly --warmup untimed steps, the driver's contract; > 0: launches until three in a row agree within 1 %%)")
    ap.add_argument("--cpu-baseline-child", metavar="DIR", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_child:
        sys.exit(cpu_baseline_child(args.cpu_baseline_child, args.cpu_gates_per_thread))
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.workload == "config3-c-abi":
        sys.exit(run_config3_c_abi(args))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch(args))                      # parent: spawn the ranks, never touch the GPU
    if "WORLD_SIZE" not in os.environ and not os.environ.get("RTFHE_BENCH_INNER") and not under_profiler():
        # N = 1, started by hand or by the driver: this process only supervises (never imports torch, never touches the GPU)
        sys.exit(supervise([sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%s; the launcher's world size is used\n" % (args.gpus, os.environ["WORLD_SIZE"]))
    run_rank(args)


if __name__ == "__main__":
    main()
