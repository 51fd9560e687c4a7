#!/usr/bin/env python3
"""bench.py -- HomNAND gates/sec on N MI355X (BASELINE.json metric).

A "step" is one pass of the hot path (pre-step, blind rotate, sample extract, identity key switch)
over one batch of `--gates` independent NAND gates per GPU, ONE kernel launch, inputs and keys already
resident in HBM.  Weak scaling: every rank owns its own batch, no data-path collective (gates are
independent).  Rank 0 prints ONE JSON line.

    python bench.py                                   # 1 GPU, defaults finish in well under a minute
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# SURVEY 8(d) / BASELINE.md 4: algorithmic bytes one gate must consume at N=1024, n=635, l=3
BK_BYTES_PER_GATE = 635 * 2 * 6 * 1024 * 8            # 62,423,040  whole bootstrapping key once
KSK_BYTES_PER_GATE = 1024 * 8 * 3 // 4 * 2544          # 15,630,336  expected touched key-switch rows
IO_BYTES_PER_GATE = 3 * 2544                           #      7,632  two inputs + one output
ALG_BYTES_PER_GATE = BK_BYTES_PER_GATE + KSK_BYTES_PER_GATE + IO_BYTES_PER_GATE   # 78,061,008
HBM_PEAK = 8.0e12                                      # MI355X_MICROARCH.md: 8 TB/s spec


def cpu_baseline(R, params, key_bk_t, ksk, in0, in1, gpu_out, per_thread):
    """The CPU restatement (oracle/, FP64 mirror of the reference's spqlios) timed on this host's cores.
    Checker/baseline only -- never on the product path."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C
    import orc
    threads = max(1, len(os.sched_getaffinity(0)))
    sample = min(len(in0), threads * per_thread)
    p = orc.Params(n=params.n, N=params.N, l=params.l, bgbit=params.bgbit, ks_t=params.ks_t, ks_basebit=params.ks_basebit)
    pl = orc.Plan(p.N)
    bk_f = np.empty(key_bk_t.size, np.float64)
    orc.lib().orc_trgsw_to_fft(pl.h, key_bk_t.ctypes.data_as(C.POINTER(C.c_uint32)),
                               bk_f.ctypes.data_as(C.POINTER(C.c_double)), key_bk_t.size // p.N)
    out, secs = orc.gate_batch_mt(p, orc.NAND, bk_f, None, ksk, in0[:sample], in1[:sample], threads)
    one, secs1 = orc.gate_batch_mt(p, orc.NAND, bk_f, None, ksk, in0[:4], in1[:4], 1)
    port = {
        "value": round(sample / secs, 2), "unit": "gates/s", "cores": threads, "kind": "port",
        "sample": "%d NAND gates (%d per thread, one independent gate stream per thread), oracle/tfhe_oracle.c "
                  "FP64 mirror of the reference spqlios FFT, gcc -O3 -march=native -ffp-contract=off" % (sample, per_thread),
        "single_thread_ms_per_gate": round(1e3 * secs1 / 4, 2),
        "matches_gpu_bit_exact": bool(np.array_equal(out, gpu_out[:sample])),
    }
    if not orc.have_ref():
        return port
    # the reference's OWN compiled native FFT (oracle/_ref: utils/src/spqlios/*.cpp + AVX .s built with its build.rs flags)
    # under the restated Rust glue (the Rust half cannot be built here: no toolchain); one handle per thread
    orc.use_reference_fft_in_mt()
    out_r, secs_r = orc.gate_batch_mt(p, orc.NAND, bk_f, None, ksk, in0[:sample], in1[:sample], threads, backend=orc.BACKEND_HOOK)
    one_r, secs_r1 = orc.gate_batch_mt(p, orc.NAND, bk_f, None, ksk, in0[:4], in1[:4], 1, backend=orc.BACKEND_HOOK)
    return {
        "value": round(sample / secs_r, 2), "unit": "gates/s", "cores": threads, "kind": "reference",
        "sample": "%d NAND gates (%d per thread, one gate stream + one Spqlios handle per thread): the reference's own compiled "
                  "spqlios AVX FFT (oracle/_ref, flags of utils/build.rs) under the C restatement of its Rust glue" % (sample, per_thread),
        "single_thread_ms_per_gate": round(1e3 * secs_r1 / 4, 2),
        "matches_gpu_bit_exact": bool(np.array_equal(out_r, gpu_out[:sample])),
        "port": port,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--gates", type=int, default=1024, help="gates per GPU per step (BASELINE configs[1]: 1024)")
    ap.add_argument("--backend", choices=["fft64-mirror", "ntt-exact"], default="fft64-mirror",
                    help="fft64-mirror (default): bit-identical to the reference CPU path; ntt-exact: exact-integer NTT")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-gates-per-thread", type=int, default=8)
    args = ap.parse_args()

    import torch
    import rustfhe_amd as R

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # RTFHE_BENCH_BACKEND=gloo lets several ranks share one GPU (rehearsal of the N > 1 path on a 1-GPU box)
    backend = os.environ.get("RTFHE_BENCH_BACKEND", "nccl")
    dev = local if backend == "nccl" else local % torch.cuda.device_count()
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend)

    params = R.Params()
    key0, key1, bk, ksk = R.keygen(params, 20211003)           # same key set on every rank (replicated keys)
    eng = R.Engine(params, dev)
    eng.load_bk_torus(bk)
    eng.load_ksk(ksk)
    if args.backend == "ntt-exact":
        eng.set_backend(R._ffi.BACKEND_NTT_EXACT)

    G = args.gates
    rng = np.random.default_rng(1000 + rank)
    b0, b1 = rng.integers(0, 2, G).astype(np.uint8), rng.integers(0, 2, G).astype(np.uint8)
    in0 = R.encrypt_bits(params, key0, b0, 5000 + 2 * rank)
    in1 = R.encrypt_bits(params, key0, b1, 5001 + 2 * rank)
    d_in0 = torch.from_numpy(in0.view(np.int32)).cuda()
    d_in1 = torch.from_numpy(in1.view(np.int32)).cuda()
    d_out = torch.empty_like(d_in0)
    stream = torch.cuda.current_stream().cuda_stream

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        eng.gate_batch_dev(R.NAND, d_in0, d_in1, d_out, G, stream)
    barrier()
    t0 = time.perf_counter()
    eng.timer_begin(stream)
    for _ in range(args.steps):
        eng.gate_batch_dev(R.NAND, d_in0, d_in1, d_out, G, stream)
    kern_ms, launches = eng.timer_end(stream)
    barrier()
    elapsed = time.perf_counter() - t0

    out = d_out.cpu().numpy().view(np.uint32)
    ok = bool(np.array_equal(R.decrypt_bits(params, key0, out), 1 - (b0 & b1)))
    t = torch.tensor([elapsed, 0.0 if ok else 1.0], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, bad = float(t[0]), float(t[1])

    if rank == 0:
        value = world * G * args.steps / elapsed
        launch_s = kern_ms * 1e-3 / args.steps          # one batch; whole rounds of 4 gates per CU are ONE kernel launch
        achieved = ALG_BYTES_PER_GATE * G / launch_s
        line = {
            "metric": "HomNAND gates/sec (whole node), N=1024", "value": round(value, 1), "unit": "gates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "batch of %d independent HomNAND gates per GPU per step, N=1024, n=635, l=3, Bgbit=6, "
                                   "ks t=8 basebit=2 (BASELINE configs[1])" % G,
                       "gates_per_gpu": G, "backend": args.backend, "sharding": "independent gate batches, replicated keys"},
            "outputs_decrypt_correctly": ok and bad == 0.0,
            "roofline": {"bound": "hbm", "achieved": round(achieved / 1e9, 2), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK, 4), "traffic": None,
                         "kernel": "k_bootstrap_pair" if args.backend == "fft64-mirror" else "k_bootstrap_ntt_pair", "avg_launch_ms": round(1e3 * launch_s, 3),
                         "alg_bytes_per_gate": ALG_BYTES_PER_GATE, "gates_per_launch": G, "launches_per_batch": round(launches / args.steps, 2)},
        }
        if args.backend == "fft64-mirror" and G % 1024 == 0:
            # what actually binds the kernel: FP64 issue.  The mirror arithmetic may not fuse multiply-add, a
            # gate costs 3,808 FP64 wave-instructions per CMUX, and the two waves a SIMD holds (LDS and VGPR budget) issue
            # one every 5.4 cycles together (scripts/ubench/fp64_issue.hip, profiles/ubench/fp64_lds_issue_rates.log);
            # clock from GRBM_GUI_ACTIVE (2.36 GHz).  One gate per SIMD per round of 1024 gates.
            dp_ops, cyc, clk = 3808 * params.n, 5.4, 2.36e9
            floor_s = dp_ops * cyc / clk * (G // 1024)
            line["fp64_issue_roofline"] = {"bound": "fp64 issue, two waves per SIMD, no FMA", "dp_wave_instr_per_gate": dp_ops,
                                           "cycles_per_instr": cyc, "clock_hz": clk, "floor_ms_per_launch": round(1e3 * floor_s, 3),
                                           "frac": round(floor_s / launch_s, 4)}
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            with open(pmc) as f:
                j = json.load(f)
            if j.get("gates_per_launch") == G and args.backend == "fft64-mirror":
                line["roofline"]["traffic"] = j.get("hbm_bytes_per_launch")
        if world == 1 and not args.no_cpu_baseline and args.backend == "fft64-mirror":
            line["cpu_baseline"] = cpu_baseline(R, params, bk, ksk, in0, in1, out, args.cpu_gates_per_thread)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
