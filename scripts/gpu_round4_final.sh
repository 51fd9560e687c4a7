#!/bin/bash
# Round-4 evidence pass: GPU tests, bench lines (default incl. CPU child + secondary / driver's flags / config 3 / 2-rank rehearsal / NTT backend), sweeps,
# circuits, rocprofv3 kernel trace + PMC passes of the headline, N = 2048 counters.  Outputs under gpurun_out/final4/ (copy what is to be judged to profiles/r04/).
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/final4; mkdir -p $O
python scripts/host_probe.py > $O/host_probe.json 2>&1
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-300 $O/bench.json
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench_steps20_warmup5.json 2> $O/bench20.err; echo "bench20 rc=$?"; cut -c1-300 $O/bench_steps20_warmup5.json
timeout -k 10 300 python bench.py --workload config3 --steps 3 --warmup 1 > $O/bench_config3_1gpu.json 2> $O/bench_config3.err; echo "config3 rc=$?"
RTFHE_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 4 --warmup 1 > $O/bench_gpus2_gloo_rehearsal.json 2> $O/bench_gpus2.err; echo "gpus2 rc=$?"; grep "bench.py rank" $O/bench_gpus2.err
timeout -k 10 300 python bench.py --backend ntt-exact --no-cpu-baseline > $O/bench_ntt_exact.json 2> $O/bench_ntt.err; echo "ntt rc=$?"; cut -c1-200 $O/bench_ntt_exact.json
RTFHE_SKIP_STAGES=1 timeout -k 10 300 python scripts/sweep.py 1,256,512,768,1024,1280,1536,2048,4096,8192 > $O/sweep.log 2>&1; echo "sweep rc=$?"; grep -v amdgpu.ids $O/sweep.log
RTFHE_N=2048 RTFHE_SKIP_STAGES=1 timeout -k 10 300 python scripts/sweep.py 1,256,512,768,1024,2048 > $O/sweep_n2048.log 2>&1; echo "sweep2048 rc=$?"; grep -v amdgpu.ids $O/sweep_n2048.log
RTFHE_BACKEND=ntt RTFHE_SKIP_STAGES=1 timeout -k 10 300 python scripts/sweep.py 1,512,1024 > $O/sweep_ntt.log 2>&1; echo "sweep ntt rc=$?"; grep -v amdgpu.ids $O/sweep_ntt.log
timeout -k 10 300 python scripts/bench_circuit.py > $O/bench_circuit.log 2>&1; echo "circuit rc=$?"; grep -v amdgpu.ids $O/bench_circuit.log | tail -12
bash scripts/profile_gpu.sh r04 > $O/profile.log 2>&1; echo "profile rc=$?"; tail -3 $O/profile.log
bash scripts/profile_n2048.sh halves > $O/profile_n2048.log 2>&1; echo "profile n2048 rc=$?"
cp -r gpurun_out/profiles_r04 $O/ 2>/dev/null; cp gpurun_out/profiles_n2048/pmc_n2048_halves.json $O/ 2>/dev/null
