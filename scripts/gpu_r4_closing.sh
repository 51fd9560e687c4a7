#!/bin/bash
# round 4, closing evidence of the final tree: what scripts/gpu_r4_t.sh (GPU suite, smoke, profile, the two bench lines) does not retake --
# config 3 on one GPU, the NTT backend's line, the three sweeps, the adder netlists, the counters of the N = 2048 kernel, the host probe
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/closing4; mkdir -p $O
python scripts/host_probe.py > $O/host_probe.json 2>&1
timeout -k 10 300 python bench.py --workload config3 --steps 3 --warmup 1 > $O/bench_config3_1gpu.json 2> $O/bench_config3.err; echo "config3 rc=$?"; cut -c1-160 $O/bench_config3_1gpu.json
timeout -k 10 300 python bench.py --backend ntt-exact --no-cpu-baseline > $O/bench_ntt_exact.json 2> $O/bench_ntt.err; echo "ntt rc=$?"; cut -c1-160 $O/bench_ntt_exact.json
RTFHE_SKIP_STAGES=1 timeout -k 10 300 python scripts/sweep.py 1,256,512,768,1024,1280,1536,2048,4096,8192 > $O/sweep.log 2>&1; echo "sweep rc=$?"; grep -v amdgpu.ids $O/sweep.log
RTFHE_N=2048 RTFHE_SKIP_STAGES=1 timeout -k 10 300 python scripts/sweep.py 1,256,512,768,1024,2048 > $O/sweep_n2048.log 2>&1; echo "sweep2048 rc=$?"; grep -v amdgpu.ids $O/sweep_n2048.log
RTFHE_BACKEND=ntt RTFHE_SKIP_STAGES=1 timeout -k 10 300 python scripts/sweep.py 1,512,1024 > $O/sweep_ntt.log 2>&1; echo "sweep ntt rc=$?"; grep -v amdgpu.ids $O/sweep_ntt.log
timeout -k 10 300 python scripts/bench_circuit.py > $O/bench_circuit.log 2>&1; echo "circuit rc=$?"; grep -v amdgpu.ids $O/bench_circuit.log | tail -12
bash scripts/profile_n2048.sh eo > $O/profile_n2048.log 2>&1; echo "profile n2048 rc=$?"
cp gpurun_out/profiles_n2048/pmc_n2048_eo.json $O/ 2>/dev/null
