#!/bin/bash
# after a device-header change: GPU suite, soak, the two headline bench lines, NTT sweep at N = 2048, and the rocprofv3 passes (new PMC stamp)
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/restamp; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log
timeout -k 10 600 python scripts/soak.py 200 2>&1 | grep -v amdgpu.ids > $O/soak.log; echo "soak rc=$?"; tail -1 $O/soak.log
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/bench_steps20_warmup5.json 2> $O/bench20.err; echo "bench20 rc=$?"; cut -c1-200 $O/bench_steps20_warmup5.json
RTFHE_N=2048 RTFHE_BACKEND=ntt RTFHE_SKIP_STAGES=1 timeout -k 10 300 python scripts/sweep.py 1,256,512,1024,2048 > $O/sweep_ntt_n2048.log 2>&1; echo "sweep ntt2048 rc=$?"; grep -v amdgpu.ids $O/sweep_ntt_n2048.log
bash scripts/profile_gpu.sh r03 > $O/profile.log 2>&1; echo "profile rc=$?"; tail -3 $O/profile.log
