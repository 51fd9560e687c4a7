#!/bin/bash
# round 3: GPU tests + the sweeps of every kernel family (tag = output directory under gpurun_out/)
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${1:-r3b}; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-200 $O/bench.json
RTFHE_SKIP_STAGES=1 timeout -k 10 300 python scripts/sweep.py 1,256,512,768,1024,1280,2048,4096 > $O/sweep.log 2>&1; echo "sweep rc=$?"; grep -v amdgpu.ids $O/sweep.log
RTFHE_N=2048 RTFHE_SKIP_STAGES=1 timeout -k 10 300 python scripts/sweep.py 1,1024,2048 > $O/sweep_n2048.log 2>&1; echo "sweep2048 rc=$?"; grep -v amdgpu.ids $O/sweep_n2048.log
RTFHE_BACKEND=ntt RTFHE_SKIP_STAGES=1 timeout -k 10 300 python scripts/sweep.py 1,256,1024 > $O/sweep_ntt.log 2>&1; echo "sweep ntt rc=$?"; grep -v amdgpu.ids $O/sweep_ntt.log
RTFHE_N=2048 RTFHE_BACKEND=ntt RTFHE_SKIP_STAGES=1 timeout -k 10 300 python scripts/sweep.py 1,1024 > $O/sweep_ntt_n2048.log 2>&1; echo "sweep ntt2048 rc=$?"; grep -v amdgpu.ids $O/sweep_ntt_n2048.log
