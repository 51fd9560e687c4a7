#!/bin/bash
# GPU validation pass A (round 2): full GPU test suite, default bench, multi-rank rehearsal, config 3 on one GPU, host-pointer rate
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
mkdir -p gpurun_out/r02a
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02a/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r02a/pytest_gpu.log
tail -5 gpurun_out/r02a/pytest_gpu.log
timeout -k 10 300 python bench.py > gpurun_out/r02a/bench.json 2> gpurun_out/r02a/bench.err; echo "bench rc=$?"; cat gpurun_out/r02a/bench.json | cut -c1-1500
RTFHE_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 4 --warmup 1 > gpurun_out/r02a/bench_gpus2_gloo.json 2> gpurun_out/r02a/bench_gpus2_gloo.err; echo "bench gpus2 rc=$?"; cut -c1-600 gpurun_out/r02a/bench_gpus2_gloo.json
RTFHE_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --workload config3 --gates 2048 --steps 3 --warmup 1 > gpurun_out/r02a/bench_config3_gloo2.json 2> gpurun_out/r02a/bench_config3_gloo2.err; echo "bench config3 gloo2 rc=$?"; cut -c1-900 gpurun_out/r02a/bench_config3_gloo2.json
timeout -k 10 300 python bench.py --workload config3 --steps 3 --warmup 1 > gpurun_out/r02a/bench_config3_1gpu.json 2> gpurun_out/r02a/bench_config3_1gpu.err; echo "bench config3 1gpu rc=$?"; cut -c1-900 gpurun_out/r02a/bench_config3_1gpu.json
timeout -k 10 300 python scripts/host_rate.py > gpurun_out/r02a/host_rate.log 2>&1; echo "host_rate rc=$?"; cat gpurun_out/r02a/host_rate.log
