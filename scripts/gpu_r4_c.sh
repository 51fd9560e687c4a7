#!/bin/bash
# round 4, run C: early-Q priority grid (N = 1024), N = 2048 variants, then the whole GPU suite and the bench lines
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r4c; mkdir -p $O
echo "== N=1024 gates=1024 early-Q grid" | tee -a $O/ab_earlyq.log
timeout -k 10 500 python scripts/ab_libs.py 1024 5 build/ab/p_base.so build/ab/p_eq_2_10.so build/ab/p_eq_1_10.so build/ab/p_eq_2_9.so build/ab/p_eq_1_9.so build/ab/p_eq_5_10.so build/ab/p_eq_2_8.so build/ab/p_eq_2_7.so build/ab/p_eq_flat.so 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_earlyq.log
for g in 1024 768; do
  echo "== N=2048 gates=$g" | tee -a $O/ab_n2048.log
  RTFHE_N=2048 timeout -k 10 400 python scripts/ab_libs.py $g 5 build/ab/n_base.so build/ab/n_prio0.so build/ab/n_prio2.so build/ab/n_g2.so 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_n2048.log
done
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-400 $O/bench.json
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/bench_steps20_warmup5.json 2> $O/bench20.err; echo "bench20 rc=$?"; cut -c1-300 $O/bench_steps20_warmup5.json
