#!/bin/bash
# round 4, run D: N = 2048 variants (priority of half B, gather on byte offsets, pass 1 under the last row's wait, opaque wait) + bench with the quota-aware CPU child
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r4d; mkdir -p $O
for g in 1024 768; do
  echo "== N=2048 gates=$g" | tee -a $O/ab_n2048.log
  RTFHE_N=2048 timeout -k 10 500 python scripts/ab_libs.py $g 5 build/ab/n_base.so build/ab/n_prio0.so build/ab/n_prio2.so build/ab/n_g2.so build/ab/n_ep1.so build/ab/n_ow.so build/ab/n_ep1_ow.so 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_n2048.log
done
timeout -k 10 600 python bench.py --no-secondary > $O/bench_nosec.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-300 $O/bench_nosec.json
