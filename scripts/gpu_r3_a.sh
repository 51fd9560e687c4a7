#!/bin/bash
# round 3, first GPU pass: LDS-form / issue-rate ubench, same-process A/B of the pair-kernel variants, bench line of the tree as it stands
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3a; mkdir -p $O
timeout -k 10 120 scripts/ubench/lds_forms > $O/lds_forms.log 2>&1; echo "ubench rc=$?"; cat $O/lds_forms.log
timeout -k 10 400 python scripts/ab_libs.py 1024 6 $(ls build/ab/p_*.so) 2>&1 | grep -v amdgpu.ids | tee $O/ab.log
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-300 $O/bench.json
