#!/bin/bash
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r4e4; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_soak.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest2.log &&
timeout -k 10 500 python bench.py --no-cpu-baseline > $O/bench_nocpu.json 2> $O/bench.err; echo "bench rc=$?"; python3 -c "
import json; d=json.loads(open('$O/bench_nocpu.json').read().strip().splitlines()[-1]); print(d['value'], d['secondary']['config5_n2048_1024_gates']['gates_per_s'], d['secondary']['config5_n2048_small_batches'])"
