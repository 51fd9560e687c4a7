#!/bin/bash
# round 4, closing run: the long soak of every kernel shape (the N = 2048 shapes now on k_bootstrap_eo with half-width trades, the latency shape on
# parity-split transforms: shapes of 1, 64 and 300 gates added), then the two-rank rehearsal of bench.py on the one card
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r4soak; mkdir -p $O
timeout -k 10 900 python - > $O/soak.log 2>&1 <<'PY'
import sys
sys.path.insert(0, "scripts")
import soak
shapes = ((1024, ("fft", "ntt"), (1024, 768, 512, 300, 1280, 64, 1)), (2048, ("fft", "ntt"), (1024, 768, 512, 256, 100, 1)))
bad = soak.run(300, shapes)
print("failures:", bad)
sys.exit(1 if bad else 0)
PY
rc=$?; tail -4 $O/soak.log; [ $rc -eq 0 ] || exit $rc
RTFHE_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 4 --warmup 1 > $O/bench_gpus2_gloo_rehearsal.json 2> $O/bench_gpus2.err; echo "gpus2 rc=$?"; grep "bench.py rank" $O/bench_gpus2.err > $O/bench_gpus2_rank_lines.txt; cut -c1-300 $O/bench_gpus2_gloo_rehearsal.json
