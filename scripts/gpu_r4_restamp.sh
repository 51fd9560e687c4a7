#!/bin/bash
# after the last change to a device header: smoke(), the PMC / kernel-trace profile of the headline again (the traffic stamp carries a hash of
# rustfhe_amd/csrc/*.hpp), then two bench lines that find the stamp matching
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/restamp4; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
bash scripts/profile_gpu.sh r04 > $O/profile.log 2>&1; echo "profile rc=$?"; tail -2 $O/profile.log
cp gpurun_out/profiles_r04/pmc_traffic.json profiles/pmc_traffic.json
cp -r gpurun_out/profiles_r04 $O/
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > $O/bench_steps20_warmup5.json 2> $O/bench20.err; echo "bench20 rc=$?"; cut -c1-200 $O/bench_steps20_warmup5.json
timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-200 $O/bench.json
