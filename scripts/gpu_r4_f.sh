#!/bin/bash
# round 4, run F: N = 2048 phase stamps and PMC counters, top-bit split vs parity split
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r4f; mkdir -p $O
for k in halves eo; do
  echo "== stamps $k" | tee -a $O/n2048_phase_stamps.log
  RTFHE_N2048_KERNEL=$k timeout -k 10 200 python scripts/ubench/halves_stamps.py 2>&1 | grep -v amdgpu.ids | tee -a $O/n2048_phase_stamps.log
done
for k in halves eo; do
  timeout -k 10 500 bash scripts/profile_n2048.sh $k > $O/profile_$k.log 2>&1; echo "profile $k rc=$?"; tail -3 $O/profile_$k.log
done
cp gpurun_out/profiles_n2048/*.json $O/ 2>/dev/null
