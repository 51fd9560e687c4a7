#!/bin/bash
# paired (compiler default: ds_read2_b64 / ds_write2_b64) vs unpaired LDS exchange accesses for every kernel family
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/lds_ab; mkdir -p $O
for cfg in "1024 fft" "1024 ntt" "2048 fft" "2048 ntt"; do
  set -- $cfg
  echo "== N=$1 backend=$2" | tee -a $O/ab.log
  RTFHE_N=$1 RTFHE_BACKEND=$2 RTFHE_KS_MM_MIN=0 timeout -k 10 300 python scripts/ab_libs.py 1024 4 build/ab/p_paired.so build/ab/p_unpaired.so 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.log
done
echo "== N=1024 fft single gate / 256 gates (wg kernel)" | tee -a $O/ab.log
for g in 1 256; do RTFHE_KS_MM_MIN=0 timeout -k 10 300 python scripts/ab_libs.py $g 4 build/ab/p_paired.so build/ab/p_unpaired.so 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.log; done
