#!/bin/bash
# round 4, run U: the latency kernel -- each inverse transform on two waves (parity split; wg_r0) and rows 4, 5 of the F phase on two waves each
# (the library) against one wave per transform with the head -> tail hand-off (wg_i1); phase stamps; the parity tests
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r4u; mkdir -p $O
LIB=rustfhe_amd/librtfhe_hip.so
for g in 1 64 256; do
  echo "== N=1024 gates=$g" | tee -a $O/ab_wg.log
  timeout -k 10 300 python scripts/ab_libs.py $g 7 build/ab/wg_i1.so build/ab/wg_r0.so $LIB 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_wg.log
done &&
RTFHE_STAMPS_LIB_PATH=$PWD/build/ab/wgstamps.so timeout -k 10 120 python scripts/ubench/wg_stamps.py 2>&1 | grep -v amdgpu.ids | tee -a $O/wg_stamps.log &&
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_soak.py -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest.log
