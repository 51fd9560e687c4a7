#!/bin/bash
# compiler-reported resources + DP/LDS instruction mix of k_bootstrap_pair<...,4> for a set of -D flags (CPU only)
# usage: res.sh name [-DX ...]
name=$1; shift
mkdir -p /tmp/isa/$name && cd /tmp/isa/$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math --cuda-device-only -S "$@" -Rpass-analysis=kernel-resource-usage -x hip /root/repo/rustfhe_amd/csrc/rtfhe_api.hip -o api.s 2> remarks.txt
grep -A12 "Function Name: _ZN5rtfhe16k_bootstrap_pairILi3ELi6ELi8ELi2ELi3ELi4E" remarks.txt | grep -E "VGPRs:|Spill|ScratchSize|Occupancy|SGPRs:" | sed 's/.*remark: //' | tr '\n' ';'; echo
python3 /root/repo/scripts/isa/hist.py api.s k_bootstrap_pairILi3ELi6ELi8ELi2ELi3ELi4E | grep TOTAL
