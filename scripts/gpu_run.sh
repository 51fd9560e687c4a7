#!/bin/bash
# One parametrised GPU-box runner (round 5; replaces the per-experiment scripts/gpu_*.sh of rounds 2-4).
#   usage (through gpurun):  bash scripts/gpu_run.sh <tag> <step> [<step> ...]      outputs under gpurun_out/<tag>/
# steps:
#   tests            the GPU suite (pytest -m gpu, one process)
#   testfile:<file>  one file of it
#   smoke            __graft_entry__.smoke()
#   bench            python bench.py (default flags: headline + cpu baseline + secondary)
#   bench20          python bench.py --steps 20 --warmup 5 (the driver's flags)
#   config3[:D]      bench.py --workload config3 on one GPU (RCCL harness); with D: --workload config3-c-abi over D entries naming GPU 0 (C-ABI sharding)
#   ntt              bench.py --backend ntt-exact
#   sweep[:N[:backend]]   scripts/sweep.py over the usual batch sizes (N = 1024 default, 2048; backend fft|ntt|xfft)
#   circuit          scripts/bench_circuit.py (the adder netlists)
#   profile          rocprofv3 kernel trace + PMC passes of the headline on the driver's flags, --steps 20 --warmup 5 (scripts/profile_gpu.sh r06)
#   profile2048      counters of the N = 2048 kernel (scripts/profile_n2048.sh eo)
#   pmc:<kernel>:<gates>[:N[:matrix|cache|plain[:fft|ntt|xfft]]]   counters of one kernel family on a batch (scripts/profile_kernel.sh)
#   ab:<N>:<gates>:<rounds>:<lib>[,<lib>...]   same-process A/B of builds under build/ab/ ("shipped" = rustfhe_amd/librtfhe_hip.so)
#   ubench:<name>    scripts/ubench/<name> (a prebuilt micro-benchmark binary)
#   clock:<name>     the shader clock scripts/ubench/<name>'s kernels ran at, launch by launch (scripts/kernel_clock.sh)
#   keycache         FETCH_SIZE / time of a tail launch alone and behind a full round (scripts/profile_keycache.sh)
#   soak             scripts/soak.py
#   py:<script>[:args,comma,separated]        any python script of scripts/
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export RTFHE_BENCH_STRICT=1     # bench.py: a measuring process that dies in a side leg is exit code 86 here, not 0 (the line is printed either way)
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
# A step that was KILLED (timeout, signal) stops the run: no GPU step is started behind a dead one.  A step that merely failed (a red test, a
# non-zero exit of its own) is reported, the run goes on and ends non-zero.
FAILED=0
run() { local name=$1; shift; echo "== $name: $*"; "$@"; local rc=$?; echo "== $name rc=$rc"; if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then exit $rc; fi; [ $rc -eq 0 ] || FAILED=$rc; }
for step in "$@"; do
  IFS=: read -r kind a1 a2 a3 a4 a5 <<< "$step"
  case $kind in
    testfile) run "tests $a1" bash -c "timeout -k 10 900 python -m pytest tests/$a1 -m gpu -x -q > $O/pytest_$a1.log 2>&1; rc=\$?; tail -15 $O/pytest_$a1.log; exit \$rc" ;;
    tests)   run tests bash -c "timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=\$?; tail -3 $O/pytest_gpu.log; exit \$rc" ;;
    smoke)   run smoke bash -c "timeout -k 10 300 python -c 'import __graft_entry__ as g; g.smoke()' > $O/smoke.log 2>&1; rc=\$?; tail -2 $O/smoke.log; exit \$rc" ;;
    bench)   run bench bash -c "timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err; rc=\$?; cut -c1-400 $O/bench.json; exit \$rc" ;;
    bench20) run bench20 bash -c "timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench_steps20_warmup5.json 2> $O/bench20.err; rc=\$?; cut -c1-400 $O/bench_steps20_warmup5.json; exit \$rc" ;;
    config3) if [ -n "$a1" ]; then devs=$(python3 -c "print(','.join(['0'] * $a1))")
               run "config3 c-abi x$a1" bash -c "timeout -k 10 300 python bench.py --workload config3-c-abi --devices $devs --steps 3 --warmup 1 > $O/bench_config3_c_abi_$a1.json 2> $O/bench_config3_c_abi.err; rc=\$?; cut -c1-400 $O/bench_config3_c_abi_$a1.json; exit \$rc"
             else run config3 bash -c "timeout -k 10 300 python bench.py --workload config3 --steps 3 --warmup 1 > $O/bench_config3_1gpu.json 2> $O/bench_config3.err; rc=\$?; cut -c1-300 $O/bench_config3_1gpu.json; exit \$rc"; fi ;;
    ntt)     run ntt bash -c "timeout -k 10 300 python bench.py --backend ntt-exact --no-cpu-baseline > $O/bench_ntt_exact.json 2> $O/bench_ntt.err; rc=\$?; cut -c1-300 $O/bench_ntt_exact.json; exit \$rc" ;;
    sweep)   n=${a1:-1024}; be=${a2:-fft}; sizes=1,256,512,768,1024,1280,1536,2048,4096,8192; [ $n = 2048 ] && sizes=1,256,512,768,1024,2048; [ $be = ntt ] && sizes=1,512,1024; [ $be = xfft ] && sizes=1,256,512,768,1024,2048,4096
             run sweep env RTFHE_N=$n RTFHE_BACKEND=$be RTFHE_SKIP_STAGES=1 bash -c "timeout -k 10 300 python scripts/sweep.py $sizes > $O/sweep_N${n}_$be.log 2>&1; rc=\$?; grep -v amdgpu.ids $O/sweep_N${n}_$be.log; exit \$rc" ;;
    circuit) run circuit bash -c "timeout -k 10 300 python scripts/bench_circuit.py > $O/bench_circuit.log 2>&1; rc=\$?; grep -v amdgpu.ids $O/bench_circuit.log | tail -12; exit \$rc" ;;
    profile) run profile bash -c "bash scripts/profile_gpu.sh r06 > $O/profile.log 2>&1; rc=\$?; tail -3 $O/profile.log; cp -r gpurun_out/profiles_r06 $O/ 2>/dev/null; exit \$rc" ;;
    profile2048) run profile2048 bash -c "bash scripts/profile_n2048.sh ${a1:-eo} > $O/profile_n2048.log 2>&1; rc=\$?; tail -3 $O/profile_n2048.log; cp gpurun_out/profiles_n2048/pmc_n2048_${a1:-eo}.json $O/ 2>/dev/null; exit \$rc" ;;
    pmc)     run "pmc $a1" env RTFHE_BACKEND=${a5:-fft} bash -c "bash scripts/profile_kernel.sh $a1 $a2 ${a3:-1024} $a4 > $O/pmc_$a1.log 2>&1; rc=\$?; tail -60 $O/pmc_$a1.log; cp gpurun_out/pmc_kernel/pmc_${a1}_N${a3:-1024}_g$a2.json $O/ 2>/dev/null; exit \$rc" ;;
    ab)      libs=$(echo "$a4" | tr ',' ' ' | sed -E 's#(^| )shipped#\1rustfhe_amd/librtfhe_hip.so#g; s#(^| )([A-Za-z0-9_]+)( |$)#\1build/ab/\2.so\3#g; s#(^| )([A-Za-z0-9_]+)( |$)#\1build/ab/\2.so\3#g')
             first=$(echo $libs | cut -d" " -f1)
             run "ab N=$a1 gates=$a2" env RTFHE_N=$a1 RTFHE_LIB=$first bash -c "timeout -k 10 400 python scripts/ab_libs.py $a2 $a3 $libs 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_N${a1}_g${a2}.log" ;;
    ubench)  run "ubench $a1" bash -c "timeout -k 10 300 scripts/ubench/$a1 > $O/ubench_$a1.log 2>&1; rc=\$?; cat $O/ubench_$a1.log; exit \$rc" ;;
    clock)   run "clock $a1" bash -c "bash scripts/kernel_clock.sh $REPO/scripts/ubench/$a1 > $O/clock_$a1.log 2>&1; rc=\$?; tail -70 $O/clock_$a1.log; cp gpurun_out/clock/$a1.json $O/clock_$a1.json 2>/dev/null; exit \$rc" ;;
    keycache) run keycache bash -c "bash scripts/profile_keycache.sh > $O/keycache.log 2>&1; rc=\$?; tail -40 $O/keycache.log; cp gpurun_out/keycache/keycache.json $O/ 2>/dev/null; exit \$rc" ;;
    soak)    run soak bash -c "timeout -k 10 600 python scripts/soak.py > $O/soak.log 2>&1; rc=\$?; tail -5 $O/soak.log; exit \$rc" ;;
    py)      args=$(echo "$a2" | tr ',' ' '); run "py $a1" bash -c "timeout -k 10 500 python scripts/$a1 $args > $O/$(basename $a1 .py).log 2>&1; rc=\$?; grep -v amdgpu.ids $O/$(basename $a1 .py).log | tail -40; exit \$rc" ;;
    *) echo "unknown step $step"; exit 64 ;;
  esac
done
exit $FAILED
