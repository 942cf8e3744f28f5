#!/bin/bash
# round 2: block-pattern K2 and thread-per-element K1: parity tests, bench lines with ablations, per-wave timing
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_thermal_gpu.py -x -q -k "row_owner" > gpurun_out/r2_first_tests.log 2>&1 || { tail -30 gpurun_out/r2_first_tests.log; exit 1; }
tail -3 gpurun_out/r2_first_tests.log
for mode in "MHA_VERBOSE=1 MHA_K1K2_OVERLAP=1" "MHA_K1K2_OVERLAP=0" "MHA_K2=blocks MHA_K1K2_OVERLAP=0" "MHA_BP_DBG=1 MHA_K1K2_OVERLAP=0" "MHA_BP_DBG=2 MHA_K1K2_OVERLAP=0" "MHA_BP_DBG=4 MHA_K1K2_OVERLAP=0" "MHA_BP_DBG=6 MHA_K1K2_OVERLAP=0"; do
  echo "== $mode" 
  env $mode timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>gpurun_out/r2_first_err.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])" || exit 1
  grep mrhyde_amd gpurun_out/r2_first_err.log
done
MHA_K1K2_OVERLAP=0 MHA_BP_TIMING=gpurun_out/bp_timing.bin timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 && python profiles/r2_timing.py gpurun_out/bp_timing.bin
