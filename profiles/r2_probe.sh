#!/bin/bash
# parity, bench lines (pattern K2: serialised / overlapped with K1; row-block K2), per-workgroup end times
cd "$GRAFT_REPO_ROOT"
MHA_K2=pattern timeout -k 10 600 python -m pytest tests/test_thermal_gpu.py -x -q -k "row_owner" 2>&1 | tail -2
for mode in "MHA_K2=pattern MHA_K1K2_OVERLAP=0" "MHA_K2=pattern MHA_K1K2_OVERLAP=1" "MHA_K2=pattern MHA_K1K2_OVERLAP=2" "MHA_K2=blocks MHA_K1K2_OVERLAP=1" "MHA_K2=blocks MHA_K1=lanes MHA_K1K2_OVERLAP=1"; do
  echo "== $mode"
  env $mode timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])" || exit 1
done
MHA_K2=pattern MHA_K1K2_OVERLAP=0 MHA_BP_TIMING=gpurun_out/bp_timing.bin timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 && python profiles/r2_timing.py gpurun_out/bp_timing.bin
