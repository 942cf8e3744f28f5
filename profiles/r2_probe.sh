#!/bin/bash
# are partial 128-byte lines what slows the stores down?  MHA_BP_DBG=16: every store covers whole aligned lines (wrong values)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
for mode in "MHA_BP_DBG=0" "MHA_BP_DBG=16"; do
  echo "== $mode"
  env $mode MHA_K1K2_OVERLAP=0 timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])" || exit 1
  env $mode MHA_K1K2_OVERLAP=0 MHA_BP_TIMING=gpurun_out/bp_timing.bin timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 && python profiles/r2_timing.py gpurun_out/bp_timing.bin
done
