#!/bin/bash
# K1 (thread per element) after the register diet: parity + kernel-level durations
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_thermal_gpu.py -x -q -k "row_owner" 2>&1 | tail -2
export MHA_K1K2_OVERLAP=0
i=0
for mode in "MHA_K1_DBG=0" "MHA_K1_DBG=1" "MHA_K1_DBG=2"; do
  i=$((i+1))
  echo "== $mode"
  env $mode bash profiles/kstats.sh r2p$i bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  python - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/r2p${i}_kernel_stats.csv")):
    if "residual" in r["Name"] or "affine_element" in r["Name"] or "jacobian" in r["Name"]:
        print("  %-60s %4s calls  %9.1f us" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
