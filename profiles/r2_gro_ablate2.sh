#!/bin/bash
out=gpurun_out/r2_gro_ablate2.log
: > $out
for d in 31 63 32 36 48; do
  echo "== MHA_GRO_DBG=$d" >> $out
  MHA_GRO_DBG=$d timeout -k 10 200 python bench.py --mesh perturbed --no-cpu-baseline --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel_ms'])" >> $out || exit 1
done
cat $out
