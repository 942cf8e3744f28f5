#!/bin/bash
# phase ablation of thermal_general_row_owner_kernel on the perturbed config-2 mesh (MHA_GRO_DBG bits: 1 no geometry,
# 2 no residual phase, 4 no products, 8 no LDS adds, 16 no CRS stores); results of the ablated runs are wrong by design
out=gpurun_out/r2_gro_ablate.log
: > $out
for d in 0 1 2 4 8 12 16 28 31; do
  echo "== MHA_GRO_DBG=$d" >> $out
  MHA_GRO_DBG=$d timeout -k 10 200 python bench.py --mesh perturbed --no-cpu-baseline --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel_ms'])" >> $out || exit 1
done
cat $out
