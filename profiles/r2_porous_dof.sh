#!/bin/bash
# A/B of the porousMixed element kernel behind the row gather at config 3 (128^3): MHA_GATHER_ORDER=pos = LDS-staged arrays in
# LID-position order, one wavefront per SIMD (before); default = dof order from registers, two wavefronts per SIMD
set -e
for mode in pos dof pos dof; do
  echo "== MHA_GATHER_ORDER=$mode"
  MHA_GATHER_ORDER=$mode python bench.py --config 3 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('config 3: ms_per_step %.4f frac %.4f' % (d['ms_per_step'], d['roofline']['frac']))"
done
