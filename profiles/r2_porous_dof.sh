#!/bin/bash
# A/Bs of the porousMixed pair at config 3 (128^3).  Element kernel: MHA_GATHER_ORDER=pos = LDS-staged arrays in LID-position
# order, one wavefront per SIMD (before); default = dof order from registers, two wavefronts per SIMD.  Row gather:
# MHA_GATHER_LPR=16 = four rows per wavefront (before); default = eight.
set -e
for mode in pos dof pos dof; do
  echo "== MHA_GATHER_ORDER=$mode"
  MHA_GATHER_ORDER=$mode python bench.py --config 3 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('config 3: ms_per_step %.4f frac %.4f' % (d['ms_per_step'], d['roofline']['frac']))"
done
for lpr in 16 8 16 8; do
  echo "== MHA_GATHER_LPR=$lpr (dof order)"
  MHA_GATHER_LPR=$lpr python bench.py --config 3 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('config 3: ms_per_step %.4f frac %.4f' % (d['ms_per_step'], d['roofline']['frac']))"
done
echo "== before both changes"
MHA_GATHER_LPR=16 MHA_GATHER_ORDER=pos python bench.py --config 3 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('config 3: ms_per_step %.4f frac %.4f' % (d['ms_per_step'], d['roofline']['frac']))"
