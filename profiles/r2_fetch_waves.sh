#!/bin/bash
# A/B of the general row-owner kernel's fetching-wave count (MHA_RG_FETCH_WAVES = 2, 3, 4) on one box:
# perturbed config 2 (64^3 Q2), perturbed 3D Q1 and 2D Q2 timing via bench.py --mesh perturbed.
set -e
mkdir -p gpurun_out
for nf in 4 2 3 4; do
  cp build/ab/nf$nf.so mrhyde_amd/lib/libmrhyde_amd.so
  echo "== fetch waves $nf"
  python bench.py --mesh perturbed --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('config2 perturbed ms_per_step', d['ms_per_step'])"
done
cp build/ab/nf4.so mrhyde_amd/lib/libmrhyde_amd.so
