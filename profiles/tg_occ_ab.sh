#!/bin/bash
# A/B of thermal_general.hip occupancy variants on the perturbed config-2 mesh, one box, interleaved:
# "EPB QM MINW" = waves per workgroup, points per MFMA chunk, waves per SIMD the register budget is cut for.
cd $GRAFT_REPO_ROOT
for r in 1 2; do
for cfg in "4 8 2" "8 4 2" "12 4 3" "6 4 3"; do
  set -- $cfg
  rm -f build/obj/k_thermal_general.o
  make -s -C mrhyde_amd/csrc HIPFLAGS="--offload-arch=gfx950 -munsafe-fp-atomics -ffp-contract=fast -DMHA_TG_EPB=$1 -DMHA_TG_QM=$2 -DMHA_TG_MINW=$3" > /dev/null 2>&1 || { echo "build failed $cfg"; continue; }
  timeout -k 10 300 python bench.py --mesh perturbed --no-cpu-baseline --steps 10 --warmup 2 > /tmp/bp.log 2>&1
  echo "EPB=$1 QM=$2 MINW=$3 $(tail -1 /tmp/bp.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms/step' % d['ms_per_step'])" 2>/dev/null || tail -2 /tmp/bp.log)"
done
done
rm -f build/obj/k_thermal_general.o
make -s -C mrhyde_amd/csrc > /dev/null 2>&1
