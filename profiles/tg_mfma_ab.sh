#!/bin/bash
# A/B of thermal_general.hip Jacobian product: matrix cores (-DMHA_TG_MFMA=1) against 4x4 register tiles (=0) on the
# perturbed config-2 mesh, one box, interleaved.  Leaves the default build behind.
cd $GRAFT_REPO_ROOT
for r in 1 2; do
for w in 0 1; do
  rm -f build/obj/k_thermal_general.o
  make -s -C mrhyde_amd/csrc HIPFLAGS="--offload-arch=gfx950 -munsafe-fp-atomics -ffp-contract=fast -DMHA_TG_MFMA=$w" > /dev/null 2>&1 || { echo "build failed $w"; exit 1; }
  timeout -k 10 300 python bench.py --mesh perturbed --no-cpu-baseline --steps 10 --warmup 2 > /tmp/bp.log 2>&1
  echo "MFMA=$w $(tail -1 /tmp/bp.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms/step  kernel_ms %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms']))")"
done
done
