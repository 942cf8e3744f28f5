#!/bin/bash
# Phase ablation of thermal_general_element_kernel (-DMHA_TG_STOP=k) on the perturbed config-2 mesh: HIP-event time of
# the element kernel + row gather with the kernel leaving after phase k.  Leaves the default build behind.
cd $GRAFT_REPO_ROOT
for w in 1 2 3 4 9; do
  rm -f build/obj/k_thermal_general.o
  make -s -C mrhyde_amd/csrc HIPFLAGS="--offload-arch=gfx950 -munsafe-fp-atomics -ffp-contract=fast -DMHA_TG_STOP=$w" > /dev/null 2>&1 || { echo "build failed $w"; exit 1; }
  ls -la --time-style=full-iso mrhyde_amd/lib/libmrhyde_amd.so | awk '{print $6, $7}'
  timeout -k 10 300 python bench.py --mesh perturbed --no-cpu-baseline --steps 10 --warmup 2 > /tmp/bp.log 2>&1
  echo "STOP=$w $(tail -1 /tmp/bp.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms/step  kernel_ms %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms']))")"
done
