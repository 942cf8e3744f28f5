#!/bin/bash
# A/B of two row_gather.hip sources (build/ab/rg_A.hip, rg_B.hip) inside ONE gpurun call, interleaved.
cd $GRAFT_REPO_ROOT
for v in A B A B; do
  cp build/ab/rg_$v.hip mrhyde_amd/csrc/kernels/row_gather.hip
  make -s -C mrhyde_amd/csrc > /dev/null 2>&1 || { echo "build failed $v"; exit 1; }
  timeout -k 10 300 python bench.py --mesh perturbed --no-cpu-baseline --steps 10 --warmup 2 > /tmp/bp.log 2>&1
  echo "$v perturbed thermal 64^3: $(tail -1 /tmp/bp.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms/step' % d['ms_per_step'])")"
  for k in porous:128 ns:32; do
    timeout -k 10 300 python tests/engine_bench.py ${k%%:*} ${k##*:} gather > /tmp/eb.log 2>&1
    echo "$v $(tail -1 /tmp/eb.log | cut -c1-75)"
  done
done
