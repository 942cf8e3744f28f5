#!/bin/bash
# A/B of row_gather.hip against a previous version of the file (build/row_gather_prev.hip, made with
# `git show <rev>:mrhyde_amd/csrc/kernels/row_gather.hip`) on one box: perturbed config 2 + the engine configs.
cd $GRAFT_REPO_ROOT
cp mrhyde_amd/csrc/kernels/row_gather.hip /tmp/row_gather_new.hip
run() {
  timeout -k 10 300 python bench.py --mesh perturbed --no-cpu-baseline --steps 10 --warmup 2 > /tmp/bp.log 2>&1
  echo "$1 perturbed $(tail -1 /tmp/bp.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms/step' % d['ms_per_step'])")"
  timeout -k 10 300 python tests/engine_bench.py porous 128 gather 2>&1 | tail -1 | sed "s/^/$1 /"
  timeout -k 10 300 python tests/engine_bench.py ns 32 gather 2>&1 | tail -1 | sed "s/^/$1 /"
}
for r in 1 2; do
  for w in prev new; do
    if [ $w = prev ]; then cp build/row_gather_prev.hip mrhyde_amd/csrc/kernels/row_gather.hip; else cp /tmp/row_gather_new.hip mrhyde_amd/csrc/kernels/row_gather.hip; fi
    make -s -C mrhyde_amd/csrc > /dev/null 2>&1 || { echo "build failed $w"; cp /tmp/row_gather_new.hip mrhyde_amd/csrc/kernels/row_gather.hip; exit 1; }
    run $w
  done
done
cp /tmp/row_gather_new.hip mrhyde_amd/csrc/kernels/row_gather.hip
