#!/bin/bash
# A/B of two point_engine.hip sources (build/ab/pe_A.hip, build/ab/pe_B.hip) inside ONE gpurun call, interleaved.
cd $GRAFT_REPO_ROOT
for v in A B A B; do
  cp build/ab/pe_$v.hip mrhyde_amd/csrc/kernels/point_engine.hip
  make -s -C mrhyde_amd/csrc > /dev/null 2>&1 || { echo "build failed $v"; exit 1; }
  for k in ${CASES:-thermal:64 porous:128 ns:32}; do
    timeout -k 10 300 python tests/engine_bench.py ${k%%:*} ${k##*:} gather > /tmp/eb.log 2>&1
    echo "$v $(tail -1 /tmp/eb.log | cut -c1-75)"
  done
done
