#!/bin/bash
# A/B of point-engine build variants inside ONE gpurun call (boxes differ): rebuilds the kernel with
# -DMHA_ENGINE_MINW=<w> for each w and times the three modules.  Usage: bash profiles/engine_ab.sh "2 3 4"
cd $GRAFT_REPO_ROOT
for w in $1; do
  rm -f build/obj/k_point_engine.o
  make -s -C mrhyde_amd/csrc HIPFLAGS="--offload-arch=gfx950 -munsafe-fp-atomics -ffp-contract=fast -DMHA_ENGINE_MINW=$w" > /dev/null 2>&1 || { echo "build failed w=$w"; exit 1; }
  for k in thermal:64 porous:128 ns:16; do
    timeout -k 10 300 python profiles/engine_bench.py ${k%%:*} ${k##*:} gather > /tmp/eb.log 2>&1
    echo "minw=$w $(tail -1 /tmp/eb.log | cut -c1-75)"
  done
done
# Round-1 note: MINW=3 and 4 make the 64-thread variants spill (46-359 VGPRs to scratch) and were slower on thermal;
# the porous run of those two variants aborted once (cause not investigated: the variants are not shipped).  The
# committed default is MINW=2 (no spills).
