#!/bin/bash
# round 3: workgroup size of the merged K1 (256 default, variants built with -DMHA_K1_THREADS=128 / 64)
R=$GRAFT_REPO_ROOT
cp $R/mrhyde_amd/lib/libmrhyde_amd.so /tmp/lib_default.so
for v in default k1t128 k1t64 default; do
  if [ $v = default ]; then cp /tmp/lib_default.so $R/mrhyde_amd/lib/libmrhyde_amd.so; else cp $R/build/lib_$v.so $R/mrhyde_amd/lib/libmrhyde_amd.so; fi
  echo "== $v"; python3 $R/profiles/r3_k1_alone.py 2>/dev/null | tail -2
done
cp /tmp/lib_default.so $R/mrhyde_amd/lib/libmrhyde_amd.so
