#!/bin/bash
# phase split of the point engine on navierstokes Q2/Q1 at 32^3 (MHA_ENGINE_STOP = leave the element after phase k:
# 1 gather+geometry, 2 fields, 3 point functions (C-hat), 4 residual rows, 5 P panels of the first row tile; 0 = full)
cd $GRAFT_REPO_ROOT
for s in 0 1 2 3 4 5; do
  echo "== MHA_ENGINE_STOP=$s"
  MHA_ENGINE_STOP=$s timeout -k 10 300 python tests/engine_bench.py ns 32 local 2>/dev/null | tail -2 || exit 1
done
