#!/bin/bash
# round 3: the deck-string NaN of commit ea45dcc (point engine, __noinline__ interpreter), rebuilt in build/old_tree:
# A as committed, B point_engine.hip compiled with -mllvm -enable-ipra=false (no interprocedural register allocation:
# the callee saves its callee-saved registers itself), C the interpreter's stack zero-initialised.  ONE run of each.
cd build/old_tree
for v in A B C; do
  cp mrhyde_amd/lib/lib_$v.so mrhyde_amd/lib/libmrhyde_amd.so
  echo "== variant $v"
  timeout -k 10 300 python -m pytest tests/test_multi_gpu.py -x -q -k "deck_string" 2>&1 | tail -6
done
