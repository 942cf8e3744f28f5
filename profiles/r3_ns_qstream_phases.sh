#!/bin/bash
# config 4 at 32^3: phase ablation of the q-streamed point engine (MHA_ENGINE_STOP), ms per assembly incl. the row gather
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3b; mkdir -p $O
for v in ${STOPS:-0 1 2 3 4 5 6 7 16 32}; do
  echo -n "MHA_ENGINE_STOP=$v: "
  MHA_ENGINE_STOP=$v timeout -k 10 300 python bench.py --config 4 --ncell 32 --steps 10 --warmup 3 --no-cpu-baseline 2>$O/ns_ph_$v.err | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'])"
done
