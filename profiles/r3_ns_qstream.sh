#!/bin/bash
# config 4: q-streamed Jacobian phase of the point engine against the panel-by-panel form (MHA_ENGINE_STOP=32), one box
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3b; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "navierstokes or ns or config4 or engine" > $O/ns_tests.log 2>&1; tail -3 $O/ns_tests.log
for v in ${STOPS:-0 32}; do
  echo -n "MHA_ENGINE_STOP=$v, 32^3: "
  MHA_ENGINE_STOP=$v timeout -k 10 300 python bench.py --config 4 --ncell 32 --steps 10 --warmup 3 --no-cpu-baseline 2>$O/ns_q_$v.err | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel_ms'])"
done
for v in ${STOPS64:-0 32}; do
  echo -n "MHA_ENGINE_STOP=$v, 64^3: "
  MHA_ENGINE_STOP=$v timeout -k 10 400 python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline 2>$O/ns_q64_$v.err | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel_ms'])"
done
