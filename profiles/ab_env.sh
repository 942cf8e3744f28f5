#!/bin/bash
# A/B of environment settings on bench.py inside ONE gpurun call, interleaved, wall-clock ms/step (no profiler).
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  for cfg in "$@"; do
    if [ "$cfg" = "-" ]; then envs=""; else envs="$cfg"; fi
    env $envs timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline ${BENCH_ARGS} > /tmp/abe.log 2>&1
    echo "round $r [$cfg] $(tail -1 /tmp/abe.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms/step  kernel_ms %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))")"
  done
done
