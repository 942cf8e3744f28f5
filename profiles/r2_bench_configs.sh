#!/bin/bash
# bench.py lines of every BASELINE.json configuration (N = 1)
cd "$GRAFT_REPO_ROOT"
for c in 2 3 5; do
  echo "== config $c"
  timeout -k 10 500 python bench.py --config $c --steps 10 --warmup 2 2>gpurun_out/r2_bench_cfg$c.err | tee gpurun_out/r2_bench_cfg$c.json | cut -c1-1500
  tail -3 gpurun_out/r2_bench_cfg$c.err
done
