#!/bin/bash
# config 4 at its stated size: the 64^3 navierstokes test and its bench line
cd "$GRAFT_REPO_ROOT"
( while true; do sleep 60; echo "[alive $(date +%T)]"; done ) &
KA=$!
timeout -k 10 1000 python -m pytest tests/test_full_size_gpu.py -x -q -k "config4" 2>&1 | tail -5
timeout -k 10 900 python bench.py --config 4 --steps 5 --warmup 1 2>gpurun_out/r2_bench_cfg4.err | tee gpurun_out/r2_bench_cfg4.json | cut -c1-1600
tail -4 gpurun_out/r2_bench_cfg4.err
kill $KA
