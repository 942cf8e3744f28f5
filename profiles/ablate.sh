#!/bin/bash
# Phase ablations of the row-owner kernel (profiling aid): prints ms/step per MHA_DEBUG_SKIP mask.
for m in 0 1 2 4 8 16 3 7 15 31; do
  MHA_DEBUG_SKIP=$m python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('skip=$m', 'ms/step=%.3f'%d['ms_per_step'], 'kernel_ms=%.3f'%d['roofline']['kernel_ms'])"
done
