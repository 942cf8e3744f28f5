#!/bin/bash
# A/B of the K1 / K2 launch arrangement at config 2 with the final kernels: MHA_K1K2_OVERLAP=1 (K2 on the context's stream,
# K1 on a side stream, the default so far), 2 (K1 launched first), 0 (one stream, K1 then K2)
set -e
for ov in 1 0 2 1 0 2; do
  echo "== MHA_K1K2_OVERLAP=$ov"
  MHA_K1K2_OVERLAP=$ov python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('config 2: ms_per_step %.4f kernel_ms %.4f frac %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))"
done
