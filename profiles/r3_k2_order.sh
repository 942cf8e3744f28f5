#!/bin/bash
# round 3: order of a role's blocks in the block-pattern kernel: partition (Morton) order against CRS order dealt round
# robin over the workgroups; MHA_BP_DBG=4: stores only (no products)
for o in morton interleaved; do for d in 0 4; do
  echo "== MHA_BP_ORDER=$o MHA_BP_DBG=$d"
  MHA_BP_ORDER=$o MHA_BP_DBG=$d MHA_K1K2_OVERLAP=0 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('ms_per_step %.4f kernel_ms %.4f frac %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))"
done; done
python -m pytest tests/test_thermal_gpu.py tests/test_full_size_gpu.py -x -q -m gpu -k "row_owner or affine or config2 or auto_path" 2>&1 | tail -3
