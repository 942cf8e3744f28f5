#!/bin/bash
# K1 with / without the in-register merge of x-neighbour face nodes (MHA_K1_DBG bit 8 switches it off), one box
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_thermal_gpu.py -x -q -k "row_owner or auto_path or deterministic" 2>&1 | tail -2 || exit 1
for d in 0 8 0 8; do
  echo "== MHA_K1_DBG=$d MHA_K1K2_OVERLAP=0"
  MHA_K1_DBG=$d MHA_K1K2_OVERLAP=0 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('ms_per_step %.4f kernel_ms %.4f frac %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))" || exit 1
done
echo "== default (overlap)"; timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('ms_per_step %.4f kernel_ms %.4f frac %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))"
