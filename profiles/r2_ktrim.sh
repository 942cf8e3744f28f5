#!/bin/bash
# block-pattern Jacobian kernel with / without the k-step trim of the 14 x 4 units (MHA_BP_DBG bit 16 switches it off), one box
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_thermal_gpu.py tests/test_full_size_gpu.py -x -q -k "row_owner or auto_path or deterministic or config2" 2>&1 | tail -2 || exit 1
for d in 0 16 0 16; do
  echo "== MHA_BP_DBG=$d"
  MHA_BP_DBG=$d timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('ms_per_step %.4f kernel_ms %.4f frac %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))" || exit 1
done
