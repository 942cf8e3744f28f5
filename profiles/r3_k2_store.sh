#!/bin/bash
# round 3: block-pattern Jacobian kernel with paired-tile 16-byte stores; ablations on one box
# (MHA_BP_DBG: 2 no stores, 4 no products, 16 no k-step trim)
mkdir -p gpurun_out/r3
python -m pytest tests/test_thermal_gpu.py tests/test_full_size_gpu.py -x -q -m gpu -k "row_owner or affine or config2 or deterministic or auto_path" 2>&1 | tail -3
for d in 0 2 4 6 16 0; do
  echo "== MHA_BP_DBG=$d"
  MHA_BP_DBG=$d MHA_K1K2_OVERLAP=0 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('ms_per_step %.4f kernel_ms %.4f frac %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))"
done
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3/prof_k2 -o k2 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 20 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT; find gpurun_out/r3/prof_k2 -name "*kernel_stats.csv" | head -1 | xargs head -8 | cut -c1-200
