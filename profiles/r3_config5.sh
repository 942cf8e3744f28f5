#!/bin/bash
# round 3: config 5 with the fused HDG element kernel: tests, bench line, per-kernel durations
python -m pytest tests/test_swhdg_gpu.py -x -q 2>&1 | tail -2
python bench.py --config 5 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('config 5: ms_per_step %.4f kernel_ms %.4f frac %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3/prof_c5 -o p -- python3 $GRAFT_REPO_ROOT/bench.py --config 5 --no-cpu-baseline --steps 20 > /dev/null 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/r3/prof_c5 -name "*kernel_stats.csv" | head -1)
cp $f $GRAFT_REPO_ROOT/gpurun_out/r3/config5_kernel_stats.csv
python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if 'mha' in n and int(r['Calls'])>5: print('%-100s calls %4s avg %9.1f us' % (n[:100], r['Calls'], float(r['AverageNs'])/1e3))
PY
