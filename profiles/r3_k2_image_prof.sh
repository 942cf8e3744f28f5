#!/bin/bash
# per-kernel durations (rocprofv3 --kernel-trace --stats) of the two block-pattern launches
mkdir -p gpurun_out/r3
cd /tmp && export TMPDIR=/tmp
for d in 0 4 2 6; do
  MHA_BP_IMAGE=1 MHA_BP_DBG=$d MHA_K1K2_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3/prof_img_$d -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 20 > /dev/null 2>&1
  echo "== MHA_BP_DBG=$d"
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/r3/prof_img_$d -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if 'mha' in n: print('%-90s calls %4s avg %9.1f us' % (n[:90], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
