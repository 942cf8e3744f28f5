#!/bin/bash
# round 3: geometry-database mode of the Jacobian kernel (one representative block per pattern + replication) against the
# full kernel (MHA_BP_DATABASE=0); per-kernel times from rocprofv3
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  rm -rf /tmp/prof_db
  MHA_BP_DATABASE=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_db -o p -- python3 $R/bench.py --no-cpu-baseline --steps 20 > /tmp/db_$v.json 2>/dev/null
  f=$(find /tmp/prof_db -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$v" /tmp/db_$v.json <<'PY'
import csv,sys,json
d=json.loads(open(sys.argv[3]).readlines()[-1])
print('MHA_BP_DATABASE=%s: ms_per_step %.4f frac %.3f' % (sys.argv[2], d['ms_per_step'], d['roofline']['frac']))
for r in csv.DictReader(open(sys.argv[1])):
    if 'mha' in r['Name'] and int(r['Calls'])>5: print('   %-80s calls %4s avg %9.1f us' % (r['Name'][:80], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
