#!/bin/bash
# round 3: config 3 (porousMixed 128^3): direct form (element threads store into the CRS + finishing pass) against the
# dense element arrays + row gather (MHA_POROUS_DIRECT=0); per-kernel durations from rocprofv3
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in direct gather; do
  rm -rf /tmp/prof_c3
  if [ $v = gather ]; then export MHA_POROUS_DIRECT=0; else unset MHA_POROUS_DIRECT; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3 -o p -- python3 $R/bench.py --config 3 --no-cpu-baseline --steps 20 > /tmp/c3_$v.json 2>/dev/null
  f=$(find /tmp/prof_c3 -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$v" /tmp/c3_$v.json <<'PY'
import csv,sys,json
d=json.loads(open(sys.argv[3]).readlines()[-1])
print('%s: ms_per_step %.4f' % (sys.argv[2], d['ms_per_step']))
for r in csv.DictReader(open(sys.argv[1])):
    if 'mha' in r['Name'] and int(r['Calls'])>5: print('   %-80s calls %4s avg %9.1f us' % (r['Name'][:80], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
