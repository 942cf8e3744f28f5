#!/bin/bash
# round 3: config 3 database mode: lean residual-only kernel over all elements + full kernel over the listed elements
# (default, three waves per SIMD; build/lib_pres2.so: two) against the single kernel (MHA_POROUS_DB_LEAN=0)
R=$GRAFT_REPO_ROOT
cp $R/mrhyde_amd/lib/libmrhyde_amd.so /tmp/lib_default.so
cd /tmp && export TMPDIR=/tmp
for v in lean single; do
  cp /tmp/lib_default.so $R/mrhyde_amd/lib/libmrhyde_amd.so
  unset MHA_POROUS_DB_LEAN

  if [ $v = single ]; then export MHA_POROUS_DB_LEAN=0; fi
  rm -rf /tmp/prof_c3
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
cp /tmp/lib_default.so $R/mrhyde_amd/lib/libmrhyde_amd.so
