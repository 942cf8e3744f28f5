#!/bin/bash
# round 3: workgroup-merged K1 (thermal_affine_residual_wg_kernel) against the thread-per-element form (MHA_K1=thread)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in wg thread; do
  rm -rf /tmp/prof_k1
  if [ $v = thread ]; then export MHA_K1=thread; else unset MHA_K1; fi
  MHA_K1K2_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k1 -o p -- python3 $R/bench.py --no-cpu-baseline --steps 20 > /tmp/k1_$v.json 2>/tmp/k1_$v.err
  f=$(find /tmp/prof_k1 -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$v" /tmp/k1_$v.json <<'PY'
import csv,sys,json
d=json.loads(open(sys.argv[3]).readlines()[-1])
k1=[(r['Name'][:60],float(r['AverageNs'])/1e3) for r in csv.DictReader(open(sys.argv[1])) if 'thermal_affine_residual' in r['Name']]
k2=[float(r['AverageNs'])/1e3 for r in csv.DictReader(open(sys.argv[1])) if 'block_pattern_jacobian' in r['Name']]
print('%-8s K1 %s  K2 %s us  ms_per_step %.4f frac %.3f' % (sys.argv[2], k1, ' + '.join('%.1f' % x for x in k2), d['ms_per_step'], d['roofline']['frac']))
PY
done
unset MHA_K1
python3 $R/bench.py --no-cpu-baseline --steps 50 --warmup 10 | tail -c 700
