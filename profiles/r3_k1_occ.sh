#!/bin/bash
# round 3: register budget of the thread-per-element residual kernel K1 (variants of the library built with
# -DMHA_K1_WAVES=N: __launch_bounds__(256, N) -> 2 (default, 256 registers), 3 (168), 4 (128)); K1 alone under rocprofv3
cp mrhyde_amd/lib/libmrhyde_amd.so /tmp/lib_default.so
cd /tmp && export TMPDIR=/tmp
for v in default k1w3 k1w4 default; do
  if [ $v = default ]; then cp /tmp/lib_default.so $GRAFT_REPO_ROOT/mrhyde_amd/lib/libmrhyde_amd.so; else cp $GRAFT_REPO_ROOT/build/lib_$v.so $GRAFT_REPO_ROOT/mrhyde_amd/lib/libmrhyde_amd.so; fi
  rm -rf /tmp/prof_k1
  MHA_K1K2_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k1 -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 20 > /tmp/k1_$v.json 2>/dev/null
  f=$(find /tmp/prof_k1 -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$v" /tmp/k1_$v.json <<'PY'
import csv,sys,json
d=json.loads(open(sys.argv[3]).readlines()[-1])
k1=[float(r['AverageNs'])/1e3 for r in csv.DictReader(open(sys.argv[1])) if 'thermal_affine_residual' in r['Name']]
k2=[float(r['AverageNs'])/1e3 for r in csv.DictReader(open(sys.argv[1])) if 'block_pattern_jacobian' in r['Name']]
print('%-8s K1 %.1f us  K2 %s us  ms_per_step %.4f' % (sys.argv[2], k1[0], ' + '.join('%.1f' % x for x in k2), d['ms_per_step']))
PY
done
cp /tmp/lib_default.so $GRAFT_REPO_ROOT/mrhyde_amd/lib/libmrhyde_amd.so
