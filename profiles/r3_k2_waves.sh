#!/bin/bash
# round 3: does the block-pattern kernel K2 (store-bound) need its 12 waves per CU?  Variants of the library built with
# -DMHA_BP_WAVES=8 / 10 (build/lib_bpw8.so, lib_bpw10.so); K1 and K2 serialised, per-kernel times from rocprofv3.
cp mrhyde_amd/lib/libmrhyde_amd.so /tmp/lib_default.so
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in default bpw10 bpw8 default; do
  if [ $v = default ]; then cp /tmp/lib_default.so $R/mrhyde_amd/lib/libmrhyde_amd.so; else cp $R/build/lib_$v.so $R/mrhyde_amd/lib/libmrhyde_amd.so; fi
  rm -rf /tmp/prof_k2w
  MHA_K1K2_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k2w -o p -- python3 $R/bench.py --no-cpu-baseline --steps 20 > /tmp/k2w_$v.json 2>/tmp/k2w_$v.err
  f=$(find /tmp/prof_k2w -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$v" /tmp/k2w_$v.json <<'PY'
import csv,sys,json
d=json.loads(open(sys.argv[3]).readlines()[-1])
k1=[float(r['AverageNs'])/1e3 for r in csv.DictReader(open(sys.argv[1])) if 'thermal_affine_residual' in r['Name']]
k2=[float(r['AverageNs'])/1e3 for r in csv.DictReader(open(sys.argv[1])) if 'block_pattern_jacobian' in r['Name']]
print('%-8s K1 %.1f us  K2 %s us  ms_per_step %.4f' % (sys.argv[2], k1[0], ' + '.join('%.1f' % x for x in k2), d['ms_per_step']))
PY
done
cp /tmp/lib_default.so $R/mrhyde_amd/lib/libmrhyde_amd.so
