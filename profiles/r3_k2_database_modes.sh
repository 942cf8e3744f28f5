#!/bin/bash
# round 3: variants of the replication kernel of the geometry-database mode (MHA_REP_MODE: 1 nontemporal stores,
# 2 eight chunks per pass, 3 both; MHA_REP_WGS: workgroups)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
IFS=";" read -ra LIST <<< "${MODES:-0 2048;1 2048;2 2048;3 2048;0 1024;0 4096;0 512}"; for v in "${LIST[@]}"; do
  set -- $v
  rm -rf /tmp/prof_db
  MHA_REP_MODE=$1 MHA_REP_WGS=$2 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_db -o p -- python3 $R/bench.py --no-cpu-baseline --steps 20 > /tmp/db.json 2>/dev/null
  f=$(find /tmp/prof_db -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$1 wgs $2" /tmp/db.json <<'PY'
import csv,sys,json
d=json.loads(open(sys.argv[3]).readlines()[-1])
k=[float(r['AverageNs'])/1e3 for r in csv.DictReader(open(sys.argv[1])) if 'replicate_runs' in r['Name']]
print('mode %s: replicate %.1f us  ms_per_step %.4f frac %.3f' % (sys.argv[2], k[0], d['ms_per_step'], d['roofline']['frac']))
PY
done
