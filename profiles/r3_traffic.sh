#!/bin/bash
# Round-3 evidence run on one MI355X:
#  (a) rocprofv3 --kernel-trace --stats of the driver's bench command (config 2) and of the perturbed mesh
#  (b) HBM traffic of one assembly from the L2's memory-side counters: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes
#      (kernel-trace only), for config 2 (affine and perturbed), config 3 (porousMixed 128^3), config 4 (navierstokes 64^3)
#      and config 5 (HDG element step 256^2)
# Outputs under gpurun_out/r3_traffic/; profiles/r3_traffic_post.py turns them into profiles/r3_traffic.json.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3_traffic
mkdir -p $O
LOG=$O/progress.log
: > $LOG
stats() {  # name, bench args...
  local name=$1; shift
  echo "stats $name" >> $LOG
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$name -- python $R/bench.py "$@" > $O/stats_$name.log 2>&1 || { echo "stats $name FAILED" >> $LOG; return 1; }
}
pmc() {  # name, counter, bench args...
  local name=$1 ctr=$2; shift 2
  echo "pmc $name $ctr" >> $LOG
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/pmc_${name}_$ctr -- python $R/bench.py "$@" > $O/pmc_${name}_$ctr.log 2>&1 || { echo "pmc $name $ctr FAILED" >> $LOG; return 1; }
}
stats config2 --steps 20 --warmup 3 --no-cpu-baseline --no-full-compare &&
stats config2_perturbed --mesh perturbed --steps 20 --warmup 3 --no-cpu-baseline &&
stats config3 --config 3 --steps 20 --warmup 3 --no-cpu-baseline &&
stats config4 --config 4 --steps 5 --warmup 1 --no-cpu-baseline &&
stats config5 --config 5 --steps 20 --warmup 3 --no-cpu-baseline &&
for c in FETCH_SIZE WRITE_SIZE; do
  pmc config2 $c --steps 3 --warmup 1 --no-cpu-baseline --no-full-compare &&
  pmc config2_perturbed $c --mesh perturbed --steps 3 --warmup 1 --no-cpu-baseline &&
  pmc config3 $c --config 3 --steps 3 --warmup 1 --no-cpu-baseline &&
  pmc config4 $c --config 4 --steps 3 --warmup 1 --no-cpu-baseline &&
  pmc config5 $c --config 5 --steps 3 --warmup 1 --no-cpu-baseline || break
done
cat $LOG
find $O -name "*kernel_stats.csv" -o -name "*counter_collection.csv" | head -20
