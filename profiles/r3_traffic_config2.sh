#!/bin/bash
# round 3: the config-2 part of r3_traffic.sh alone (geometry-database mode, without the full-kernel comparison run), and
# the full kernel's statistics beside it (stats_config2_full)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3_traffic
mkdir -p $O
rm -rf $O/stats_config2 $O/pmc_config2_FETCH_SIZE $O/pmc_config2_WRITE_SIZE $O/stats_config2_full
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_config2 -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-full-compare > $O/stats_config2.log 2>&1 &&
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_config2_full -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --jacobian full > $O/stats_config2_full.log 2>&1 &&
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_config2_$c -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-full-compare > $O/pmc_config2_$c.log 2>&1 || break
done
find $O -name "*kernel_stats.csv" | head
