#!/bin/bash
# config 4 after the q-streamed point engine: bench line with CPU baseline, rocprofv3 kernel statistics, FETCH / WRITE passes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3_traffic
mkdir -p $O $R/gpurun_out/r3
python $R/bench.py --config 4 --steps 20 --warmup 5 > $R/gpurun_out/r3/final_bench_config4.json 2> $R/gpurun_out/r3/final_bench_config4.err
tail -c 700 $R/gpurun_out/r3/final_bench_config4.json
rm -rf $O/stats_config4 $O/pmc_config4_FETCH_SIZE $O/pmc_config4_WRITE_SIZE
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_config4 -- python $R/bench.py --config 4 --steps 5 --warmup 1 --no-cpu-baseline > $O/stats_config4.log 2>&1 || echo "stats FAILED"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_config4_$c -- python $R/bench.py --config 4 --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_config4_$c.log 2>&1 || echo "pmc $c FAILED"
done
find $O -name "*kernel_stats.csv" -o -name "*counter_collection.csv" | grep config4
