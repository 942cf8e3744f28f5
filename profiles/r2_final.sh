#!/bin/bash
# Round-2 closing evidence on one MI355X box: bench lines of every configuration, rocprofv3 kernel statistics of the
# driver's command and of the perturbed mesh, phase timing + ablation of the general row-owner kernel.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2_final
mkdir -p $O
LOG=$O/progress.log
: > $LOG
step() { echo "$(date +%T) $*" >> $LOG; }
step bench config2 && timeout -k 10 400 python $R/bench.py > $O/bench_config2.json 2> $O/bench_config2.err || exit 1
step bench perturbed && timeout -k 10 300 python $R/bench.py --mesh perturbed > $O/bench_config2_perturbed.json 2> $O/bench_config2_perturbed.err || exit 1
for c in 3 4 5; do
  step bench config$c && timeout -k 10 400 python $R/bench.py --config $c --no-cpu-baseline > $O/bench_config$c.json 2> $O/bench_config$c.err || exit 1
done
step stats config2 && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_config2 -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/stats_config2.log 2>&1 || exit 1
step stats perturbed && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_config2_perturbed -- python $R/bench.py --mesh perturbed --steps 20 --warmup 3 --no-cpu-baseline > $O/stats_config2_perturbed.log 2>&1 || exit 1
step timing && MHA_GRO_TIMING=$O/gro_timing.bin timeout -k 10 200 python $R/bench.py --mesh perturbed --no-cpu-baseline --steps 5 > /dev/null 2>&1 && python $R/profiles/r2_gro_timing.py $O/gro_timing.bin > $O/gro_phase_cycles.txt
step ablation
: > $O/gro_ablation.log
for d in 0 1 2 4 8 16 31; do
  echo "== MHA_GRO_DBG=$d (profiling build)" >> $O/gro_ablation.log
  MHA_GRO_TIMING=$O/gro_timing_tmp.bin MHA_GRO_DBG=$d timeout -k 10 200 python $R/bench.py --mesh perturbed --no-cpu-baseline --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('ms_per_step %.3f kernel_ms %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms']))" >> $O/gro_ablation.log || exit 1
done
rm -f $O/gro_timing_tmp.bin
step done
cat $LOG
for f in $O/bench_*.json; do tail -1 $f | cut -c1-260; done
cat $O/gro_phase_cycles.txt
