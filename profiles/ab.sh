#!/bin/bash
# A/B timing of library builds ON ONE BOX (boxes of the pool differ by up to 1.5x: never compare
# numbers from different gpurun calls).  usage: ab.sh "<env assignments or ->" ...   e.g.
#   ab.sh "-" "MHA_K2_MODE=1" "MHA_K2_MODE=2"
# Each configuration is run ROUNDS times, interleaved, under rocprofv3 --kernel-trace --stats.
cd /tmp && export TMPDIR=/tmp
ROUNDS=${ROUNDS:-2}
for r in $(seq $ROUNDS); do
  i=0
  for cfg in "$@"; do
    i=$((i+1))
    rm -rf /tmp/ab_$i
    if [ "$cfg" = "-" ]; then envs=""; else envs="$cfg"; fi
    env $envs rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_$i -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline ${BENCH_ARGS} > /tmp/ab_$i.log 2>&1
    k2=$(grep -E "row_owner_jacobian|row_wave_affine" /tmp/ab_$i/*/*kernel_stats.csv | awk -F, '{printf "%.1f", $(NF-4)/1000}')
    k1=$(grep thermal_affine_element /tmp/ab_$i/*/*kernel_stats.csv | awk -F, '{printf "%.1f", $(NF-4)/1000}')
    ms=$(tail -1 /tmp/ab_$i.log | python -c "import sys,json; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null)
    echo "round $r [$cfg] K2=${k2}us K1=${k1}us step=${ms}ms"
  done
done
