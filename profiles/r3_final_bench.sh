#!/bin/bash
# Round-3 closing bench lines (with cpu_baseline) for every configuration; run from the repo root on the GPU box.
set -e
O=gpurun_out/r3
mkdir -p $O
for c in 2 3 4 5; do
  python bench.py --config $c --steps 20 --warmup 5 > $O/final_bench_config$c.json 2> $O/final_bench_config$c.err
  python -c "import json,sys; d=json.loads(open(sys.argv[1]).readlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'] if d.get('cpu_baseline') else None)" $O/final_bench_config$c.json
done
python bench.py --config 2 --mesh perturbed --steps 20 --warmup 5 --no-cpu-baseline > $O/final_bench_config2_perturbed.json 2> $O/final_bench_config2_perturbed.err
python bench.py --config 2 --jacobian full --steps 20 --warmup 5 --no-cpu-baseline > $O/final_bench_config2_full.json 2> $O/final_bench_config2_full.err
tail -c 400 $O/final_bench_config2_perturbed.json
