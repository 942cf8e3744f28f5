#!/bin/bash
# general row-owner kernel: parity tests, bench line of the perturbed mesh, per-phase cycles (profiling build), HBM traffic
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_thermal_gpu.py -x -q -k "general_row_owner or auto_path" 2>&1 | tail -2 || exit 1
timeout -k 10 200 python bench.py --mesh perturbed --no-cpu-baseline --steps 10 2>/dev/null | tail -1 | cut -c1-330 || exit 1
MHA_GRO_TIMING=gpurun_out/gro_timing.bin timeout -k 10 200 python bench.py --mesh perturbed --no-cpu-baseline --steps 5 > /dev/null 2>&1 && python profiles/r2_gro_timing.py gpurun_out/gro_timing.bin
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2_traffic; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_config2_perturbed_$c
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_config2_perturbed_$c -- python $GRAFT_REPO_ROOT/bench.py --mesh perturbed --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_config2_perturbed_$c.log 2>&1 || exit 1
done
echo done
