#!/bin/bash
# kernel stats (rocprofv3 --kernel-trace --stats) of bench.py with K1 and K2 serialised; env vars pass through
cd "$GRAFT_REPO_ROOT"
TAG=${1:-r2}
export MHA_K1K2_OVERLAP=0
bash profiles/kstats.sh $TAG bench.py --steps 10 --warmup 2 --no-cpu-baseline
cut -d, -f1-8 gpurun_out/${TAG}_kernel_stats.csv | head -8
