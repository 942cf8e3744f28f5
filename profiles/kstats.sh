#!/bin/bash
# rocprofv3 --kernel-trace --stats of one python command; keeps the kernel_stats csv.  usage: kstats.sh <tag> <script> [args...]
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$TAG -- python $GRAFT_REPO_ROOT/"$@" > $GRAFT_REPO_ROOT/gpurun_out/ks_$TAG.log 2>&1
f=$(ls /tmp/ks_$TAG/*/*kernel_stats.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then cp "$f" $GRAFT_REPO_ROOT/gpurun_out/${TAG}_kernel_stats.csv; head -6 "$f" | cut -c1-160; else echo "no kernel_stats for $TAG"; fi
