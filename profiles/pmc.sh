#!/bin/bash
# PMC passes for bench.py (one counter group per run, kernel-trace only): usage: pmc.sh <outdir-tag>
set -e
TAG=${1:-r1}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
cd /tmp && export TMPDIR=/tmp
run() { # name, counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python $GRAFT_REPO_ROOT/${PMC_PROG:-bench.py} ${PMC_ARGS:---steps 3 --warmup 1 --no-cpu-baseline} > $OUT.$name.log 2>&1 || echo "pass $name failed"
}
mkdir -p $OUT
run inst SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES
run wait SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE
python - <<PY
import csv, glob, collections
for name in ("inst","wait","fetch","write","grbm"):
    for f in glob.glob("$OUT/%s/*/*counter_collection.csv" % name):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-60:]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in agg.items():
            if "rocclr" in k or "at::" in k: continue
            print(name, k, {c: "%.4g" % (sum(v)/len(v)) for c, v in d.items()}, "n=%d" % len(next(iter(d.values()))))
PY
