#!/bin/bash
# PMC passes for the point-engine kernel alone (one counter group per run).  usage: pmc_engine.sh <module> <ncell>
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_engine
cd /tmp && export TMPDIR=/tmp
mkdir -p $OUT
run() { local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python $GRAFT_REPO_ROOT/tests/engine_bench.py $MOD $NC gather > $OUT.$name.log 2>&1 || echo "pass $name failed"; }
MOD=${1:-thermal}; NC=${2:-32}
run inst SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES
run wait SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run mfma SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD
python - <<PY
import csv, glob, collections
for name in ("inst","wait","mfma"):
    for f in glob.glob("$OUT/%s/*/*counter_collection.csv" % name):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "point_engine" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(name, {c: "%.4g" % (sum(v)/len(v)) for c, v in agg.items()}, "n=%d" % (len(next(iter(agg.values()))) if agg else 0))
PY
