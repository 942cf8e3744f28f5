#!/bin/bash
# memory-path counters of the Jacobian kernel, pattern form against row blocks (two counters per --pmc pass, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
export MHA_K1K2_OVERLAP=0
LOG=$GRAFT_REPO_ROOT/gpurun_out/r2_pmc_mem_progress.log
for k2 in pattern blocks; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcmem_$k2
  mkdir -p $OUT
  run() { local name=$1; shift; echo "$k2 $name" >> $LOG; MHA_K2=$k2 timeout -k 5 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT.$name.log 2>&1 || echo "pass $name failed" >> $LOG; }
  run ta1 TA_BUFFER_WRITE_WAVEFRONTS_sum TA_TA_BUSY_sum
  run ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
  run tcp1 TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum
  run tcp2 TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
  run tcc1 TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum
  run tcc2 TCC_TAG_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum
  run tcc3 TCC_WRITE_sum TCC_EA0_WRREQ_64B_sum
  run tlb TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum
done
python - <<PY
import csv, glob, collections, os
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out"
for k2 in ("pattern", "blocks"):
    for f in sorted(glob.glob(root + "/pmcmem_%s/*/*/*counter_collection.csv" % k2)):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "block_pattern_jacobian" in k: k = "K2pattern"
            elif "row_owner_jacobian" in k: k = "K2blocks"
            else: continue
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in agg.items():
            print(k2, k, {c: "%.4g" % (sum(v)/len(v)) for c, v in d.items()})
PY
