#!/bin/bash
# rocprofv3 kernel stats + SQ counter passes of the large-d (config 5) log-likelihood; run on the GPU box from the repo root.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-big}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/scripts/bench_big.py --iters 3"
rm -rf /tmp/pb_stats; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb_stats -- $CMD > $OUT/stats.log 2>&1
f=$(find /tmp/pb_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && { head -1 $f; grep "mf::" $f; } > $OUT/kernel_stats.csv
cut -c1-160 $OUT/kernel_stats.csv
for CTRS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"; do
  n=$(echo $CTRS | tr ' ' '_' | cut -c1-30); rm -rf /tmp/pb_$n
  timeout 300 rocprofv3 --pmc $CTRS --kernel-trace --kernel-include-regex "mf::(big|pn)" --output-format csv -d /tmp/pb_$n -- $CMD > $OUT/pmc_$n.log 2>&1
  f=$(find /tmp/pb_$n -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY' | tee -a $OUT/pmc_summary.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); seen = collections.defaultdict(set)
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); seen[k].add(row["Dispatch_Id"])
for k in acc:
    n = len(seen[k]); print(k, "dispatches", n, {c: round(v / n, 1) for c, v in acc[k].items()})
PY
done
