#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/chol; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/scripts/prof_chol.py"
rm -rf /tmp/pc_stats; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc_stats -- $CMD > $OUT/stats.log 2>&1
f=$(find /tmp/pc_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && { head -1 $f; grep "par_chol_emit\|par_solve_emit" $f; } > $OUT/kernel_stats.csv
for CTRS in "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pc_$CTRS
  timeout 300 rocprofv3 --pmc $CTRS --kernel-trace --kernel-include-regex "par_chol_emit|par_solve_emit" --output-format csv -d /tmp/pc_$CTRS -- $CMD > $OUT/pmc_$CTRS.log 2>&1
  f=$(find /tmp/pc_$CTRS -name "*counter_collection.csv" | head -1)
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
cat $OUT/kernel_stats.csv | cut -c1-200
