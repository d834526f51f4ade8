#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r05_icache; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/scripts/bench_wave.py --dims 16,32"
for CTRS in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_IFETCH" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES"; do
  n=$(echo $CTRS | tr ' ' '_' | cut -c1-30); rm -rf /tmp/pi_$n
  timeout 300 rocprofv3 --pmc $CTRS --kernel-trace --kernel-include-regex "mf::wv" --output-format csv -d /tmp/pi_$n -- $CMD > $OUT/pmc_$n.log 2>&1
  f=$(find /tmp/pi_$n -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY' | tee -a $OUT/icache.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); seen = collections.defaultdict(set)
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); seen[k].add(row["Dispatch_Id"])
for k in acc:
    n = len(seen[k]); print(k, "dispatches", n, {c: round(v / n, 1) for c, v in acc[k].items()})
PY
done

