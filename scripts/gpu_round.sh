#!/bin/bash
# Run on the GPU box from the repo root (gpurun): GPU parity tests, bench line, rocprofv3 kernel stats and
# separate PMC passes (HBM traffic) of the SAME bench command.  Everything lands in gpurun_out/<tag>/.
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs"
rm -rf /tmp/prof_stats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- $BENCH > $OUT/bench_prof.log 2>&1
f=$(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && { head -1 $f; grep "mf::" $f; } > $OUT/kernel_stats.csv
cat $OUT/kernel_stats.csv | cut -c1-200
for CTRS in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
  n=$(echo $CTRS | tr ' ' '_' | cut -c1-40)
  rm -rf /tmp/pmc_$n
  rocprofv3 --pmc $CTRS --kernel-trace --kernel-include-regex "mf::" --output-format csv -d /tmp/pmc_$n -- $BENCH > $OUT/pmc_$n.log 2>&1
  f=$(find /tmp/pmc_$n -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY' | tee -a $OUT/pmc_summary.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); seen = collections.defaultdict(set)
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"]
    if "mf::" not in k: continue
    k = k.split("(")[0].replace("void ", "")[:70]
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); seen[k].add(row["Dispatch_Id"])
for k in acc:
    n = len(seen[k]); print(k, "dispatches", n, {c: round(v / n, 1) for c, v in acc[k].items()})
PY
done
