#!/bin/bash
# round 5: host profile of config 2, kernel breakdown of the CVI chain, the d = 16 operators, counters of the wave kernels
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r05_prof; mkdir -p $OUT; cd $R
timeout 300 python3 scripts/prof_host.py 3000 > $OUT/prof_host.txt 2>&1; head -45 $OUT/prof_host.txt
timeout 300 python3 scripts/prof_cvi.py > $OUT/prof_cvi.txt 2>&1; cat $OUT/prof_cvi.txt
timeout 300 python3 scripts/bench_bigops.py --batch 512 --T 1000 --d 16 --m 1 --dtype f64 > $OUT/bigops_d16.txt 2>&1; cat $OUT/bigops_d16.txt
timeout 300 python3 scripts/bench_bigops.py --batch 512 --T 1000 --d 32 --m 1 --dtype f64 > $OUT/bigops_d32.txt 2>&1; cat $OUT/bigops_d32.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pc; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc -- python3 $R/scripts/prof_cvi.py > $OUT/cvi_prof.log 2>&1
f=$(find /tmp/pc -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -30 $f | cut -c1-180 | tee $OUT/cvi_kernel_stats.csv
CMD="python3 $R/scripts/bench_wave.py --dims 16,32"
rm -rf /tmp/pw; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pw -- $CMD > $OUT/wave_stats.log 2>&1
f=$(find /tmp/pw -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && { head -1 $f; grep "mf::" $f; } | cut -c1-200 | tee $OUT/wave_kernel_stats.csv
for CTRS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $CTRS | tr ' ' '_' | cut -c1-30); rm -rf /tmp/pw_$n
  timeout 300 rocprofv3 --pmc $CTRS --kernel-trace --kernel-include-regex "mf::wv" --output-format csv -d /tmp/pw_$n -- $CMD > $OUT/pmc_$n.log 2>&1
  f=$(find /tmp/pw_$n -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY' | tee -a $OUT/wave_pmc_summary.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); seen = collections.defaultdict(set)
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); seen[k].add(row["Dispatch_Id"])
for k in acc:
    n = len(seen[k]); print(k, "dispatches", n, {c: round(v / n, 1) for c, v in acc[k].items()})
PY
done
