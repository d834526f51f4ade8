#!/bin/bash
# round 5, final set on the last library commit: scripts/gpu_round.sh (tests, bench line, K0 stats + PMC), config 5 counters,
# wave-kernel stats + counters, the d = 16 / 32 operators, config 4's training step.   bash scripts/r05_final.sh r05_v2
TAG=${1:-r05_v2}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
bash scripts/gpu_round.sh $TAG > gpurun_out/${TAG}_round.log 2>&1; tail -3 gpurun_out/$TAG/pytest_gpu.log
bash scripts/pmc_big.sh ${TAG}_big > /dev/null 2>&1; cat gpurun_out/${TAG}_big/pmc_summary.txt | cut -c1-400
OUT=$R/gpurun_out/${TAG}_wave; mkdir -p $OUT
cd $R
for d in 16 32; do timeout 300 python3 scripts/bench_bigops.py --batch 512 --T 1000 --d $d --m 1 --dtype f64 2>&1 | grep -v amdgpu > $OUT/bigops_d$d.txt; done
timeout 300 python3 scripts/bench_wave.py --dims 15,16,17,24,30,32 2>&1 | grep -v amdgpu > $OUT/bench_wave_f64.txt
timeout 300 python3 scripts/bench_wave.py --dims 16,24,32 --dtype f32 2>&1 | grep -v amdgpu > $OUT/bench_wave_f32.txt
timeout 300 python3 scripts/bench_gpr_grad.py --batch 512 --T 1000 --sig 5,5,5 --multi --iters 20 2>&1 | tail -1 > $OUT/config4_step.txt
cat $OUT/bench_wave_f64.txt $OUT/bench_wave_f32.txt $OUT/config4_step.txt
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/scripts/bench_wave.py --dims 16,32"
rm -rf /tmp/pw; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pw -- $CMD > $OUT/wave_stats.log 2>&1
f=$(find /tmp/pw -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && { head -1 $f; grep "mf::" $f; } | cut -c1-200 | tee $OUT/wave_kernel_stats.csv
for CTRS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"; do
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
