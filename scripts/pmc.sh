#!/bin/bash
# usage: scripts/pmc.sh <tag> <python args...>   (run on the GPU box from the repo root)
# Runs separate rocprofv3 --pmc passes (counters only, plus --kernel-trace) and prints per-kernel sums for mf:: kernels.
TAG=$1; shift
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
mkdir -p $R/gpurun_out/pmc_$TAG
cd /tmp && export TMPDIR=/tmp
i=0
for CTRS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" \
            "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM" \
            "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$TAG_$i
  rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d /tmp/pmc_${TAG}_$i -- python3 $R/scripts/prof_kf.py "$@" > /tmp/pmc_${TAG}_$i.log 2>&1
  f=$(find /tmp/pmc_${TAG}_$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
f = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
seen = set()
for row in csv.DictReader(open(f)):
    k = row["Kernel_Name"]
    if "mf::" not in k: continue
    k = k.split("(")[0][:60]
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    key = (row["Dispatch_Id"], k)
    if key not in seen: seen.add(key); cnt[k] += 1
for k in acc:
    print(k, "dispatches", cnt[k], {c: round(v / cnt[k]) for c, v in acc[k].items()})
PY
done
