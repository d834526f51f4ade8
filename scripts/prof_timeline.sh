#!/bin/bash
# kernel timeline (start offset, duration, gap to the previous kernel) of the LAST <count> kernels of a python script:
#   bash scripts/prof_timeline.sh <tag> <count> script.py [args]   -> gpurun_out/<tag>_timeline.txt
TAG=$1; CNT=$2; shift; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pt_$TAG
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/pt_$TAG -- python3 $R/"$@" > /tmp/pt_$TAG.log 2>&1
python3 - <<PY | tee $R/gpurun_out/${TAG}_timeline.txt
import csv, glob
f = glob.glob("/tmp/pt_$TAG/**/*kernel_trace.csv", recursive=True)
rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Start_Timestamp"])) if f else []
rows = rows[-$CNT:]
print("# $*  (last $CNT kernels; us)")
t0 = int(rows[0]["Start_Timestamp"]) if rows else 0
prev = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f"{(s - t0) / 1e3:10.1f}  dur {(e - s) / 1e3:8.1f}  gap {gap:8.1f}  {r['Kernel_Name'][:80]}")
    prev = e
PY
