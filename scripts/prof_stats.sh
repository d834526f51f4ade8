#!/bin/bash
# rocprofv3 kernel stats of any python script (mf:: kernels first): bash scripts/prof_stats.sh <tag> script.py [args]  -> gpurun_out/<tag>_stats.txt
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ps_$TAG
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps_$TAG -- python3 $R/"$@" > /tmp/ps_$TAG.log 2>&1
python3 - <<PY | tee $R/gpurun_out/${TAG}_stats.txt
import csv, glob
f = glob.glob("/tmp/ps_$TAG/**/*kernel_stats.csv", recursive=True)
print("# $*")
for r in list(csv.DictReader(open(f[0])))[:40] if f else []:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>6s}  avg {float(r['AverageNs']) / 1e3:10.1f} us  {r['Percentage']}%")
PY
