#!/bin/bash
# GPR training step (forward + backward w.r.t. the hyper-parameters) at the headline shape, kernel by kernel.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/scripts/bench_gpr_grad.py --batch 1024 --T 10000 --sig 5,5 --iters 5 2>&1 | tail -1
rm -rf /tmp/pgpr && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pgpr -- python3 $R/scripts/bench_gpr_grad.py --batch 1024 --T 10000 --sig 5,5 --iters 5 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/pgpr/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(f"{r['Name'][:120]:120s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e6:8.3f} ms  {r['Percentage']}%")
PY
