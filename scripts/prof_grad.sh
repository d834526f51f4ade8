#!/bin/bash
# Per-kernel times of log_likelihood().backward() at the headline shape (streamed route), and of the route it replaces.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/scripts/bench_grad.py $PGARGS --iters 5 2>&1 | tail -5
rm -rf /tmp/pg && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pg -- python3 $R/scripts/bench_grad.py $PGARGS --iters 5 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/pg/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:6] + [r for r in rows[6:] if "mf::" in r["Name"]]:
    print(f"{r['Name'][:100]:100s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e6:8.3f} ms  {r['Percentage']}%")
PY
