#!/bin/bash
# config 4's model (3 x Matern-5/2, 3 outputs, d = 9, B = 512, T = 1000): the training step kernel by kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r05_cfg4; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/scripts/bench_gpr_grad.py --batch 512 --T 1000 --sig 5,5,5 --multi --iters 10 2>&1 | tail -1 | tee $OUT/step.txt
rm -rf /tmp/pgpr && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pgpr -- python3 $R/scripts/bench_gpr_grad.py --batch 512 --T 1000 --sig 5,5,5 --multi --iters 10 > /dev/null 2>&1
python3 - <<'PY' | tee -a $OUT/step.txt
import csv, glob
f = glob.glob('/tmp/pgpr/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:30]:
    print(f"{r['Name'][:150]:150s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e6:8.3f} ms  {r['Percentage']}%")
PY
