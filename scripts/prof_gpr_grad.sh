#!/bin/bash
# kernel stats of the GPR training step (scripts/bench_gpr_grad.py) -> gpurun_out/gpr_grad_prof/
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/gpr_grad_prof; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pgg
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pgg -- python3 $R/scripts/bench_gpr_grad.py "$@" > $OUT/run.log 2>&1
python3 $R/scripts/kstats.py /tmp/pgg 22 > $OUT/kstats.txt
tail -1 $OUT/run.log; cat $OUT/kstats.txt
