#!/bin/bash
# kernel stats of the GPR routes at a given signature -> gpurun_out/gpr_prof/
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/gpr_prof; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pgpr
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pgpr -- python3 $R/scripts/bench_gpr.py "$@" > $OUT/run.log 2>&1
python3 $R/scripts/kstats.py /tmp/pgpr 25 > $OUT/kstats.txt
tail -1 $OUT/run.log; cat $OUT/kstats.txt
