#!/bin/bash
# rocprofv3 kernel stats of scripts/bench_bigops.py -> gpurun_out/bigops_prof/
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/bigops_prof; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pbo
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pbo -- python3 $R/scripts/bench_bigops.py "$@" > $OUT/run.log 2>&1
python3 $R/scripts/kstats.py /tmp/pbo 40 > $OUT/kstats.txt
grep -v amdgpu.ids $OUT/run.log; cat $OUT/kstats.txt
