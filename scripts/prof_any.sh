#!/bin/bash
# rocprofv3 kernel stats of any python script: bash scripts/prof_any.sh <rows> <script> [args]  -> gpurun_out/prof_any/kstats.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/prof_any; mkdir -p $OUT
ROWS=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pany
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pany -- python3 $R/"$@" > $OUT/run.log 2>&1
python3 $R/scripts/kstats.py /tmp/pany $ROWS | tee $OUT/kstats.txt
