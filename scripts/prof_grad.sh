#!/bin/bash
# Kernel-level breakdown of log_likelihood forward + backward (scripts/bench_grad.py) at two shapes -> gpurun_out/grad_prof/
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/grad_prof; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for SH in "1024 10000" "16384 500"; do
  set -- $SH; tag=B$1_T$2
  rm -rf /tmp/pg_$tag
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pg_$tag -- python3 $R/scripts/bench_grad.py --batch $1 --T $2 --iters 3 > $OUT/$tag.log 2>&1
  python3 $R/scripts/kstats.py /tmp/pg_$tag 30 > $OUT/${tag}_kstats.txt
done
