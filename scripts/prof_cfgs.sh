#!/bin/bash
# usage: bash scripts/prof_cfgs.sh <tag> cfg...   (on the GPU box)
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in "$@"; do
  rm -rf /tmp/prof_$c
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$c -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/prof_cfg.py $c 4 > $out/prof_$c.log 2>&1
  f=$(find /tmp/prof_$c -name '*kernel_trace.csv' | head -1)
  python3 $GRAFT_REPO_ROOT/scripts/trace_summary.py $f > $out/timeline_$c.txt 2>&1
  s=$(find /tmp/prof_$c -name '*kernel_stats.csv' | head -1)
  head -40 $s > $out/stats_$c.csv
done
