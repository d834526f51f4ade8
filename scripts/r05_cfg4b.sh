#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r05_cfg4; mkdir -p $OUT; cd $R
timeout 600 python3 scripts/prof_cfg4_ops.py > $OUT/ops.txt 2>&1; tail -5 $OUT/ops.txt
