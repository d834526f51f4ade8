#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r05_few; mkdir -p $OUT; cd $R
for d in 16 32; do for dt in f64 f32; do timeout 300 python3 scripts/bench_bigops.py --batch 8 --T 2048 --d $d --m 1 --dtype $dt 2>&1 | grep -v amdgpu | grep -E "B=|posterior|cholesky|upper|log_lik|solve " | tee -a $OUT/few.txt; done; done
