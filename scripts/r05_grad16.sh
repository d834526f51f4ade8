#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r05_grad16; mkdir -p $OUT; cd $R
for d in 16 32; do timeout 600 python3 scripts/bench_bigops.py --batch 512 --T 1000 --d $d --m 1 --dtype f64 --grad 2>&1 | grep -v amdgpu | tail -4 | tee -a $OUT/grad.txt; done
