#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r05_fuzz; mkdir -p $OUT; cd $R
for seed in 11 12; do timeout 1200 python3 scripts/fuzz_wave.py 120 $seed 2>&1 | grep -v amdgpu | tail -4 | tee -a $OUT/fuzz_wave.txt; done
timeout 900 python3 scripts/fuzz_large_d.py 150 7 2>&1 | grep -v amdgpu | tail -2 | tee -a $OUT/fuzz_wave.txt
