#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r05_udl2; mkdir -p $OUT; cd $R
timeout 900 python3 -m pytest tests/test_gpu_wave.py tests/test_gpu_kalman_large_d.py tests/test_gpu_large_d_ops.py -x -q > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
timeout 600 python3 scripts/fuzz_wave.py 150 71 2>&1 | grep -v amdgpu | tail -2 | tee $OUT/fuzz.txt
for d in 16 32; do timeout 600 python3 scripts/bench_bigops.py --batch 512 --T 1000 --d $d --m 1 --dtype f64 2>&1 | grep -v amdgpu | grep -E "B=|upper|posterior|cholesky|moments|inverse" | tee -a $OUT/bigops.txt; done
