#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r05_wave7; mkdir -p $OUT; cd $R
timeout 900 python3 -m pytest tests/test_gpu_wave.py tests/test_gpu_large_d_ops.py tests/test_gpu_kalman_large_d.py -x -q > $OUT/pytest.log 2>&1; tail -6 $OUT/pytest.log
timeout 300 python3 scripts/bench_bigops.py --batch 512 --T 1000 --d 16 --m 1 --dtype f64 > $OUT/bigops_d16.txt 2>&1; cat $OUT/bigops_d16.txt
timeout 300 python3 scripts/bench_bigops.py --batch 512 --T 1000 --d 32 --m 1 --dtype f64 > $OUT/bigops_d32.txt 2>&1; cat $OUT/bigops_d32.txt
timeout 300 python3 scripts/bench_wave.py --dims 16,32 2>&1 | grep -v amdgpu | tee $OUT/bench_wave.txt
