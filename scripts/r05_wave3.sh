#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r05_wave3; mkdir -p $OUT; cd $R
timeout 900 python3 -m pytest tests/test_gpu_wave.py tests/test_gpu_conditionals.py tests/test_gpu_kalman_large_d.py -x -q > $OUT/pytest_wave.log 2>&1; tail -5 $OUT/pytest_wave.log
timeout 300 python3 scripts/bench_wave.py --dims 16,17,24,32 --chunks 0,2,4,8 > $OUT/bench_wave_f64.txt 2>&1; cat $OUT/bench_wave_f64.txt
timeout 300 python3 scripts/bench_wave.py --dims 16,32 --dtype f32 --chunks 0,4,8,16 > $OUT/bench_wave_f32.txt 2>&1; cat $OUT/bench_wave_f32.txt
timeout 1500 python3 -m pytest tests -m gpu -q --deselect tests/test_gpu_wave.py --deselect tests/test_gpu_conditionals.py --deselect tests/test_gpu_kalman_large_d.py > $OUT/pytest_gpu.log 2>&1; tail -5 $OUT/pytest_gpu.log
