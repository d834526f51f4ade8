#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r05_wave4; mkdir -p $OUT; cd $R
timeout 900 python3 -m pytest tests/test_gpu_wave.py tests/test_gpu_kalman_large_d.py tests/test_gpu_transformations.py tests/test_gpu_autograd_ops.py tests/test_gpu_gradients.py -x -q > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
timeout 300 python3 scripts/bench_wave.py --dims 16,17,24,32 > $OUT/bench_wave_f64.txt 2>&1; cat $OUT/bench_wave_f64.txt
timeout 300 python3 scripts/bench_wave.py --dims 16,32 --dtype f32 > $OUT/bench_wave_f32.txt 2>&1; cat $OUT/bench_wave_f32.txt
timeout 300 python3 scripts/prof_cvi.py > $OUT/prof_cvi.txt 2>&1; cat $OUT/prof_cvi.txt | grep -v Warn
