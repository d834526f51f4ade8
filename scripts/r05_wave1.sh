#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r05_wave1; mkdir -p $OUT; cd $R
timeout 900 python3 -m pytest tests/test_gpu_wave.py -x -q > $OUT/pytest_wave.log 2>&1; tail -15 $OUT/pytest_wave.log
timeout 300 python3 scripts/bench_wave.py --dims 15,16,17,24,32 > $OUT/bench_wave_f64.txt 2>&1; cat $OUT/bench_wave_f64.txt
timeout 300 python3 scripts/bench_wave.py --dims 16,32 --dtype f32 > $OUT/bench_wave_f32.txt 2>&1; cat $OUT/bench_wave_f32.txt
for i in 1 2; do
  MF_TREE=$R/_prev python3 scripts/ab_config2.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab_config2.txt
  python3 scripts/ab_config2.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab_config2.txt
done
timeout 600 python3 -m pytest tests/test_gpu_errors.py tests/test_gpu_kalman_large_d.py tests/test_gpu_posterior_streamed.py -x -q > $OUT/pytest_other.log 2>&1; tail -5 $OUT/pytest_other.log
