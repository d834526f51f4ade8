#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r05_wave5; mkdir -p $OUT; cd $R
timeout 900 python3 -m pytest tests/test_gpu_wave.py tests/test_gpu_kalman_large_d.py -x -q > $OUT/pytest.log 2>&1; tail -6 $OUT/pytest.log
for mode in 1 0; do
  echo "MF_WAVE_MULTI=$mode" | tee -a $OUT/bench.txt
  MF_WAVE_MULTI=$mode timeout 300 python3 scripts/bench_wave.py --dims 16 --chunks 0,4,8,16 2>&1 | grep -v amdgpu | tee -a $OUT/bench.txt
  MF_WAVE_MULTI=$mode timeout 300 python3 scripts/bench_wave.py --dims 16 --dtype f32 --chunks 0,8,16,32 2>&1 | grep -v amdgpu | tee -a $OUT/bench.txt
done
