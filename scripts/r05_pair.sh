#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r05_pair3; mkdir -p $OUT; cd $R
timeout 900 python3 -m pytest tests/test_gpu_wave.py -x -q -k "log_likelihood or partition or sites or d30 or oracle_at" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for i in 1 2; do
  for lib in prev new; do
    if [ $lib = prev ]; then export MF_LIB_PATH=$R/markovflow_amd/libmf_prev.so; else unset MF_LIB_PATH; fi
    echo "== $lib" | tee -a $OUT/bench.txt
    timeout 300 python3 scripts/bench_wave.py --dims 16 --chunks 0,4 2>&1 | grep -v amdgpu | tee -a $OUT/bench.txt
    timeout 300 python3 scripts/bench_wave.py --dims 16 --dtype f32 --chunks 0 2>&1 | grep -v amdgpu | tee -a $OUT/bench.txt
  done
done
