#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r05_marg; mkdir -p $OUT; cd $R
timeout 1500 python3 -m pytest tests/test_gpu_wave.py tests/test_gpu_large_d_ops.py tests/test_gpu_gradients.py -x -q -k "marginal or sites_filter_posterior or kl" > $OUT/pytest.log 2>&1; tail -6 $OUT/pytest.log
for d in 16 32; do
  timeout 300 python3 scripts/bench_bigops.py --batch 512 --T 1000 --d $d --m 1 --dtype f64 2>&1 | grep -v amdgpu | tee $OUT/bigops_d$d.txt
done
