#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r05_wgrad; mkdir -p $OUT; cd $R
timeout 1500 python3 -m pytest tests/test_gpu_gradients.py -x -q -k "large_d_local or tensor_gradients_vs_dense or kl_" > $OUT/pytest.log 2>&1; tail -6 $OUT/pytest.log
for d in 16 32; do timeout 600 python3 scripts/bench_bigops.py --batch 512 --T 1000 --d $d --m 1 --dtype f64 --grad 2>&1 | grep -v amdgpu | tail -1 | tee -a $OUT/grad.txt; done
