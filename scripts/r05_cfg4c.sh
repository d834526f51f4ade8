#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r05_cfg4; mkdir -p $OUT; cd $R
timeout 900 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_gpr_grad.py tests/test_gpu_gradients.py -x -q > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
python3 scripts/bench_gpr_grad.py --batch 512 --T 1000 --sig 5,5,5 --multi --iters 20 2>&1 | tail -1 | tee $OUT/step_after.txt
python3 scripts/bench_gpr_grad.py --batch 1024 --T 10000 --sig 5,5 --iters 5 2>&1 | tail -1 | tee -a $OUT/step_after.txt
