#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r05_post_grad; mkdir -p $OUT; cd $R
timeout 1500 python3 -m pytest tests/test_gpu_gradients.py tests/test_gpu_kernels.py tests/test_gpu_gpr_grad.py tests/test_gpu_autograd_ops.py -x -q > $OUT/pytest.log 2>&1; tail -30 $OUT/pytest.log
