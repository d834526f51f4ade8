#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r05_ops; mkdir -p $OUT; cd $R
timeout 900 python3 -m pytest tests/test_gpu_autograd_ops.py tests/test_gpu_conditionals.py tests/test_gpu_wave.py -x -q > $OUT/pytest_new.log 2>&1; tail -15 $OUT/pytest_new.log
timeout 1500 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -5 $OUT/pytest_gpu.log
timeout 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 6000 $OUT/bench.json; tail -5 $OUT/bench.err
