#!/bin/bash
# first GPU pass of the panel kernels: parity (large-d tests + config 5 at full shape), then the level-0 timing of config 5
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kalman_large_d.py -x -q 2>&1 | tail -15 > gpurun_out/r06_panel1_tests.txt
timeout 600 python -m pytest tests/test_gpu_baseline_configs.py -x -q -k config5 2>&1 | tail -8 >> gpurun_out/r06_panel1_tests.txt
timeout 300 python scripts/bench_big.py --iters 10 > gpurun_out/r06_panel1_bench.txt 2>&1
timeout 300 python scripts/bench_big.py --iters 10 --batch 64 >> gpurun_out/r06_panel1_bench.txt 2>&1
cat gpurun_out/r06_panel1_tests.txt gpurun_out/r06_panel1_bench.txt
