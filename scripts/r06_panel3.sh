#!/bin/bash
# parity of the panel kernels (large-d tests, config 5 full shape) and level-0 timings
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kalman_large_d.py -q 2>&1 | tail -4
timeout 600 python -m pytest tests/test_gpu_baseline_configs.py -x -q -k config5 2>&1 | tail -3
python scripts/bench_big.py --iters 10 2>&1 | grep -v amdgpu.ids
python scripts/bench_big.py --iters 10 --d 48 2>&1 | grep -v amdgpu.ids
python scripts/bench_big.py --iters 10 --d 40 2>&1 | grep -v amdgpu.ids
