#!/bin/bash
# the large-d log-likelihood tests (incl. m > 4 at 16 < d <= 32 on the two-wavefront panel kernels) and the whole GPU suite's quick parts
timeout 1200 python -m pytest tests/test_gpu_kalman_large_d.py tests/test_gpu_wave.py tests/test_gpu_large_d_ops.py -q 2>&1 | tail -5
python scripts/bench_wave.py --dims 24,32 --dtype f64 --m 8 2>&1 | grep -v amdgpu.ids
