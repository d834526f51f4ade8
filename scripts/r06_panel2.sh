#!/bin/bash
# level-0 timing of config 5 on the panel kernels + the per-phase stamps of a step (diagnostic build) at two and one workgroup per CU
mkdir -p gpurun_out
O=gpurun_out/r06_panel2.txt
python scripts/bench_big.py --iters 10 > $O 2>&1
python scripts/bench_big.py --iters 10 --chunks 32 >> $O 2>&1
python scripts/bench_big.py --iters 10 --chunks 128 >> $O 2>&1
if [ -f markovflow_amd/libmf_pstamp.so ]; then
MF_LIB_PATH=$PWD/markovflow_amd/libmf_pstamp.so python scripts/bench_big.py --iters 1 >> $O 2>&1
MF_LIB_PATH=$PWD/markovflow_amd/libmf_pstamp.so python scripts/bench_big.py --iters 1 --chunks 32 >> $O 2>&1
fi
grep -v amdgpu.ids $O
