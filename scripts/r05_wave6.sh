#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r05_wave6; mkdir -p $OUT; cd $R
for lib in wpe2 wpe3 wpe2 wpe3; do
  echo "lib $lib" | tee -a $OUT/bench.txt
  MF_LIB_PATH=$R/markovflow_amd/libmf_$lib.so timeout 300 python3 scripts/bench_wave.py --dims 16 --chunks 0,8,12,16,24 2>&1 | grep -v amdgpu | tee -a $OUT/bench.txt
done
