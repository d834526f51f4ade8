#!/bin/bash
# A/B: 16 < d <= 32 on the wave kernels (shipped) against the panel kernels with two wavefronts per workgroup (variant build)
for lib in libmarkovflow_amd.so libmf_nt2.so; do
  echo "== $lib"
  for dt in f64 f32; do
    MF_LIB_PATH=$PWD/markovflow_amd/$lib python scripts/bench_wave.py --dims 17,24,32 --dtype $dt 2>&1 | grep -v amdgpu.ids
  done
  MF_LIB_PATH=$PWD/markovflow_amd/$lib python scripts/bench_wave.py --dims 24,32 --dtype f64 --m 8 2>&1 | grep -v amdgpu.ids
  MF_LIB_PATH=$PWD/markovflow_amd/$lib python scripts/bench_wave.py --dims 24,32 --dtype f32 --m 8 2>&1 | grep -v amdgpu.ids
done
