#!/bin/bash
# randomised / determinism campaigns on the final library of round 5
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r05_fuzz_final; mkdir -p $OUT; cd $R
{ echo "== fuzz_parity.py 300 17"; timeout 1200 python3 scripts/fuzz_parity.py 300 17 2>&1 | grep -v amdgpu | tail -3
  echo "== fuzz_moments.py"; timeout 900 python3 scripts/fuzz_moments.py 2>&1 | grep -v amdgpu | tail -2
  echo "== fuzz_large_d.py 200 19"; timeout 900 python3 scripts/fuzz_large_d.py 200 19 2>&1 | grep -v amdgpu | tail -2
  echo "== fuzz_wave.py 200 61"; timeout 900 python3 scripts/fuzz_wave.py 200 61 2>&1 | grep -v amdgpu | tail -2
  echo "== stress_determinism.py"; timeout 900 python3 scripts/stress_determinism.py 2>&1 | grep -v amdgpu | tail -4; } | tee $OUT/fuzz.txt
