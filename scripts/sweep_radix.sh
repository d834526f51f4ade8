#!/bin/bash
# config 3 (T=100000, d=6, fp32, one chain): sweep of the level-0 chunk length and of the radix of the reduced levels.
# Needs the experiment build: MF_LIB_PATH=markovflow_amd/libmarkovflow_amd_exp.so scripts/sweep_radix.sh [bench_btd args]
for r in ${RADII:-16 12 10 8 6 5 4}; do for l in ${LENS:-0 8 12 16 24}; do
  if [ $l -eq 0 ]; then MF_BTD_RADIX=$r python3 scripts/bench_btd.py "$@" | sed "s/^/radix=$r len0=auto  /"
  else MF_BTD_RADIX=$r MF_BTD_PAR_LEN=$l python3 scripts/bench_btd.py "$@" | sed "s/^/radix=$r len0=$l  /"; fi
done; done
