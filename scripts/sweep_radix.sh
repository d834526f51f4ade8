#!/bin/bash
# config 3 (T=100000, d=6, fp32, one chain): sweep of the level-0 chunk length and of the radix of the reduced levels
for r in 8 6 5 4; do for l in 0 8; do
  [ $l -ne 0 ] && [ $l -lt $r ] && continue
  if [ $l -eq 0 ]; then MF_BTD_RADIX=$r python3 scripts/bench_btd.py "$@" | sed "s/^/radix=$r len0=auto  /"
  else MF_BTD_RADIX=$r MF_BTD_PAR_LEN=$l python3 scripts/bench_btd.py "$@" | sed "s/^/radix=$r len0=$l  /"; fi
done; done
