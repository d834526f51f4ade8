#!/bin/bash
# A/B on ONE box (boxes of the pool differ by several per cent): an operator with markovflow_amd/libmf_prev.so against the
# current library, alternating.  Usage: bash scripts/ab_lib.sh "<grep pattern of scripts/bench_ops.py rows>" [bench_ops args]
PAT=$1; shift
for i in 1 2; do
  MF_LIB_PATH=$PWD/markovflow_amd/libmf_prev.so python3 scripts/bench_ops.py "$@" 2>&1 | grep -E "$PAT" | sed 's/^/prev /'
  python3 scripts/bench_ops.py "$@" 2>&1 | grep -E "$PAT" | sed 's/^/new  /'
done
