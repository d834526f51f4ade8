#!/bin/bash
# round 5, first GPU call: new cache / double-backward tests, config-2 A/B against the end of round 3 (tree _prev), MFMA rates
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r05_first; mkdir -p $OUT; cd $R
python3 -m pytest tests/test_gpu_posterior_streamed.py tests/test_gpu_gpr_grad.py -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
./scripts/micro/mfma_rate > $OUT/mfma_rate.txt 2>&1; cat $OUT/mfma_rate.txt
for i in 1 2; do
  MF_TREE=$R/_prev python3 scripts/ab_config2.py 2>&1 | tee -a $OUT/ab_config2.txt
  python3 scripts/ab_config2.py 2>&1 | tee -a $OUT/ab_config2.txt
done
cd /tmp && export TMPDIR=/tmp
for tree in _prev .; do
  tag=$(basename $(realpath $R/$tree)); rm -rf /tmp/ps_$tag
  MF_TREE=$R/$tree rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps_$tag -- python3 $R/scripts/ab_config2.py > $OUT/prof_$tag.log 2>&1
  f=$(find /tmp/ps_$tag -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && { echo "== $tag"; head -12 $f | cut -c1-220; } | tee -a $OUT/kernel_stats_ab.txt
done
