#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r05_cvi2; mkdir -p $OUT; cd $R
timeout 900 python3 -m pytest tests/test_gpu_transformations.py tests/test_gpu_autograd_ops.py tests/test_gpu_gradients.py -x -q > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
timeout 300 python3 scripts/prof_cvi.py > $OUT/prof_cvi.txt 2>&1; cat $OUT/prof_cvi.txt | grep -v Warn
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pc; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc -- python3 $R/scripts/prof_cvi.py > $OUT/cvi_prof.log 2>&1
python3 $R/scripts/kstats.py /tmp/pc 16 | tee $OUT/cvi_kstats.txt
