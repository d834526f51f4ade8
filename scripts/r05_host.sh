#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r05_host; mkdir -p $OUT; cd $R
timeout 900 python3 -m pytest tests/test_gpu_errors.py tests/test_gpu_kalman.py tests/test_gpu_transformations.py tests/test_gpu_autograd_ops.py tests/test_gpu_block_tri_diag.py tests/test_gpu_distributed.py -x -q > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
timeout 300 python3 scripts/prof_host.py 3000 > $OUT/prof_host.txt 2>&1; head -22 $OUT/prof_host.txt
for i in 1 2; do
  MF_TREE=$R/_prev python3 scripts/ab_config2.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab_config2.txt
  python3 scripts/ab_config2.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab_config2.txt
done
timeout 300 python3 scripts/prof_cvi.py > $OUT/prof_cvi.txt 2>&1; cat $OUT/prof_cvi.txt | grep -v Warn
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pc; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc -- python3 $R/scripts/prof_cvi.py > $OUT/cvi_prof.log 2>&1
python3 $R/scripts/kstats.py /tmp/pc 30 | tee $OUT/cvi_kstats.txt
