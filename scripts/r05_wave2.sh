#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r05_wave2; mkdir -p $OUT; cd $R
timeout 900 python3 -m pytest tests/test_gpu_wave.py -x -q > $OUT/pytest_wave.log 2>&1; tail -5 $OUT/pytest_wave.log
timeout 300 python3 scripts/bench_wave.py --dims 16,17,24,32 --chunks 0,1,2,4,8 > $OUT/bench_wave_f64.txt 2>&1; cat $OUT/bench_wave_f64.txt
timeout 300 python3 scripts/bench_wave.py --dims 16,32 --dtype f32 --chunks 0,2,4,8 > $OUT/bench_wave_f32.txt 2>&1; cat $OUT/bench_wave_f32.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ps_wave
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps_wave -- python3 $R/scripts/bench_wave.py --dims 16,32 > $OUT/prof.log 2>&1
f=$(find /tmp/ps_wave -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && { head -8 $f | cut -c1-200; } | tee $OUT/kernel_stats.txt
