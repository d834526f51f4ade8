#!/bin/bash
# rocprofv3 kernel stats of bench.py WITH the other configurations (configs 2-5, the wave kernels, the CVI chain): one line per
# kernel instantiation, i.e. per configuration (config 2: <double, 4, ..>, config 3: <float, 6, ..>, config 4: <double, 9, 3>, config 5:
# big<64>, wave: mf::wv::*) - the table in which a per-configuration regression shows.
# usage: bash scripts/prof_all_configs.sh <tag>  ->  gpurun_out/<tag>_all_configs_kernel_stats.csv
TAG=${1:-r05}; R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pall; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pall -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
f=$(find /tmp/pall -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && { head -1 $f; grep "mf::" $f; } > $R/gpurun_out/${TAG}_all_configs_kernel_stats.csv
wc -l $R/gpurun_out/${TAG}_all_configs_kernel_stats.csv; head -12 $R/gpurun_out/${TAG}_all_configs_kernel_stats.csv | cut -c1-170
