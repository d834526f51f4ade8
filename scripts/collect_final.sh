#!/bin/bash
# copy a set of `scripts/gpu_job.sh final <tag>` from gpurun_out/ into profiles/ (replacing the previous one) and re-key the traffic
# bookkeeping.  usage: bash scripts/collect_final.sh <new tag> [<old tag>]
NEW=$1; OLD=$2; cd "$(dirname "$0")/.."
[ -n "$OLD" ] && { git rm -q --cached profiles/${OLD}_* 2>/dev/null; rm -f profiles/${OLD}_*; }
for f in bench.json kernel_stats.csv pmc_summary.txt pytest_gpu.log; do cp gpurun_out/$NEW/$f profiles/${NEW}_$f; done
cp gpurun_out/${NEW}_panel/kernel_stats.csv profiles/${NEW}_panel_kernel_stats.csv; cp gpurun_out/${NEW}_panel/pmc_summary.txt profiles/${NEW}_panel_pmc_summary.txt
grep -v amdgpu gpurun_out/${NEW}_panel.txt > profiles/${NEW}_panel_bench.txt; grep -v amdgpu gpurun_out/${NEW}_wave.txt > profiles/${NEW}_wave_bench.txt
cp gpurun_out/${NEW}_config4_step.txt profiles/${NEW}_config4_step.txt
python3 scripts/update_traffic.py $NEW
tail -3 gpurun_out/$NEW/pytest_gpu.log
