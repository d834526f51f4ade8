#!/bin/bash
# copy a set of scripts/r05_final.sh from gpurun_out/ into profiles/ (replacing the previous one) and re-key the traffic bookkeeping
# usage: bash scripts/collect_final.sh <new tag> <old tag>
NEW=$1; OLD=$2; cd "$(dirname "$0")/.."
[ -n "$OLD" ] && { git rm -q --cached profiles/${OLD}_* 2>/dev/null; rm -f profiles/${OLD}_*; }
for f in bench.json kernel_stats.csv pmc_summary.txt pytest_gpu.log; do cp gpurun_out/$NEW/$f profiles/${NEW}_$f; done
cp gpurun_out/${NEW}_big/kernel_stats.csv profiles/${NEW}_big_kernel_stats.csv; cp gpurun_out/${NEW}_big/pmc_summary.txt profiles/${NEW}_big_pmc_summary.txt
cp gpurun_out/${NEW}_wave/wave_kernel_stats.csv profiles/${NEW}_wave_kernel_stats.csv; cp gpurun_out/${NEW}_wave/wave_pmc_summary.txt profiles/${NEW}_wave_pmc_summary.txt
cat gpurun_out/${NEW}_wave/bench_wave_f64.txt gpurun_out/${NEW}_wave/bench_wave_f32.txt > profiles/${NEW}_wave_bench.txt
cp gpurun_out/${NEW}_wave/bigops_d16.txt profiles/${NEW}_wave_bigops_d16.txt; cp gpurun_out/${NEW}_wave/bigops_d32.txt profiles/${NEW}_wave_bigops_d32.txt
cp gpurun_out/${NEW}_wave/config4_step.txt profiles/${NEW}_wave_config4_step.txt
python3 scripts/update_traffic.py $NEW
tail -3 gpurun_out/$NEW/pytest_gpu.log
