#!/bin/bash
# The GPU jobs of a round, by name (one file instead of a script per gpurun call; rounds 1-5 kept theirs as scripts/rNN_*.sh, which
# git history still holds).  Run from the repo root on the GPU box:
#     gpurun --timeout 3600 -- 'bash scripts/gpu_job.sh <job> [tag]'
#   full <tag>     the whole GPU suite, smoke, the bench line                       -> gpurun_out/<tag>_{pytest,smoke,bench}.*
#   round <tag>    scripts/gpu_round.sh: suite + bench + K0 kernel stats + PMC passes (the figures hbm_traffic.json quotes)
#   panel <tag>    large-d parity, config 5 at full shape, level-0 timings at d = 64 / 48 / 40, kernel stats + SQ counters
#   wave <tag>     16 <= d <= 32: parity + bench_wave (both dtypes, m = 1 and m = 8)
#   fuzz <tag>     the randomised / determinism campaigns
#   final <tag>    round + panel + wave + config 4's training step: the set a round's last commit is measured with
JOB=${1:?job}; TAG=${2:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
noids() { grep -v amdgpu.ids; }
case $JOB in
full)
  timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > gpurun_out/${TAG}_pytest.txt
  python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.txt 2>&1
  timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
  tail -3 gpurun_out/${TAG}_pytest.txt; tail -2 gpurun_out/${TAG}_smoke.txt; tail -c 1500 gpurun_out/${TAG}_bench.json ;;
round)
  bash scripts/gpu_round.sh $TAG > gpurun_out/${TAG}_round.log 2>&1; tail -3 gpurun_out/$TAG/pytest_gpu.log; cat gpurun_out/$TAG/kernel_stats.csv | cut -c1-200 ;;
panel)
  { timeout 900 python -m pytest tests/test_gpu_kalman_large_d.py -q 2>&1 | tail -3
    timeout 600 python -m pytest tests/test_gpu_baseline_configs.py -x -q -k config5 2>&1 | tail -2
    for d in 64 48 40; do python scripts/bench_big.py --iters 10 --d $d 2>&1 | noids; done
    python scripts/bench_big.py --iters 10 --batch 64 2>&1 | noids
    echo "== operators at config 5's shape (scripts/bench_bigops.py)"; python scripts/bench_bigops.py 2>&1 | noids; } | tee gpurun_out/${TAG}_panel.txt
  bash scripts/pmc_big.sh ${TAG}_panel > gpurun_out/${TAG}_panel_pmc.log 2>&1; cat gpurun_out/${TAG}_panel/pmc_summary.txt | cut -c1-500 ;;
wave)
  { timeout 1200 python -m pytest tests/test_gpu_wave.py tests/test_gpu_large_d_ops.py -q 2>&1 | tail -3
    for dt in f64 f32; do python scripts/bench_wave.py --dims 15,16,17,24,30,32 --dtype $dt 2>&1 | noids; done
    for dt in f64 f32; do python scripts/bench_wave.py --dims 24,32 --dtype $dt --m 8 2>&1 | noids; done
    echo "== operators, B = 512 / 64, T = 1000, f64 (scripts/bench_bigops.py)"
    for d in 16 32; do for b in 512 64; do echo "-- d=$d B=$b"; python scripts/bench_bigops.py --batch $b --T 1000 --d $d --m 1 --dtype f64 --iters 5 2>&1 | noids | grep -v "^$"; done; done
    echo "== kl_divergence pieces (scripts/prof_kl_wave.py)"; python scripts/prof_kl_wave.py 2>&1 | noids | tail -8
    echo "== operator adjoints (scripts/bench_adjoints.py)"; python scripts/bench_adjoints.py 2>&1 | noids
    python scripts/bench_adjoints.py --dtype f32 --dims 16,32 2>&1 | noids; } | tee gpurun_out/${TAG}_wave.txt ;;
fuzz)
  { echo "== fuzz_parity.py 300 17"; timeout 1200 python3 scripts/fuzz_parity.py 300 17 2>&1 | noids | tail -3
    echo "== fuzz_large_d.py 200 19"; timeout 900 python3 scripts/fuzz_large_d.py 200 19 2>&1 | noids | tail -2
    echo "== fuzz_wave.py 200 61"; timeout 900 python3 scripts/fuzz_wave.py 200 61 2>&1 | noids | tail -2
    echo "== stress_determinism.py"; timeout 900 python3 scripts/stress_determinism.py 2>&1 | noids | tail -4; } | tee gpurun_out/${TAG}_fuzz.txt ;;
final)
  bash $0 round $TAG; bash $0 panel $TAG; bash $0 wave $TAG
  timeout 300 python3 scripts/bench_gpr_grad.py --batch 512 --T 1000 --sig 5,5,5 --multi --iters 20 2>&1 | tail -1 | tee gpurun_out/${TAG}_config4_step.txt ;;
*) echo "unknown job $JOB"; exit 2 ;;
esac
