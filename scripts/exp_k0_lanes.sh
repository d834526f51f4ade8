#!/bin/bash
# Headline K0 with fewer resident lanes (bench.py --chunks: time partitions per series; lanes = 1024 series x chunks): does the
# L2 keep the partially used 128-B lines between two steps of a lane when fewer lanes compete for it, and what does the
# arithmetic lose with fewer than one wave per SIMD?  For each setting: kernel time by HIP events (bench.py) and FETCH_SIZE per
# launch (rocprofv3 --pmc, its own pass).  Run on the GPU box from the repo root.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-exp_lanes}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for LANES in 65536 49152 32768 16384; do
  CH=$((LANES / 1024))
  python3 $R/bench.py --steps 20 --warmup 3 --chunks $CH --no-cpu-baseline --no-other-configs > $OUT/bench_$LANES.json 2> $OUT/bench_$LANES.err
  python3 - $OUT/bench_$LANES.json $LANES <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(f"lanes {sys.argv[2]}: kernel {j['roofline']['kernel_ms']:.3f} ms, evaluation {j['ms_per_step']:.3f} ms, frac {j['roofline']['frac']:.3f}")
PY
  rm -rf /tmp/pmc_l$LANES
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex "kf_chunk_lds" --output-format csv -d /tmp/pmc_l$LANES -- python3 $R/bench.py --steps 5 --warmup 2 --chunks $CH --no-cpu-baseline --no-other-configs > $OUT/pmc_$LANES.log 2>&1
  f=$(find /tmp/pmc_l$LANES -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" $LANES <<'PY'
import csv, sys
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if r["Counter_Name"] == "FETCH_SIZE"]
n = len(set(r["Dispatch_Id"] for r in csv.DictReader(open(sys.argv[1]))))
print(f"lanes {sys.argv[2]}: FETCH_SIZE {sum(v) / n:.0f} KiB per launch -> L2-fill traffic {2 * sum(v) / n * 1024 / 1e9:.2f} GB (algorithmic 6.96 GB)")
PY
done
