#!/bin/bash
# Ablation for the "line ring" plan of the headline kernel: an -DMF_EXPERIMENT build with MF_KF_DEBUG=8 does
# NOT fetch the units of A / cholQ rows that lie in a first line shared with the previous row - i.e. it moves the lines a ring
# would move (results are garbage) - at 64 and 48 chunks per series.  Kernel time by HIP events, FETCH_SIZE in its own pass.
#   make -C markovflow_amd/csrc BUILD=build_exp OUT=../libmarkovflow_amd_exp.so EXTRA=-DMF_EXPERIMENT
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-exp_ring}; mkdir -p $OUT
export MF_LIB_PATH=$R/markovflow_amd/libmarkovflow_amd_exp.so
cd /tmp && export TMPDIR=/tmp
for DBG in 0 8; do for CH in 64 48; do
  export MF_KF_DEBUG=$DBG
  python3 $R/scripts/prof_kf.py --chunks $CH --iters 10 --nan-ok 2>&1 | tail -1 | sed "s/^/debug=$DBG /"
  rm -rf /tmp/pmc_r
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex "kf_chunk_lds" --output-format csv -d /tmp/pmc_r -- python3 $R/scripts/prof_kf.py --chunks $CH --iters 3 --nan-ok > $OUT/pmc_${DBG}_$CH.log 2>&1
  f=$(find /tmp/pmc_r -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" $DBG $CH <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
v = [float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == "FETCH_SIZE"]
n = len(set(r["Dispatch_Id"] for r in rows))
print(f"debug={sys.argv[2]} chunks={sys.argv[3]}: FETCH_SIZE {sum(v) / n:.0f} KiB per launch -> L2-fill traffic {2 * sum(v) / n * 1024 / 1e9:.2f} GB (algorithmic 6.96 GB)")
PY
done; done
