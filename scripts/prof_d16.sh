#!/bin/bash
# Where the time goes at d = 16 (tile engine) against d = 15 (row kernels), B=512, T=1000, fp64: kernel stats and wave counters.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/scripts/sweep_d.py --dims 15,16,24,32 --m 1 2>&1 | tail -4
rm -rf /tmp/pd16 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pd16 -- python3 $R/scripts/sweep_d.py --dims 16 --m 1 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/pd16/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f))):
    if "mf::" in r["Name"]: print(f"{r['Name'][:100]:100s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e6:8.3f} ms")
PY
rm -rf /tmp/pd16b && rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --kernel-include-regex "big_kf_chunk" --output-format csv -d /tmp/pd16b -- python3 $R/scripts/sweep_d.py --dims 16 --m 1 > /dev/null 2>&1
python3 $R/scripts/pmc_sum.py /tmp/pd16b 2>/dev/null | head -5
