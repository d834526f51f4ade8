#!/bin/bash
# separate rocprofv3 --pmc passes (one per quoted counter set) over any python script, mf:: kernels only:
#   bash scripts/pmc_any.sh <tag> "<set 1>" "<set 2>" ... -- scripts/x.py [args]     -> gpurun_out/<tag>_pmc.txt
TAG=$1; shift
SETS=()
while [ "$1" != "--" ]; do SETS+=("$1"); shift; done; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/${TAG}_pmc.txt
i=0
for CTRS in "${SETS[@]}"; do
  i=$((i+1)); rm -rf /tmp/pmcany_$i
  timeout 300 rocprofv3 --pmc $CTRS --kernel-trace --kernel-include-regex "mf::" --output-format csv -d /tmp/pmcany_$i -- python3 $R/"$@" > /tmp/pmcany_$i.log 2>&1
  python3 $R/scripts/pmc_sum.py /tmp/pmcany_$i >> $OUT/${TAG}_pmc.txt 2>/dev/null || echo "set failed: $CTRS" >> $OUT/${TAG}_pmc.txt
done
cat $OUT/${TAG}_pmc.txt
