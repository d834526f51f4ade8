#!/bin/bash
# rocprofv3 kernel stats + separate PMC passes for ONE configuration of scripts/prof_cfg.py, with the shipped library and -
# when the experiment build exists (make EXTRA=-DMF_EXPERIMENT BUILD=build_exp OUT=../libmarkovflow_amd_exp.so) - with a knob
# that selects the previous kernel:   scripts/pmc_cfg.sh <tag> <cfg> [ENV=VALUE of the old variant]
# e.g. scripts/pmc_cfg.sh r03_cfg4 c4ll MF_KF_ROW=0      -> gpurun_out/<tag>_{new,old}_{kstats,pmc}.txt
TAG=$1; CFG=$2; OLDENV=$3
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
run_set() {   # $1 = variant name, rest = program and arguments (the program itself goes after `--`: no env / sh hop)
  V=$1; shift
  rm -rf /tmp/ks_$V
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$V -- "$@" > /dev/null 2>&1
  python3 $R/scripts/kstats.py /tmp/ks_$V 40 | grep "mf::" > $OUT/${TAG}_${V}_kstats.txt
  : > $OUT/${TAG}_${V}_pmc.txt
  for CTRS in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_FLAT SQ_WAVES GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"; do
    n=$(echo $CTRS | tr ' ' '_' | cut -c1-30)
    rm -rf /tmp/pmc_${V}_$n
    rocprofv3 --pmc $CTRS --kernel-trace --kernel-include-regex "mf::" --output-format csv -d /tmp/pmc_${V}_$n -- "$@" > /dev/null 2>&1
    python3 $R/scripts/pmc_sum.py /tmp/pmc_${V}_$n >> $OUT/${TAG}_${V}_pmc.txt 2>/dev/null
  done
}
run_set new python3 $R/scripts/prof_cfg.py $CFG 6
if [ -n "$OLDENV" ] && [ -f $R/markovflow_amd/libmarkovflow_amd_exp.so ]; then
  export MF_LIB_PATH=$R/markovflow_amd/libmarkovflow_amd_exp.so
  export "$OLDENV"
  run_set old python3 $R/scripts/prof_cfg.py $CFG 6
fi
tail -n +1 $OUT/${TAG}_*_kstats.txt $OUT/${TAG}_*_pmc.txt
