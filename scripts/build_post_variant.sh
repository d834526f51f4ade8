#!/bin/bash
# A/B build of the streamed posterior kernels: bash scripts/build_post_variant.sh <tag> <extra hipcc flags...>
#   -> markovflow_amd/libmf_post_<tag>.so (every other object from the regular build); use with MF_LIB_PATH=...
set -e
R=$(cd $(dirname $0)/.. && pwd); C=$R/markovflow_amd/csrc; TAG=$1; shift
mkdir -p $C/build_var_$TAG
for d in 1 2 3 4 5 6; do
  if [ "$d" = "6" ] || [ -n "$ALL_DIMS" ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DMF_D=$d "$@" -c $C/mf_post_inst.hip -o $C/build_var_$TAG/mf_post_d$d.o &
  else cp $C/build/mf_post_d$d.o $C/build_var_$TAG/; fi
done; wait
OBJS=$(ls $C/build/*.o | grep -v mf_post_d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/markovflow_amd/libmf_post_$TAG.so $OBJS $C/build_var_$TAG/*.o
echo built $R/markovflow_amd/libmf_post_$TAG.so
