#!/bin/bash
# Ablation builds of the one-wave role of k_ode_bwd_mixed (njode_ode2.h, NJ_BWD_ABL bits): only
# the segment-backward unit of the demo shape is recompiled, the other objects come from the
# product build.  Use: bash tools/ubench/bwd_ablate.sh 0 1 2 ... ; then on the GPU
#   NJODE_LIB=$PWD/tools/ubench/libnjode_bwdabl_<bits>.so python bench.py --no-cpu-baseline ...
set -e
cd "$(dirname "$0")/../.."
OBJ=njode_amd/csrc/_obj
for bits in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -DNJ_BWD_ABL=$bits ${NJ_BWD_EXTRA} \
    -DNJ_ID=0 -DNJ_PART=1 -DNJ_D=1 -DNJ_H=10 -DNJ_DO=1 -DNJ_NH=2 -DNJ_W=50 -DNJ_ACT=0 -DNJ_MASKED=0 \
    -DNJ_CURT=0 -DNJ_RES=1 -DNJ_ACC_TANH=0 -DNJ_RNN=0 njode_amd/csrc/njode_cfg.hip -o tools/ubench/cfg0_1_abl$bits.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ubench/libnjode_bwdabl_$bits.so \
    $(ls $OBJ/*.o | grep -v '/cfg0_1.o$') tools/ubench/cfg0_1_abl$bits.o
  echo built tools/ubench/libnjode_bwdabl_$bits.so
done
