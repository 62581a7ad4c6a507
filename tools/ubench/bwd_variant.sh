#!/bin/bash
# Variant builds of the demo shape's segment-backward unit (cfg0_1: k_ode_bwd_mixed and the row
# gradient kernels): only that unit is recompiled with the given -D flags, the other objects come
# from the product build.  Use:
#   bash tools/ubench/bwd_variant.sh <name> [-DFLAG ...]      (e.g. stamps -DNJ_BWD_STAMPS)
# then on the GPU:  NJODE_LIB=$PWD/tools/ubench/libnjode_<name>.so python bench.py ...
set -e
cd "$(dirname "$0")/../.."
name=$1; shift
OBJ=njode_amd/csrc/_obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$@" \
  -DNJ_ID=0 -DNJ_PART=1 -DNJ_D=1 -DNJ_H=10 -DNJ_DO=1 -DNJ_NH=2 -DNJ_W=50 -DNJ_ACT=0 -DNJ_MASKED=0 \
  -DNJ_CURT=0 -DNJ_RES=1 -DNJ_ACC_TANH=0 -DNJ_RNN=0 njode_amd/csrc/njode_cfg.hip -o tools/ubench/cfg0_1_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ubench/libnjode_$name.so \
  $(ls $OBJ/*.o | grep -v '/cfg0_1.o$') tools/ubench/cfg0_1_$name.o
echo built tools/ubench/libnjode_$name.so
