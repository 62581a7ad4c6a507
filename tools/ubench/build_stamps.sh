#!/bin/bash
# Diagnostic build of the generic kernels with the s_memtime phase stamps (-DNJ_GEN_STAMPS):
# only njode_gen.hip is recompiled, the other objects come from the product build.
# Use: NJODE_LIB=tools/ubench/libnjode_hip_stamps.so python tools/ubench/gen_split.py
set -e
cd "$(dirname "$0")/../.."
OBJ=njode_amd/csrc/_obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DNJ_GEN_STAMPS ${NJ_DIAG_FLAGS} -c njode_amd/csrc/njode_gen.hip -o tools/ubench/gen_stamps.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ubench/libnjode_hip_stamps.so \
  $(ls $OBJ/*.o | grep -v '/gen.o$') tools/ubench/gen_stamps.o
echo built tools/ubench/libnjode_hip_stamps.so
