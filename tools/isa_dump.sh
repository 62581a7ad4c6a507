#!/bin/bash
# Assembly of one translation unit of the build table's first shape (maintainer aid):
#   tools/isa_dump.sh <part 0..3> [extra -D flags]   ->  /tmp/isa/part<N>.s
set -e
part=$1; shift
mkdir -p /tmp/isa/p$part && cd /tmp/isa/p$part
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -DNJ_ID=0 -DNJ_PART=$part -DNJ_D=1 -DNJ_H=10 \
  -DNJ_DO=1 -DNJ_NH=2 -DNJ_W=50 -DNJ_ACT=0 -DNJ_MASKED=0 -DNJ_CURT=0 -DNJ_RES=1 -DNJ_ACC_TANH=0 -DNJ_RNN=0 "$@" \
  /root/repo/njode_amd/csrc/njode_cfg.hip --save-temps -o cfg.o 2>/dev/null
cp njode_cfg-hip-amdgcn-amd-amdhsa-gfx950.s /tmp/isa/part$part.s
echo /tmp/isa/part$part.s
