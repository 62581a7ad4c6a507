#!/bin/bash
# Variant build of ONE shape's translation units: the units <parts> (comma list of NJ_PART values)
# of build-table entry <cfg> are recompiled with the given -D flags, every other object comes from
# the product build.  Use:
#   bash tools/ubench/cfg_variant.sh <name> <cfg> <parts> [-DFLAG ...]
#   e.g. q4stamp 6 2,3 -DNJ_Q4_STAMP     (phase stamps of the masked four-wave lockstep kernels)
# then on the GPU:  NJODE_LIB=$PWD/tools/ubench/libnjode_<name>.so python ...
set -e
cd "$(dirname "$0")/../.."
name=$1; cfg=$2; parts=$3; shift 3
OBJ=njode_amd/csrc/_obj
defs=$(python3 - "$cfg" <<'PY'
import sys
sys.path.insert(0, '.')
from njode_amd.build import all_configs
d, h, do, nh, w, act, masked, curt, res, rnn = all_configs()[int(sys.argv[1])]
print('-DNJ_ID={} -DNJ_D={} -DNJ_H={} -DNJ_DO={} -DNJ_NH={} -DNJ_W={} -DNJ_ACT={} -DNJ_MASKED={} -DNJ_CURT={} '
      '-DNJ_RES={} -DNJ_ACC_TANH={} -DNJ_RNN={}'.format(sys.argv[1], d, h, do, nh, max(w, 1), act, masked, curt, res, masked, rnn))
PY
)
skip=""
objs=""
for p in ${parts//,/ }; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$@" $defs -DNJ_PART=$p \
    njode_amd/csrc/njode_cfg.hip -o tools/ubench/cfg${cfg}_${p}_$name.o &
  skip="$skip -e /cfg${cfg}_${p}.o\$"
  objs="$objs tools/ubench/cfg${cfg}_${p}_$name.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ubench/libnjode_$name.so \
  $(ls $OBJ/*.o | grep -v $skip) $objs
echo built tools/ubench/libnjode_$name.so
