#!/bin/bash
# round 5, GPU call V: which variant of tests/test_hip_switches.py::test_one_launch_plan_is_the_same_plan hangs
set -u
OUT=gpurun_out/r5v; mkdir -p $OUT
run() {  # tag, env...
  local tag=$1; shift
  local t0=$(date +%s)
  env "$@" timeout -k 5 150 python3 -c "
import sys, os
sys.path.insert(0, 'tests')
import test_hip_switches as t
exec(t._SNIPPET_PLAN.format(tests=os.path.abspath('tests'), repo=os.getcwd(), out='$OUT/$tag.npy'))
" > $OUT/$tag.log 2>&1
  echo "$tag rc=$? $(( $(date +%s) - t0 )) s"
}
run legacy NJODE_PLAN_DEFER=0 NJODE_PLAN_GRID=0
run side NJODE_PLAN_DEFER=0 NJODE_PLAN_GRID=1
run default A=0
run defer NJODE_PLAN_DEFER=1 NJODE_PLAN_DEFER_MAX=1000000
run defer_p3 NJODE_PLAN_DEFER=1 NJODE_PLAN_DEFER_MAX=1000000 NJODE_PLAN_BLOCKS=3
run defer_p64 NJODE_PLAN_DEFER=1 NJODE_PLAN_DEFER_MAX=1000000 NJODE_PLAN_BLOCKS=64
run defer_p200 NJODE_PLAN_DEFER=1 NJODE_PLAN_DEFER_MAX=1000000 NJODE_PLAN_BLOCKS=200
python3 -c "
import numpy as np, glob
ref=np.load('$OUT/legacy.npy')
for f in sorted(glob.glob('$OUT/*.npy')):
    a=np.load(f); print(f, a.shape, 'equal' if a.shape==ref.shape and np.array_equal(a,ref) else 'DIFFERENT')
"
tail -3 $OUT/*.log | cut -c1-300
