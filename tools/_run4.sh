#!/bin/bash
mkdir -p gpurun_out/r3d
timeout 900 python -m pytest tests/test_hip_generic.py -x -q -m gpu -k 'segment_plan or unmasked_shapes or dropout_gradient' 2>&1 | tail -2
for pf in 1 0; do
  echo "== NJODE_GEN_PF=$pf"
  NJODE_GEN_PF=$pf timeout 600 python tools/bench_generic.py 2>/dev/null | grep -E "w100|w200|w400" | cut -c1-330
done | tee gpurun_out/r3d/pf_ab.txt
