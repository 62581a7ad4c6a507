#!/bin/bash
mkdir -p gpurun_out/r3d
timeout 900 python -m pytest tests/test_hip_generic.py -x -q -m gpu -k 'segment_plan or unmasked_shapes or dropout_gradient or shards or malformed' 2>&1 | tail -2
for v in 1 0; do
  echo "== NJODE_GEN_LDSTAB=$v"
  NJODE_GEN_LDSTAB=$v timeout 600 python tools/bench_generic.py 2>/dev/null | grep -E "w100" | cut -c1-330
  NJODE_GENERIC=1 NJODE_GEN_LDSTAB=$v timeout 600 python tools/bench_generic.py 2>/dev/null | head -2 | cut -c1-330
done | tee gpurun_out/r3d/ldstab_ab.txt
