#!/bin/bash
mkdir -p gpurun_out/r3d
for nw in 0 2 4 5 8; do
  echo "== NJODE_GEN_NW=$nw"
  if [ $nw = 0 ]; then unset NJODE_GEN_NW; else export NJODE_GEN_NW=$nw; fi
  timeout 600 python tools/bench_generic.py 2>/dev/null | grep -E "w100|w400" | cut -c1-330
done | tee gpurun_out/r3d/nw_sweep.txt
