#!/bin/bash
OUT=gpurun_out/r5w; mkdir -p $OUT
NJODE_PLAN_DEFER=0 NJODE_PLAN_GRID=1 timeout -k 5 200 python3 tools/exp/r5w.py 2>&1 | grep -v "amdgpu.ids\|using loss\|use residual" | tail -40 > $OUT/log.txt 2>&1
cat $OUT/log.txt
