#!/bin/bash
mkdir -p gpurun_out/r3d
NJODE_LIB=$PWD/tools/ubench/libnjode_hip_stamps.so timeout 600 python tools/ubench/gen_split.py 2>/dev/null | tee gpurun_out/r3d/stamps2.jsonl
