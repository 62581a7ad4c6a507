#!/bin/bash
mkdir -p gpurun_out/r3d
python bench.py > gpurun_out/r3d/bench_final3.json 2> gpurun_out/r3d/bench_final3.err
head -c 300 gpurun_out/r3d/bench_final3.json; echo
timeout 900 python -m pytest tests/test_hip_plan_prefetch.py -x -q -m gpu 2>&1 | tail -2
