#!/bin/bash
mkdir -p gpurun_out/r3d
python bench.py > gpurun_out/r3d/bench_final3.json 2> gpurun_out/r3d/bench_final3.err
head -c 200 gpurun_out/r3d/bench_final3.json; echo
