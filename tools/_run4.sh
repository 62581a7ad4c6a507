#!/bin/bash
mkdir -p gpurun_out/r3d
timeout 1500 python tools/bench_configs.py > gpurun_out/r3d/configs_sweep.jsonl 2> gpurun_out/r3d/configs_sweep.err
cut -c1-260 gpurun_out/r3d/configs_sweep.jsonl
tail -2 gpurun_out/r3d/configs_sweep.err
