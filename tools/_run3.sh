#!/bin/bash
# scratch: segment plan of the generic kernels -- plan agreement, oracle parity, bench
set -x
mkdir -p gpurun_out/r3c
timeout 1500 python -m pytest tests/test_hip_generic.py -x -q -m gpu -k 'segment_plan or unmasked_shapes or shards_add_up or dropout_gradient' > gpurun_out/r3c/t.log 2>&1
echo rc=$?
tail -30 gpurun_out/r3c/t.log
timeout 900 python tools/bench_generic.py > gpurun_out/r3c/bench_generic.jsonl 2> gpurun_out/r3c/bench_generic.err
cat gpurun_out/r3c/bench_generic.jsonl
tail -3 gpurun_out/r3c/bench_generic.err
