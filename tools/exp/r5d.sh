#!/bin/bash
# round 5, GPU call D: the whole GPU suite on the library with this round's defaults (static rounds,
# swizzled images, accurate tanh, one-launch tail order), the default bench, the autograd route's timeline
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r5d
mkdir -p $OUT
rm -f gpurun_out/f64_truth_report.jsonl
( time timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -40 ) > $OUT/pytest_all.log 2>&1
cp gpurun_out/f64_truth_report.jsonl $OUT/ 2>/dev/null
python bench.py > $OUT/bench.json 2> $OUT/bench.err
head -c 600 $OUT/bench.json; echo
bash tools/trace_autograd.sh r5d_autograd > $OUT/autograd_timeline.txt 2>&1
ls -la $OUT
