#!/bin/bash
mkdir -p gpurun_out/r3d
for bits in 0 63 64 128 191 192; do
  for r in 2.25; do
    NJODE_SPLIT_R_BWD=$r NJODE_LIB=$PWD/tools/ubench/libnjode_bwdabl_$bits.so python bench.py --no-cpu-baseline --no-small-batch --no-autograd-route --steps 10 --warmup 3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print(json.dumps({'abl_bits': $bits, 'split_r_bwd': $r, 'ms_per_step': d['ms_per_step'], 'k_ode_bwd_mixed_ms': d['kernel_ms']['k_ode_bwd_mixed']}))"
  done
done | tee gpurun_out/r3d/bwd_ablate2.jsonl
