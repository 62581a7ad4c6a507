#!/bin/bash
mkdir -p gpurun_out/r3d
export NJODE_LIB=$PWD/tools/ubench/libnjode_bwd_direct.so
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_properties.py -x -q -m gpu -k "not masked and not g5 and not distribution" 2>&1 | tail -2
for i in 1 2; do python bench.py --no-cpu-baseline --no-small-batch --no-autograd-route 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('direct', d['ms_per_step'], d['kernel_ms']['k_ode_bwd_mixed'])"; done
unset NJODE_LIB
for i in 1 2; do python bench.py --no-cpu-baseline --no-small-batch --no-autograd-route 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('product', d['ms_per_step'], d['kernel_ms']['k_ode_bwd_mixed'])"; done
