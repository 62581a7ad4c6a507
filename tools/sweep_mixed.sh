#!/bin/bash
# Sweep of the mixed ODE kernels' grid / cost-model constants on the headline workload
# (run on the GPU box from the repo root): one bench line per setting, kernel_ms inside.
run() {
  env "$@" python3 bench.py --no-cpu-baseline --no-small-batch --steps 20 2>/dev/null | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print(json.dumps({'env': sys.argv[1:], 'ms_per_step': d['ms_per_step'], 'bwd': k.get('k_ode_bwd_mixed'), 'fwd': k.get('k_ode_fwd_mfma', k.get('k_ode_fwd_mixed'))}))" "$@"
}
run A=0
for b in 1024 2048 2560 3072; do run NJODE_BWD_BLOCKS=$b; done
for s in 16 24 48 64 96; do run NJODE_SPLIT_BWD_BLOCKS=$s; done
for r in 1.75 2.0 2.5 2.75 3.0; do run NJODE_SPLIT_R_BWD=$r; done
for b in 2048 4096 6144; do run NJODE_FWD_BLOCKS=$b; done
for s in 64 128 192; do run NJODE_SPLIT_FWD_BLOCKS=$s; done
for r in 1.5 1.75 2.25 2.5; do run NJODE_SPLIT_R_FWD=$r; done
