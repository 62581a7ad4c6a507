run() {
  env "$@" python3 bench.py --no-cpu-baseline --no-small-batch --steps 40 2>/dev/null | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print(json.dumps({'env': sys.argv[1:], 'ms_per_step': d['ms_per_step'], 'bwd': k.get('k_ode_bwd_mixed'), 'fwd': k.get('k_ode_fwd_mfma', k.get('k_ode_fwd_mixed'))}))" "$@"
}
for i in 1 2 3; do run NJODE_BWD_BLOCKS=1024; run NJODE_BWD_BLOCKS=1536; run NJODE_BWD_BLOCKS=1024 NJODE_SPLIT_BWD_BLOCKS=32;  run NJODE_BWD_BLOCKS=768; done
