run() {
  env "$@" python3 bench.py --no-cpu-baseline --no-small-batch --steps 40 2>/dev/null | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print(json.dumps({'env': sys.argv[1:], 'ms_per_step': d['ms_per_step'], 'enc': k.get('k_encode_rows_mfma'), 'enc_bwd': k.get('k_encode_rows_bwd_mfma')}))" "$@"
}
for i in 1 2; do for b in 2048 4096 8192 16384; do run NJODE_ENC_BLOCKS=$b; done; done
