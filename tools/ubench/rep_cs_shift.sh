run() {
  env "$@" python3 bench.py --no-cpu-baseline --no-small-batch --steps 40 2>/dev/null | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'env': sys.argv[1:], 'ms_per_step': d['ms_per_step']}))" "$@"
}
for i in 1 2; do for s in 7 8 9 10; do run NJODE_CS_SHIFT=$s; done; run NJODE_SORT=rocprim; run NJODE_PLAN_STREAM=0; done
