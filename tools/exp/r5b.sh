#!/bin/bash
# round 5, GPU call B: why is the tile queue slower per Euler step?  static / queue x 512 / 1024
# blocks with per-tile prologue and step-loop stamps; accurate tanh of the masked kernels against
# the float64 truth; new tests
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r5b
mkdir -p $OUT
B="$ROOT/bench.py --no-cpu-baseline --no-small-batch --no-autograd-route"
rm -f gpurun_out/f64_truth_report.jsonl
timeout 1500 python -m pytest tests/test_hip_f64_truth.py tests/test_hip_config5_full_size.py tests/test_hip_generic.py tests/test_hip_torch_op.py \
  tests/test_hip_parity.py tests/test_hip_paths_per_tile.py -q -m gpu -s 2>&1 | tail -40 > $OUT/pytest.log
cp gpurun_out/f64_truth_report.jsonl $OUT/ 2>/dev/null
run() {   # label, env...
  local label=$1; shift
  env "$@" python3 $B --steps 100 --warmup 20 $EXTRA 2>/dev/null | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']; print('$label', d['ms_per_step'], 'bwd', k.get('k_ode_bwd_mixed'), 'fwd', k.get('k_ode_fwd_mixed'), 'rows', k.get('k_jump_rows_bwd_mfma'), k.get('k_encode_rows_bwd_mfma'), k.get('k_encode_rows_mfma'), 'loss', d['final_loss'])"
}
EXTRA=""
for i in 1 2; do
  run static_1024 NJODE_BWD_QUEUE=0
  run static_512 NJODE_BWD_QUEUE=0 NJODE_BWD_BLOCKS=512
  run queue_512 NJODE_BWD_QUEUE=1
  run queue_1024 NJODE_BWD_QUEUE=1 NJODE_BWD_BLOCKS=1024
  run queue_2048 NJODE_BWD_QUEUE=1 NJODE_BWD_BLOCKS=2048
done > $OUT/ab.txt 2>&1
EXTRA="--dropout 0"
run static_1024_dropout0 NJODE_BWD_QUEUE=0 >> $OUT/ab.txt 2>&1
EXTRA=""
export NJODE_LIB=$ROOT/tools/ubench/libnjode_stamps.so
for cfg in "0 1024" "0 512" "1 512" "1 1024"; do
  set -- $cfg
  echo "=== NJODE_BWD_QUEUE=$1 NJODE_BWD_BLOCKS=$2"
  NJODE_BWD_QUEUE=$1 NJODE_BWD_BLOCKS=$2 python3 tools/ubench/bwd_stamps_run.py --paths 20000 --json $OUT/stamps.jsonl
done > $OUT/stamps.txt 2>&1
unset NJODE_LIB
ls -la $OUT
