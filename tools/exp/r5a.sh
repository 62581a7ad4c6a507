#!/bin/bash
# round 5, GPU call A: tile queue and swizzled images of k_ode_bwd_mixed -- correctness, A/B,
# per-wave stamps, LDS counters; dropout share of the forward's VALU stream
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r5a
mkdir -p $OUT
B="$ROOT/bench.py --no-cpu-baseline --no-small-batch --no-autograd-route"
# ---- 1. correctness of the new default (queue on) and of the static fallback
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_properties.py tests/test_hip_torch_op.py \
  tests/test_hip_f64_truth.py tests/test_hip_train_loop.py -x -q -m gpu -s 2>&1 | tail -25 > $OUT/pytest_queue.log
NJODE_BWD_QUEUE=0 timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "gradients or adam" 2>&1 | tail -5 > $OUT/pytest_static.log
NJODE_LIB=$ROOT/tools/ubench/libnjode_swz.so timeout 600 python -m pytest tests/test_hip_parity.py tests/test_hip_properties.py -x -q -m gpu -k "gradients or adam or linear or propert" 2>&1 | tail -5 > $OUT/pytest_swz.log
# ---- 2. A/B, alternating, 3 rounds
run() {   # label, env...
  local label=$1; shift
  env "$@" python3 $B --steps 100 --warmup 20 2>/dev/null | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']; print('$label', d['ms_per_step'], 'bwd', k.get('k_ode_bwd_mixed'), 'fwd', k.get('k_ode_fwd_mixed'), 'loss', d['final_loss'])"
}
for i in 1 2 3; do
  run static_1024 NJODE_BWD_QUEUE=0
  run queue_512 NJODE_BWD_QUEUE=1
  run swz_queue NJODE_LIB=$ROOT/tools/ubench/libnjode_swz.so NJODE_BWD_QUEUE=1
  run swz_static NJODE_LIB=$ROOT/tools/ubench/libnjode_swz.so NJODE_BWD_QUEUE=0
done > $OUT/ab.txt 2>&1
# queue: number of four-wave blocks / R
for ns in 16 32 64; do for r in 1.5 2.25 3.0; do
  run "queue ns=$ns r=$r" NJODE_BWD_QUEUE=1 NJODE_SPLIT_BWD_BLOCKS=$ns NJODE_SPLIT_R_BWD=$r
done; done > $OUT/ab_split.txt 2>&1
run "queue blocks=768" NJODE_BWD_QUEUE=1 NJODE_BWD_BLOCKS=768 >> $OUT/ab_split.txt 2>&1
# ---- 3. stamps
export NJODE_LIB=$ROOT/tools/ubench/libnjode_stamps.so
for q in 0 1; do for n in 20000 125000; do
  NJODE_BWD_QUEUE=$q python3 tools/ubench/bwd_stamps_run.py --paths $n --json $OUT/stamps.jsonl
done; done > $OUT/stamps.txt 2>&1
unset NJODE_LIB
# ---- 4. counters (program directly after --)
cd /tmp && export TMPDIR=/tmp
REGEX='k_ode|k_jump|k_encode'
pmc() {   # dir name, counters...
  local name=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --output-format csv --kernel-include-regex "$REGEX" -d $OUT/pmc_$name -o pmc -- \
    python3 $B --steps 3 --warmup 1 --no-kernel-timing $EXTRA > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err
  echo "pmc $name rc=$?" >> $OUT/recipe.log
}
EXTRA=""
pmc lds_base SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES
export NJODE_LIB=$ROOT/tools/ubench/libnjode_swz.so
pmc lds_swz SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES
unset NJODE_LIB
pmc valu_d01 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES
EXTRA="--dropout 0"
pmc valu_d00 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES
cd $ROOT
for n in lds_base lds_swz valu_d01 valu_d00; do
  python3 tools/summarize_pmc.py $OUT/pmc_$n > $OUT/pmc_${n}_summary.json 2>> $OUT/recipe.log
  rm -rf $OUT/pmc_$n
done
ls -la $OUT
