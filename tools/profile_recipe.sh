#!/bin/bash
# Profiling recipe of the headline bench (run on the GPU box from the repo root):
#   bash tools/profile_recipe.sh <tag>          e.g. tag = r03_final
# 1. rocprofv3 --kernel-trace --stats of `python3 bench.py` (the driver's command)
# 2. PMC passes, ONE counter group per pass, kernels of the training step only, 3 steps:
#    SQ group a / SQ group b / GRBM_GUI_ACTIVE (clock) / FETCH_SIZE / WRITE_SIZE
#    (FETCH_SIZE and WRITE_SIZE need 3 + 2 of the 4 TCC slots: never in one pass)
# 3. tools/summarize_pmc.py -> gpurun_out/<tag>_pmc_summary.json (copy it to profiles/)
# The program is named directly after `--` (no env / bash -c hop: see the pool's rules).
set -u
TAG=${1:-r03_final}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="$ROOT/bench.py --no-cpu-baseline --no-small-batch --no-autograd-route --no-config5"
REGEX='k_ode|k_jump|k_encode|k_reduce|k_adam|k_pack|k_row_time|k_dense|k_traj|k_sum'

timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $BENCH --steps 20 \
  > $OUT/stats_bench.json 2> $OUT/stats.err

pmc() {   # name, counters...
  local name=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --output-format csv --kernel-include-regex "$REGEX" -d $OUT/pmc_$name -o pmc -- \
    python3 $BENCH --steps 3 --warmup 1 --no-kernel-timing > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err
  echo "pmc $name rc=$?" >> $OUT/recipe.log
}
pmc sq_a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY
pmc sq_b SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT
pmc grbm GRBM_GUI_ACTIVE
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE

cd $ROOT
python3 tools/summarize_pmc.py --paths-per-gpu 20000 --dropout 0.1 \
  --command "rocprofv3 --pmc <one group per pass> --kernel-include-regex '$REGEX' -- python3 bench.py --no-cpu-baseline --no-small-batch --steps 3 --warmup 1 --no-kernel-timing (tools/profile_recipe.sh)" \
  $OUT/pmc_sq_a $OUT/pmc_sq_b $OUT/pmc_grbm $OUT/pmc_fetch $OUT/pmc_write > gpurun_out/${TAG}_pmc_summary.json 2>> $OUT/recipe.log
# the stats CSV of pass 1
find $OUT/stats -name '*kernel_stats.csv' -exec cp {} gpurun_out/${TAG}_kernel_stats.csv \;
ls -la gpurun_out/${TAG}_* >> $OUT/recipe.log
