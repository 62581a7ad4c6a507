#!/bin/bash
# Round-end evidence, in the order the artifacts depend on each other (run on the GPU box from the
# repo root; tools/collect_profiles.sh then copies gpurun_out/<R>_* into profiles/):
#   1. the GPU test suite; 2. rocprofv3 stats + PMC passes of `python3 bench.py` -> <R>_final_*;
#   3. the PMC summary goes to profiles/ FIRST, so that 4. `python bench.py` (the driver's command)
#   finds the counters of this very library (roofline.traffic); 5. timelines and side benches.
R=${1:-r06}
set -o pipefail
# (the suite's verdict travels with the artifacts; profiles of a library whose suite is red are not
# evidence: abort before they are produced)
python -m pytest tests -x -q -m gpu --durations=20 2>&1 | tail -40 | tee gpurun_out/${R}_final_pytest_tail.txt | tail -5
rc=$?
echo "pytest -m gpu rc=$rc" > gpurun_out/${R}_final_pytest_rc.txt
if [ $rc -ne 0 ]; then echo "GPU suite failed (rc $rc): no profiles taken"; exit $rc; fi
bash tools/profile_recipe.sh ${R}_final > gpurun_out/${R}_final_recipe.out 2>&1
cp gpurun_out/${R}_final_pmc_summary.json profiles/${R}_final_pmc_summary.json
python bench.py > gpurun_out/${R}_final_bench.json 2> gpurun_out/${R}_final_bench.err
head -c 300 gpurun_out/${R}_final_bench.json; echo
bash tools/trace_step.sh ${R}_final_trace > /dev/null; head -24 gpurun_out/${R}_final_trace_trace.txt
bash tools/trace_autograd.sh ${R}_autograd > gpurun_out/${R}_autograd_route_timeline.txt 2>&1
DROPOUT=0.1 BATCHES=50,800,1000,2048 python tools/bench_physionet.py 2>/dev/null | grep config5 > gpurun_out/${R}_config5_kernels.jsonl
DROPOUT=0.0 BATCHES=50,800,1000,2048 python tools/bench_physionet.py 2>/dev/null | grep config5 >> gpurun_out/${R}_config5_kernels.jsonl
NJODE_CHAIN_MAX=0 DROPOUT=0.1 BATCHES=50,800,1000,2048 python tools/bench_physionet.py 2>/dev/null | grep config5 | sed 's/config5-kernels/config5-kernels (NJODE_CHAIN_MAX=0: the matrix-core tiles)/' >> gpurun_out/${R}_config5_kernels.jsonl
NJODE_CHAIN_DELTA=0 DROPOUT=0.1 BATCHES=50,800,1000,2048 python tools/bench_physionet.py 2>/dev/null | grep config5 | sed 's/config5-kernels/config5-kernels (NJODE_CHAIN_DELTA=0: the recomputing pair kernel behind the sweeps)/' >> gpurun_out/${R}_config5_kernels.jsonl
bash tools/small_stats.sh ${R}_small100 100 > gpurun_out/${R}_small_batch_kernels.txt; bash tools/small_stats.sh ${R}_small200 200 >> gpurun_out/${R}_small_batch_kernels.txt
python tools/bench_generic.py 2>/dev/null | grep "^{" > gpurun_out/${R}_generic_bench.jsonl
python tools/bench_configs.py 2>/dev/null | grep "^{" > gpurun_out/${R}_configs_sweep.jsonl
python -c "
import json
b=json.load(open('gpurun_out/${R}_final_bench.json'))
print('FINAL', b['ms_per_step'], b['value'], b['roofline']['frac'], b['roofline']['kernel_ms'], b['roofline'].get('traffic'), b['autograd_route_ms'], b['b100_ms'], b['b200_ms'])
"
