#!/bin/bash
# copy the round's judged artifacts from gpurun_out/ (scratch) into profiles/ (tracked)
set -e
R=${1:-r06}
cp gpurun_out/${R}_final_bench.json profiles/${R}_final_bench.json
cp gpurun_out/${R}_final_kernel_stats.csv profiles/${R}_final_kernel_stats.csv
cp gpurun_out/${R}_final_pmc_summary.json profiles/${R}_final_pmc_summary.json
cp gpurun_out/${R}_final_trace_trace.txt profiles/${R}_final_step_timeline.txt
[ -f gpurun_out/${R}_final/stats_bench.json ] && cp gpurun_out/${R}_final/stats_bench.json profiles/${R}_final_bench_under_rocprof.json
for f in ${R}_autograd_route_timeline.txt ${R}_config5_kernels.jsonl ${R}_small_batch_kernels.txt ${R}_generic_bench.jsonl ${R}_configs_sweep.jsonl; do
  [ -f gpurun_out/$f ] && cp gpurun_out/$f profiles/$f
done
ls -la profiles/${R}_*
