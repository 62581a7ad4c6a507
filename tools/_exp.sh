for pt in 1 0; do
NJODE_GEN_PT=$pt python tools/bench_generic_physio.py 2>/dev/null | sed "s/^/GEN_PT=$pt /" | cut -c1-420
done
python tools/bench_generic.py 2>/dev/null > gpurun_out/r04_generic_bench.jsonl; cut -c1-300 gpurun_out/r04_generic_bench.jsonl
