# full GPU check of the round: test suite, bench, profiling recipe, auxiliary benches
mkdir -p gpurun_out/r3f
timeout 3000 python -m pytest tests -q -m gpu -x --durations=8 > gpurun_out/r3f/gputests3.log 2>&1; echo "rc=$?" >> gpurun_out/r3f/gputests3.log
tail -14 gpurun_out/r3f/gputests3.log | cut -c1-150
python bench.py > gpurun_out/r3f/bench_final.json 2> gpurun_out/r3f/bench_final.err
bash tools/profile_recipe.sh r03_final > gpurun_out/r3f/recipe.log 2>&1; tail -3 gpurun_out/r3f/recipe.log
python bench.py > gpurun_out/r3f/bench_final2.json 2> gpurun_out/r3f/bench_final2.err; head -c 300 gpurun_out/r3f/bench_final2.json; echo
python tools/bench_harness.py > gpurun_out/r3f/harness_final.jsonl 2>/dev/null; cat gpurun_out/r3f/harness_final.jsonl | cut -c1-400
python tools/bench_physionet.py > gpurun_out/r3f/physionet.jsonl 2>/dev/null; tail -3 gpurun_out/r3f/physionet.jsonl | cut -c1-300
