# full GPU check of the round: test suite, bench, profiling recipe, auxiliary benches
mkdir -p gpurun_out/r3f
timeout 3000 python -m pytest tests -q -m gpu -x > gpurun_out/r3f/gputests.log 2>&1; echo "rc=$?" >> gpurun_out/r3f/gputests.log
tail -5 gpurun_out/r3f/gputests.log
python bench.py > gpurun_out/r3f/bench.json 2> gpurun_out/r3f/bench.err; head -c 600 gpurun_out/r3f/bench.json; echo
bash tools/profile_recipe.sh r03_final > gpurun_out/r3f/recipe.log 2>&1; tail -3 gpurun_out/r3f/recipe.log
