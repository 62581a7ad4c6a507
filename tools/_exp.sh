python -m pytest tests -x -q -m gpu 2>&1 | tail -6
python bench.py > gpurun_out/r4p_bench.json 2> gpurun_out/r4p_bench.err; head -c 400 gpurun_out/r4p_bench.json; echo
