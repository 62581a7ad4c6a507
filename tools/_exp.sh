python -m pytest tests -x -q -m gpu 2>&1 | tail -15
python bench.py > gpurun_out/r4f/bench.json 2> gpurun_out/r4f/bench.err; head -c 600 gpurun_out/r4f/bench.json; echo
DROPOUT=0.1 python tools/bench_physionet.py 2>&1 | tail -4
