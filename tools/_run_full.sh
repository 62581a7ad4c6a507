mkdir -p gpurun_out/r3f
timeout 3000 python -m pytest tests -q -m gpu -x --durations=30 > gpurun_out/r3f/gputests2.log 2>&1; echo "rc=$?" >> gpurun_out/r3f/gputests2.log
grep -A34 "slowest" gpurun_out/r3f/gputests2.log | cut -c1-150
tail -3 gpurun_out/r3f/gputests2.log
