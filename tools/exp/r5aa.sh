#!/bin/bash
OUT=gpurun_out/r5aa; mkdir -p $OUT
timeout 600 python -m pytest tests/test_hip_switches.py tests/test_hip_parity.py -q -m gpu -x 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/st -o st -- python3 $OLDPWD/bench.py --no-cpu-baseline --no-small-batch --no-autograd-route --no-kernel-timing --steps 50 --warmup 10 > $OLDPWD/$OUT/bench.json 2>/dev/null
cd $OLDPWD
find $OUT/st -name '*kernel_stats.csv' -exec head -14 {} \; | cut -c1-150
