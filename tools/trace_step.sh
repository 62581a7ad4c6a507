#!/bin/bash
# Timeline of the kernels of one training step (run on the GPU box from the repo root):
#   bash tools/trace_step.sh <tag>   -> gpurun_out/<tag>_trace.txt
# rocprofv3 --kernel-trace of `python3 bench.py --steps 6`, then the kernels of the 4th timed step
# in start order with their queue, duration and the gap to the previous kernel's end.
set -u
TAG=${1:-r02_trace}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o trace -- python3 $ROOT/bench.py \
  --no-cpu-baseline --no-small-batch --no-config5 --no-kernel-timing --steps 6 --warmup 2 > $OUT/bench.json 2> $OUT/trace.err
cd $ROOT
python3 - "$OUT" > gpurun_out/${TAG}_trace.txt <<'PY'
import csv, glob, sys
rows = []
for p in glob.glob(sys.argv[1] + '/trace/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?')))
rows.sort()
adam = [i for i, r in enumerate(rows) if 'k_adam' in r[2]]
# kernels between the 5th-last and 4th-last Adam launches = one whole step
lo, hi = adam[-3] + 1, adam[-2] + 1
t0 = rows[lo][0]
prev_end = rows[lo - 1][1]
print('step wall (Adam end to Adam end): %.1f us' % ((rows[hi - 1][1] - rows[lo - 1][1]) / 1e3))
for s, e, n, q in rows[lo:hi]:
    n = n.replace('void njode::', '').replace('(anonymous namespace)::', '').split('<')[0].split('(')[0][:44]
    print('%9.1f  +%7.1f us  dur %7.1f  q%s  %s' % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, q, n))
    prev_end = max(prev_end, e)
PY
