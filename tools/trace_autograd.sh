#!/bin/bash
# kernels of one step of the autograd route in start order: bash tools/trace_autograd.sh <tag>
TAG=${1:-autograd_trace}; ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o trace -- python3 $ROOT/tools/autograd_step.py 8 > $OUT/run.log 2> $OUT/run.err
cd $ROOT
python3 - "$OUT" <<'PY'
import csv, glob, sys
rows = []
for p in glob.glob(sys.argv[1] + '/trace/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?')))
rows.sort()
packs = [i for i, r in enumerate(rows) if 'k_pack_all' in r[2]]
lo, hi = packs[-3], packs[-2]
t0 = rows[lo][0]; prev_end = rows[lo - 1][1]
print('step wall (pack to pack): %.1f us' % ((rows[hi][0] - rows[lo][0]) / 1e3))
for s, e, n, q in rows[lo:hi]:
    n = n.replace('void njode::', '').replace('(anonymous namespace)::', '').replace('void at::native::', '').split('<')[0].split('(')[0][:46]
    print('%9.1f  +%7.1f us  dur %7.1f  q%s  %s' % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, q, n))
    prev_end = max(prev_end, e)
PY
cat $OUT/run.log
