#!/bin/bash
# per-kernel times of the small-batch step: bash tools/small_stats.sh <tag> [B]
TAG=$1; B=${2:-100}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st -o st -- python3 $ROOT/tools/small_step.py $B 200 > $OUT/small.log 2> $OUT/small.err
cd $ROOT
python3 - "$OUT" <<'PY'
import csv, glob, sys
for p in glob.glob(sys.argv[1] + '/st/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(p)))
    rows.sort(key=lambda r: -float(r['TotalDurationNs']))
    for r in rows[:14]:
        print('%-60s calls %5s avg %8.1f us' % (r['Name'].replace('void njode::', '')[:60], r['Calls'], float(r['AverageNs']) / 1e3))
PY
cat $OUT/small.log
