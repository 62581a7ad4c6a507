"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel (mean per launch).
Usage: python tools/summarize_pmc.py [--paths-per-gpu N --dropout P --command "..."]
           <dir-with-*_counter_collection.csv> [...] > summary.json
The workload flags are recorded under "_workload": bench.py reports `roofline.traffic` from
the summary only when it is benchmarking that same workload.
FETCH_SIZE / WRITE_SIZE are reported in KiB as rocprofv3 emits them (MI355X_MICROARCH.md
section HBM: on gfx950 FETCH_SIZE under-reports wide coalesced streaming reads by 2x;
narrow accesses are uncalibrated)."""
import collections
import csv
import glob
import json
import sys


def short(name):
    name = name.replace('void njode::', '').replace('(anonymous namespace)::', '')
    return name.split('<')[0].split('(')[0]


argv = sys.argv[1:]
workload = {}
while argv and argv[0].startswith('--'):
    key, val = argv[0][2:].replace('-', '_'), argv[1]
    workload[key] = int(val) if key == 'paths_per_gpu' else (float(val) if key == 'dropout' else val)
    argv = argv[2:]
out = collections.defaultdict(dict)
if workload:
    out['_workload'] = workload
for d in argv:
    for path in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        meta = {}
        for r in csv.DictReader(open(path)):
            k = short(r['Kernel_Name'])
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
            meta[k] = {'vgpr': int(r['VGPR_Count']), 'agpr': int(r['Accum_VGPR_Count']),
                       'sgpr': int(r['SGPR_Count']), 'lds_bytes': int(r['LDS_Block_Size']),
                       'scratch': int(r['Scratch_Size']), 'grid': int(r['Grid_Size']),
                       'workgroup': int(r['Workgroup_Size'])}
        for k, counters in agg.items():
            if not k.startswith('k_'):
                continue
            out[k].update(meta[k])
            for c, v in counters.items():
                out[k][c] = round(sum(v) / len(v), 3)
                out[k]['launches'] = len(v)
print(json.dumps(out, indent=1, sort_keys=True))
