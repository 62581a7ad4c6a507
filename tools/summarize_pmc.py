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
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, 'njode_amd', 'libnjode_hip.so')
RES = os.path.join(ROOT, 'njode_amd', 'csrc', '_obj', 'kernel_resources.json')


def lib_sha256():
    h = hashlib.sha256()
    with open(LIB, 'rb') as f:
        for chunk in iter(lambda: f.read(1 << 20), b''):
            h.update(chunk)
    return h.hexdigest()


def compiler_resources():
    """{demangled kernel name: resources} from the build's -Rpass-analysis remarks
    (njode_amd/build.py).  rocprofv3's VGPR_Count column is only the ARCH half of gfx950's unified
    register file (k_ode_bwd_mixed: 128 there, 256 in the code object); the compiler's number
    is what decides the waves per SIMD."""
    try:
        with open(RES) as f:
            res = json.load(f)
    except OSError:
        return {}
    names = list(res)
    dem = names
    for tool in ('c++filt', '/opt/rocm/lib/llvm/bin/llvm-cxxfilt'):
        try:
            p = subprocess.run([tool], input='\n'.join(names), stdout=subprocess.PIPE, text=True,
                               check=True)
            dem = p.stdout.splitlines()
            break
        except (OSError, subprocess.CalledProcessError):
            continue
    return {d.replace('void ', '', 1): res[n] for n, d in zip(names, dem)}



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
    # the library the counters were taken on: bench.measured_traffic refuses any other build
    workload['lib_sha256'] = lib_sha256()
    out['_workload'] = workload
cres = compiler_resources()
for d in argv:
    for path in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        meta = {}
        for r in csv.DictReader(open(path)):
            k = short(r['Kernel_Name'])
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
            meta[k] = {'arch_vgpr_rocprof': int(r['VGPR_Count']), 'agpr_rocprof': int(r['Accum_VGPR_Count']),
                       'sgpr': int(r['SGPR_Count']), 'lds_bytes': int(r['LDS_Block_Size']),
                       'scratch': int(r['Scratch_Size']), 'grid': int(r['Grid_Size']),
                       'workgroup': int(r['Workgroup_Size'])}
            full = r['Kernel_Name'].replace('void ', '', 1)
            cr = cres.get(full) or cres.get(full.split('(')[0])
            if cr is None:      # (rocprofv3 may print the name without its parameter list)
                hits = [v for n, v in cres.items() if n.split('(')[0] == full.split('(')[0]]
                cr = hits[0] if hits else None
            if cr:
                meta[k].update({'vgpr': cr.get('vgpr'), 'agpr': cr.get('agpr'),
                                'vgpr_spill': cr.get('vgpr_spill'), 'occupancy_waves_per_simd': cr.get('occupancy')})
        for k, counters in agg.items():
            if not k.startswith('k_'):
                continue
            out[k].update(meta[k])
            for c, v in counters.items():
                out[k][c] = round(sum(v) / len(v), 3)
                out[k]['launches'] = len(v)
print(json.dumps(out, indent=1, sort_keys=True))
