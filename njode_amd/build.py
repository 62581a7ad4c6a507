"""
Build libnjode_hip.so (gfx950) in-tree with hipcc.

The kernels keep each lane's activations in registers, so every model *shape*
is a separate template instantiation.  ``CONFIGS`` is the table of compiled
shapes; ``NJODE_EXTRA_CONFIGS`` (env, ``;``-separated
``d,H,d_out,n_hidden,width,act,masked,current_t,residual,use_rnn``) appends to it.
Each shape is compiled as six translation units (segment forward, segment
backward, lockstep forward, lockstep backward, wave-per-path lockstep forward / sweep) so the
build parallelises over the host cores.

Usage:  python -m njode_amd.build [--force] [-j N]
"""
import argparse
import concurrent.futures
import hashlib
import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(CSRC, '_obj')
LIB = os.path.join(HERE, 'libnjode_hip.so')
ARCH = 'gfx950'

TANH, RELU = 0, 1
# (d, H, d_out, n_hidden, width, act, masked, input_current_t, residual, use_rnn)
CONFIGS = [
    (1, 10, 1, 2, 50, TANH, 0, 0, 1, 0),    # demo.py: BlackScholes / OU / Heston (configs 1-4)
    (1, 10, 1, 2, 50, TANH, 0, 1, 1, 0),    # options['input_current_t']
    (1, 10, 1, 2, 50, TANH, 0, 0, 0, 0),    # options['residual_enc_dec'] = False
    (2, 10, 2, 2, 50, TANH, 0, 0, 1, 0),    # func_appl_X=['power-2']
    (1, 10, 1, 2, 20, RELU, 0, 0, 1, 0),    # relu nets
    (1, 10, 1, 0, 0, TANH, 0, 0, 1, 0),     # nn_desc=None (single Linear per net)
    (41, 41, 41, 2, 50, TANH, 1, 0, 1, 0),  # PhysioNet shape, reference setting (H=41 residual)
    (41, 50, 41, 2, 50, TANH, 1, 0, 0, 0),  # PhysioNet shape, BASELINE config 5 wording (H=50)
    (1, 10, 1, 2, 50, TANH, 0, 0, 1, 1),    # use_rnn=True: GRU jump (models.py:202-217)
    # network widths of the reference's convergence study that fit the matrix-core kernels
    # (parallel_train.py:304-305: 10, 20, 40, 80, 160, 320; wider ones: NJODE_EXTRA_CONFIGS /
    # compile on first use, see models.NJODE._get_dims)
    (1, 10, 1, 2, 10, TANH, 0, 0, 1, 0),
    (1, 10, 1, 2, 20, TANH, 0, 0, 1, 0),
    (1, 10, 1, 2, 40, TANH, 0, 0, 1, 0),
]


def all_configs():
    cfgs = list(CONFIGS)
    extra = os.environ.get('NJODE_EXTRA_CONFIGS', '').strip()
    for item in filter(None, extra.split(';')):
        t = tuple(int(x) for x in item.split(','))
        if len(t) != 10:
            raise ValueError('NJODE_EXTRA_CONFIGS entries need 10 integers: ' + item)
        if t not in cfgs:
            cfgs.append(t)
    return cfgs


def _hipcc():
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return 'hipcc'


def _digest(files, extra=''):
    h = hashlib.sha256()
    for path in files:
        with open(path, 'rb') as f:
            h.update(os.path.basename(path).encode())
            h.update(f.read())
    h.update(extra.encode())
    return h.hexdigest()


_RES_KEYS = {'VGPRs': 'vgpr', 'AGPRs': 'agpr', 'SGPRs': 'sgpr', 'ScratchSize [bytes/lane]': 'scratch',
             'Occupancy [waves/SIMD]': 'occupancy', 'LDS Size [bytes/block]': 'lds_bytes',
             'VGPRs Spill': 'vgpr_spill', 'SGPRs Spill': 'sgpr_spill'}


_SNIPPET = re.compile(r'^\s*\d+\s*\|')     # source snippet lines of a remark ('  415 | {')


def _parse_resources(out):
    """(kernel -> resources, the rest of the compiler output) from -Rpass-analysis remarks."""
    res, rest, cur = {}, [], None
    for line in out.splitlines():
        if 'remark:' in line and 'kernel-resource-usage' in line:
            body = line.split('remark:', 1)[1].split('[-Rpass-analysis')[0].strip()
            if body.startswith('Function Name:'):
                cur = body.split(':', 1)[1].strip()
                res[cur] = {}
            elif cur is not None and ':' in body:
                k, v = body.rsplit(':', 1)
                k = _RES_KEYS.get(k.strip())
                if k:
                    try:
                        res[cur][k] = int(v)
                    except ValueError:
                        pass
        elif line.strip() and not line.lstrip().startswith(('|', '^', 'In file included from')) \
                and not _SNIPPET.match(line) \
                and '__global__' not in line and 'remarks generated' not in line:
            rest.append(line)
    return res, '\n'.join(rest)


def _run(cmd):
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if p.returncode != 0:
        raise RuntimeError('command failed: {}\n{}'.format(' '.join(cmd), p.stdout))
    return p.stdout


def build(force=False, jobs=None, verbose=True):
    cfgs = all_configs()
    only = os.environ.get('NJODE_CONFIGS_ONLY', '').strip()   # e.g. "0": quick kernel experiments
    if only:
        keep = {int(x) for x in only.split(',')}
        cfgs = [c for i, c in enumerate(cfgs) if i in keep]
    os.makedirs(OBJ, exist_ok=True)
    inc = ''.join('NJ_CFG({})\n'.format(i) for i in range(len(cfgs)))
    inc_path = os.path.join(CSRC, '_generated_cfgs.inc')
    if not os.path.exists(inc_path) or open(inc_path).read() != inc:
        with open(inc_path, 'w') as f:
            f.write(inc)
    cc = _hipcc()
    # (-Rpass-analysis: the compiler's per-kernel register / scratch / LDS / occupancy remarks,
    # parsed into _obj/kernel_resources.json -- what tools/summarize_pmc.py reports as a
    # kernel's register count: rocprofv3's VGPR_Count column is the ARCH half of gfx950's
    # unified register file only)
    common = [cc, '--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC', '-c',
              '-Rpass-analysis=kernel-resource-usage']
    common += os.environ.get('NJODE_EXTRA_FLAGS', '').split()   # kernel experiments
    hdr = os.path.join(os.path.dirname(HERE), 'include', 'njode_hip.h')
    hdr_prod = os.path.join(os.path.dirname(HERE), 'include', 'njode_producer.h')
    hdr_self = os.path.join(os.path.dirname(HERE), 'include', 'njode_selftest.h')
    kernel_deps = [os.path.join(CSRC, n) for n in
                   ('njode_cfg.hip', 'njode_host.h', 'njode_kernels.h', 'njode_device.h',
                    'njode_mfma.h', 'njode_mfma_rows.h', 'njode_lockstep_bwd.h',
                    'njode_mfma_lockstep.h', 'njode_mfma_split.h', 'njode_ode2.h',
                    'njode_mfma_lock4.h', 'njode_plan.h')] + [hdr]
    chain_deps = kernel_deps + [os.path.join(CSRC, n) for n in ('njode_chain.h', 'njode_dpp.h', 'njode_chain_seg.h', 'njode_chain_dw.h')]
    gen_deps = [os.path.join(CSRC, n) for n in
                ('njode_gen.hip', 'njode_gen.h', 'njode_gen_seg.h', 'njode_gen_host.h', 'njode_device.h',
                 'njode_error.h')] + [hdr]
    api_deps = [os.path.join(CSRC, n) for n in
                ('njode_api.hip', 'njode_gen_host.h', 'njode_host.h', 'njode_kernels.h', 'njode_device.h',
                 'njode_mfma.h', 'njode_mfma_rows.h', 'njode_lockstep_bwd.h',
                 'njode_mfma_lockstep.h', 'njode_mfma_split.h', 'njode_ode2.h',
                 'njode_mfma_lock4.h', 'njode_error.h', 'njode_plan.h',
                 '_generated_cfgs.inc')] + [hdr, hdr_self]
    prod_deps = [os.path.join(CSRC, n) for n in ('njode_producer.hip', 'njode_error.h')] + [
        hdr, hdr_prod]
    tasks = []   # (object, command, digest)
    for i, (d, h, do, nh, w, act, masked, curt, res, rnn) in enumerate(cfgs):
        for part in range(6):
            obj = os.path.join(OBJ, 'cfg{}_{}.o'.format(i, part))
            defs = ['-DNJ_ID={}'.format(i), '-DNJ_PART={}'.format(part), '-DNJ_D={}'.format(d),
                    '-DNJ_H={}'.format(h), '-DNJ_DO={}'.format(do), '-DNJ_NH={}'.format(nh),
                    '-DNJ_W={}'.format(max(w, 1)), '-DNJ_ACT={}'.format(act),
                    '-DNJ_MASKED={}'.format(masked), '-DNJ_CURT={}'.format(curt),
                    '-DNJ_RES={}'.format(res), '-DNJ_ACC_TANH={}'.format(masked), '-DNJ_RNN={}'.format(rnn)]
            cmd = common + defs + [os.path.join(CSRC, 'njode_cfg.hip'), '-o', obj]
            tasks.append((obj, cmd, _digest(chain_deps if part >= 4 else kernel_deps, ' '.join(cmd))))
    api_obj = os.path.join(OBJ, 'api.o')
    cmd = common + [os.path.join(CSRC, 'njode_api.hip'), '-o', api_obj]
    tasks.append((api_obj, cmd, _digest(api_deps, ' '.join(cmd))))
    gen_obj = os.path.join(OBJ, 'gen.o')
    cmd = common + [os.path.join(CSRC, 'njode_gen.hip'), '-o', gen_obj]
    tasks.append((gen_obj, cmd, _digest(gen_deps, ' '.join(cmd))))
    # the batch producer spells out the reference's float64 expression trees: no contraction
    prod_obj = os.path.join(OBJ, 'producer.o')
    cmd = common + ['-ffp-contract=off', os.path.join(CSRC, 'njode_producer.hip'), '-o', prod_obj]
    tasks.append((prod_obj, cmd, _digest(prod_deps, ' '.join(cmd))))
    if os.environ.get('NJODE_RESTAMP'):   # maintainer aid: adopt the objects on disk as current
        for t in tasks:
            if os.path.exists(t[0]):
                with open(t[0] + '.stamp', 'w') as f:
                    f.write(t[2])

    def stale(t):
        stamp = t[0] + '.stamp'
        return (force or not os.path.exists(t[0]) or not os.path.exists(stamp)
                or open(stamp).read() != t[2])

    todo = [t for t in tasks if stale(t)]
    parts_only = os.environ.get('NJODE_PARTS_ONLY', '').strip()   # maintainer aid, e.g. "0": a change
    if parts_only:                                                 # that only touches those parts
        keep = tuple('_{}.o'.format(x) for x in parts_only.split(',')) + ('api.o', 'producer.o', 'gen.o')
        for t in todo:
            if not t[0].endswith(keep) and os.path.exists(t[0]):
                with open(t[0] + '.stamp', 'w') as f:
                    f.write(t[2])
        todo = [t for t in todo if t[0].endswith(keep) or not os.path.exists(t[0])]
    if not todo and os.path.exists(LIB) and not force:
        if verbose:
            print('[njode_amd.build] up to date:', LIB)
        return LIB
    jobs = jobs or int(os.environ.get('NJODE_BUILD_JOBS', os.cpu_count() or 4))
    if verbose:
        print('[njode_amd.build] compiling {} of {} units for {} with {} jobs ...'.format(
            len(todo), len(tasks), ARCH, jobs))

    def compile_one(t):
        out = _run(t[1])
        res, rest = _parse_resources(out)
        with open(t[0] + '.res.json', 'w') as f:
            json.dump(res, f)
        with open(t[0] + '.stamp', 'w') as f:
            f.write(t[2])
        return rest

    # heaviest units first (masked 41-dim lockstep kernels dominate the wall time)
    todo.sort(key=lambda t: 0 if t[0].endswith(('_2.o', '_3.o')) else 1)
    with concurrent.futures.ThreadPoolExecutor(max_workers=jobs) as ex:
        for out in ex.map(compile_one, todo):
            if out.strip() and verbose:
                print(out)
    _run([cc, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', LIB] + [t[0] for t in tasks])
    merged = {}
    for t in tasks:
        if os.path.exists(t[0] + '.res.json'):
            with open(t[0] + '.res.json') as f:
                for k, v in json.load(f).items():
                    merged.setdefault(k, v)
    with open(os.path.join(OBJ, 'kernel_resources.json'), 'w') as f:
        json.dump(merged, f, indent=1, sort_keys=True)
    if verbose:
        print('[njode_amd.build] built', LIB)
    return LIB


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--force', action='store_true')
    ap.add_argument('-j', type=int, default=None)
    args = ap.parse_args()
    build(force=args.force, jobs=args.j)
    sys.exit(0)
