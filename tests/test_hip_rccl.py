"""The N > 1 path on RCCL itself: ``bench.py --gpus 2`` with one rank per
GPU and the collective over ``nccl`` (= RCCL on ROCm), self-spawned, against the one-rank run
over the same global dataset.  Needs two GPUs: on the one-GPU test boxes it SKIPS (visibly) --
the shared-GPU gloo runs of tests/test_hip_config4.py cover the same code there -- and the first
multi-GPU box that runs the suite exercises RCCL through it."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from hip_util import rel_l2

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(tmp_path, tag, gpus, extra):
    dump = str(tmp_path / (tag + '.npy'))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'NJODE_BENCH_SHARE_GPU'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', str(gpus), '--steps', '3',
           '--warmup', '1', '--no-cpu-baseline', '--no-small-batch', '--no-autograd-route',
           '--rank-timeout', '600', '--dump-params', dump] + extra
    p = subprocess.run(cmd, env=env, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0]), np.load(dump)


@pytest.mark.skipif(torch.cuda.device_count() < 2,
                    reason='RCCL needs two GPUs: this box has {}'.format(torch.cuda.device_count()))
def test_bench_two_ranks_over_rccl_match_the_one_rank_run(tmp_path):
    two, p2 = _bench(tmp_path, 'rccl2', 2, ['--paths-per-gpu', '20000'])
    one, p1 = _bench(tmp_path, 'rccl1', 1, ['--paths-per-gpu', '40000', '--no-kernel-timing'])
    assert two['n_gpus'] == 2 and two['rccl_world'] == 2
    assert two['collective_backend'].startswith('nccl')          # RCCL, one rank per GPU
    assert two['allreduce_floats'] == 10071 + 1                  # gradient + the scalar loss
    assert two['allreduce_ms'] is not None and two['allreduce_ms'] > 0
    assert two['params_identical_across_ranks'] is True
    assert two['config']['global_batch'] == 40000 == one['config']['global_batch']
    # same global dataset, same dropout masks; only the fp32 summation order differs (1e-5)
    assert two['final_loss'] == pytest.approx(one['final_loss'], rel=1e-5)
    assert rel_l2(p2, p1) < 1e-5
