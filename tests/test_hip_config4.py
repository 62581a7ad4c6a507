"""BASELINE config 4 (1 M Black-Scholes paths sharded over 8 GPUs = 125 000 paths per rank)
on the one GPU of the test box:

* ``bench.py --gpus 2`` itself -- no launcher: the parent spawns the ranks -- with the two ranks
  sharing device 0 (``NJODE_BENCH_SHARE_GPU=1``, collective over gloo), in weak mode and in
  strong mode at config 4's shard size, against the one-rank run over the same global dataset;
* size-independent properties at 125 000 paths: shards add up, a 2 000-path slice matches the
  oracle, the two execution plans agree.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from hip_util import LOSS_RTOL, demo_cfg, hip_forward, hip_model, oracle_forward, rel_l2, to_dev
from njode_amd import data_utils

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHARD = 125000        # config 4: 1 000 000 paths / 8 ranks


def _bench(tmp_path, tag, gpus, extra, share):
    dump = str(tmp_path / (tag + '.npy'))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    if share:
        env['NJODE_BENCH_SHARE_GPU'] = '1'
    cmd = [sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', str(gpus), '--steps', '3',
           '--warmup', '1', '--no-cpu-baseline', '--no-small-batch', '--no-autograd-route',
           '--dump-params', dump] + extra
    p = subprocess.run(cmd, env=env, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=1500)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, p.stdout[-2000:]           # rank 0 prints ONE JSON line
    return json.loads(lines[0]), np.load(dump)


def _check_pair(two, p2, one, p1, global_batch, scaling):
    for key in ('n_gpus', 'rccl_world', 'collective_backend', 'allreduce_ms', 'allreduce_floats',
                'ms_per_step_by_rank', 'params_identical_across_ranks', 'value', 'ms_per_step'):
        assert key in two, key
    assert two['n_gpus'] == 2 and two['rccl_world'] == 2
    assert two['scaling'] == scaling
    assert two['collective_backend'].startswith('gloo')        # shared-GPU self-test
    assert two['allreduce_floats'] == 10071 + 1                 # gradient + the scalar loss, one bucket
    assert two['allreduce_ms'] is not None and two['allreduce_ms'] > 0
    assert len(two['ms_per_step_by_rank']) == 2
    assert two['params_identical_across_ranks'] is True
    assert two['config']['global_batch'] == global_batch == one['config']['global_batch']
    assert 'roofline' not in two and 'cpu_baseline' not in two      # N = 1 only
    assert two['value'] == pytest.approx(global_batch / (two['ms_per_step'] * 1e-3), rel=1e-3)
    # same global dataset, same dropout masks (keyed by the global path id); only the fp32
    # summation order differs between one and two shards (1e-5)
    assert two['final_loss'] == pytest.approx(one['final_loss'], rel=1e-5)
    assert rel_l2(p2, p1) < 1e-5


def test_bench_self_spawns_two_ranks_weak_scaling(tmp_path):
    two, p2 = _bench(tmp_path, 'weak2', 2, ['--paths-per-gpu', '20000'], share=True)
    one, p1 = _bench(tmp_path, 'weak1', 1, ['--paths-per-gpu', '40000', '--no-kernel-timing'],
                     share=False)
    assert one['n_gpus'] == 1 and 'rccl_world' not in one
    _check_pair(two, p2, one, p1, 40000, 'weak')
    # the weak-scaling line also carries the strong-scaling reading of BASELINE's 20 000-path batch
    assert two['strong_20k_ms'] > 0 and two['strong_20k_paths_per_s'] == pytest.approx(
        20000 / (two['strong_20k_ms'] * 1e-3), rel=1e-3)
    assert '10000 paths on rank 0' in two['strong_20k_note']


def test_bench_self_spawns_two_ranks_strong_scaling_at_config4_shard_size(tmp_path):
    g = 2 * SHARD
    two, p2 = _bench(tmp_path, 'strong2', 2, ['--global-paths', str(g)], share=True)
    one, p1 = _bench(tmp_path, 'strong1', 1, ['--global-paths', str(g), '--no-kernel-timing'],
                     share=False)
    assert two['config']['paths_rank0'] == SHARD
    _check_pair(two, p2, one, p1, g, 'strong')
    assert 'strong_20k_ms' not in two          # (only beside a WEAK headline)


def test_bench_refuses_more_ranks_than_gpus_without_the_self_test_switch():
    if torch.cuda.device_count() >= 2:
        pytest.skip('needs a one-GPU box')
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'NJODE_BENCH_SHARE_GPU')}
    p = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2', '--steps', '1',
                        '--warmup', '0'], env=env, cwd=REPO, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode != 0
    assert 'GPU' in p.stderr


# ---- properties at the shard size of config 4 ------------------------------------------------
@pytest.fixture(scope='module')
def shard():
    sys.path.insert(0, REPO)
    import bench
    b, meta = bench.make_global_slice(0, SHARD)
    torch.manual_seed(0)
    m = hip_model(demo_cfg()).eval()
    return b, meta, m


def _sub_batch(b, meta, idx):
    return data_utils.collate_arrays(b['true_paths'][idx], b['observed_dates'][idx],
                                     b['observed_dates'][idx][:, 1:].sum(1), meta['dt'])


def _args(b, meta):
    d = to_dev(b)
    return (d['times'], d['time_ptr'], d['X'], d['obs_idx'], meta['dt'], meta['maturity'],
            d['start_X'], d['n_obs_ot'])


def test_125k_shards_add_up_and_a_slice_matches_the_oracle(shard):
    b, meta, m = shard
    m.train()                     # dropout_rate = 0 here
    try:
        m.dp_global_batch, m.dp_path_offset = 8 * SHARD, 3 * SHARD     # rank 3 of config 4
        _, loss = m.loss_and_grad(*_args(b, meta))
        g_full = m.flat_grad().clone()
        assert torch.isfinite(g_full).all()
        total, g_sum = 0.0, torch.zeros_like(g_full)
        cuts = [0, 2000, 33000, 64001, 100000, SHARD]                  # ragged shards
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            bs = _sub_batch(b, meta, np.arange(lo, hi))
            m.dp_global_batch, m.dp_path_offset = 8 * SHARD, 3 * SHARD + lo
            _, l = m.loss_and_grad(*_args(bs, meta))
            total += float(l)
            g_sum += m.flat_grad()
        assert total == pytest.approx(float(loss), rel=2e-5)
        assert rel_l2(g_sum.cpu().numpy(), g_full.cpu().numpy()) < 1e-4
        # the first 2 000 paths against the oracle (denominator 2 000: an ordinary batch)
        m.dp_global_batch, m.dp_path_offset = None, 0
        bs = _sub_batch(b, meta, np.arange(2000))
        sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
        _, l_hip = m.loss_and_grad(*_args(bs, meta))
        g_hip = m.flat_grad().cpu().numpy().copy()
        (_, l_o), params = oracle_forward(demo_cfg(), sd, bs, meta['dt'], meta['maturity'],
                                          training=True, grads=True)
        l_o.backward()
        g_o = np.concatenate([params[k].grad.reshape(-1).numpy() for k in sd])
        assert float(l_hip) == pytest.approx(float(l_o), rel=LOSS_RTOL)
        assert rel_l2(g_hip, g_o) < 1e-3
    finally:
        m.dp_global_batch, m.dp_path_offset = None, 0
        m.eval()


def test_125k_two_plans_agree(shard):
    """segment plan (loss only) and lockstep plan (return_path) are independent kernels."""
    b, meta, m = shard
    with torch.no_grad():
        hT_s, loss_s = hip_forward(m, b, meta['dt'], meta['maturity'])
        hT_l, loss_l, _, _, _ = hip_forward(m, b, meta['dt'], meta['maturity'],
                                            return_path=True, get_loss=True)
    assert float(loss_s) == pytest.approx(float(loss_l), rel=2e-5)
    np.testing.assert_allclose(hT_s.cpu().numpy(), hT_l.cpu().numpy(), atol=1e-5, rtol=1e-4)
