"""The N > 1 path on CPU: world_size-2 gloo processes shard a global batch by path,
compute partial gradients (the oracle stands in for the kernels -- test
infrastructure), all-reduce the flat gradient and take the identical Adam step.
The summed result must equal the single-process gradient of the whole batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from golden_util import Golden
from njode_amd import data_utils, parallel
from njode_amd.train import split_indices
from oracle import njode_oracle


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _flat_grad(model, params, batch, dt, T, scale):
    for p in params.values():
        p.grad = None
    _, loss = model.forward(params, batch['times'], batch['time_ptr'], batch['X'],
                            batch['obs_idx'], dt, T, batch['start_X'], batch['n_obs_ot'])
    (loss * scale).backward()
    return float(loss) * scale, torch.cat([params[k].grad.reshape(-1) for k in params])


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    w, r, _ = parallel.init_distributed('gloo')
    assert (w, r) == (world, rank)
    g = Golden('g2_bs_grads_B64')
    model = njode_oracle.make_oracle(g.cfg)
    params = {k: v.clone().requires_grad_(True) for k, v in g.state_dict().items()}
    # rank 1 starts from garbage; the broadcast must repair it
    flat = torch.cat([p.detach().reshape(-1) for p in params.values()])
    if rank == 1:
        flat = flat + 1.0
    parallel.broadcast_parameters_(flat, src=0)
    off = 0
    for p in params.values():
        p.data.copy_(flat[off:off + p.numel()].view_as(p))
        off += p.numel()

    hp = dict(data_utils.hyperparam_default, nb_paths=64)
    paths, obs, nb_obs, meta = data_utils.create_dataset('BlackScholes', hp, seed=0)
    B = 37                                                   # ragged: 19 + 18
    perm = parallel.epoch_permutation(64, epoch=3, seed=0)[:B]
    lo, hi = parallel.shard_range(B, world, rank)
    mine = perm[lo:hi]
    local = data_utils.collate_arrays(paths[mine], obs[mine], nb_obs[mine], meta['dt'])
    full = data_utils.collate_arrays(paths[perm], obs[perm], nb_obs[perm], meta['dt'])
    dt, T = meta['dt'], meta['maturity']

    l_loc, g_loc = _flat_grad(model, params, local, dt, T, scale=(hi - lo) / B)
    parallel.allreduce_flat_(g_loc)
    l_sum = float(parallel.allreduce_flat_(torch.tensor([l_loc], dtype=torch.float64)))
    l_full, g_full = _flat_grad(model, params, full, dt, T, scale=1.0)
    assert l_sum == pytest.approx(l_full, rel=1e-5)
    assert float((g_loc - g_full).norm() / g_full.norm()) < 1e-5

    # identical Adam step on every rank from the all-reduced gradient
    opt = torch.optim.Adam(list(params.values()), lr=1e-3, weight_decay=0.0005)
    off = 0
    for p in params.values():
        p.grad = g_loc[off:off + p.numel()].view_as(p).clone()
        off += p.numel()
    opt.step()
    after = torch.cat([p.detach().reshape(-1) for p in params.values()])
    gathered = [torch.zeros_like(after) for _ in range(world)]
    dist.all_gather(gathered, after)
    assert torch.equal(gathered[0], gathered[1])
    np.save(os.path.join(out_dir, 'ok{}.npy'.format(rank)), np.array([l_sum]))
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_matches_single_process(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = np.load(tmp_path / 'ok0.npy')
    b = np.load(tmp_path / 'ok1.npy')
    assert a[0] == b[0]


class _FlatModel:
    """What parallel.empty_shard_step touches of the model (the flat gradient)."""

    def __init__(self, n):
        self._g = torch.full((n,), 7.0)        # stale gradient of the previous step

    def flat_grad(self):
        return self._g


def _worker_empty_shard(rank, world, port, out_dir):
    """Last partial batch of an epoch with fewer paths than ranks: the rank with
    the empty shard must not run the kernels, must contribute zeros and must still join the
    collective -- the loop of njode_amd/train.py, with the oracle as the compute."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    parallel.init_distributed('gloo')
    g = Golden('g2_bs_grads_B64')
    model = njode_oracle.make_oracle(g.cfg)
    params = {k: v.clone().requires_grad_(True) for k, v in g.state_dict().items()}
    hp = dict(data_utils.hyperparam_default, nb_paths=8)
    paths, obs, nb_obs, meta = data_utils.create_dataset('BlackScholes', hp, seed=0)
    idx = np.array([5])                                      # ONE path, two ranks
    lo, hi = parallel.shard_range(len(idx), world, rank)
    mine = idx[lo:hi]
    dt, T = meta['dt'], meta['maturity']
    n_flat = sum(p.numel() for p in params.values())
    fm = _FlatModel(n_flat)
    if len(mine) == 0:
        loss = parallel.empty_shard_step(fm, fused=True)
        assert float(loss) == 0.0
        g_loc = fm.flat_grad()
    else:
        local = data_utils.collate_arrays(paths[mine], obs[mine], nb_obs[mine], dt)
        _, g_loc = _flat_grad(model, params, local, dt, T, scale=(hi - lo) / len(idx))
    parallel.allreduce_flat_(g_loc)
    full = data_utils.collate_arrays(paths[idx], obs[idx], nb_obs[idx], dt)
    _, g_full = _flat_grad(model, params, full, dt, T, scale=1.0)
    assert float((g_loc - g_full).norm() / g_full.norm()) < 1e-6
    np.save(os.path.join(out_dir, 'empty{}.npy'.format(rank)), np.array([hi - lo]))
    dist.destroy_process_group()


def test_rank_with_an_empty_shard_still_joins_the_allreduce(tmp_path):
    port = _free_port()
    mp.spawn(_worker_empty_shard, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    sizes = sorted(int(np.load(tmp_path / 'empty{}.npy'.format(r))[0]) for r in range(2))
    assert sizes == [0, 1]


def test_shard_ranges_partition_the_batch():
    for n in (1, 7, 64, 100, 20000):
        for world in (1, 2, 3, 8):
            cover = []
            for r in range(world):
                lo, hi = parallel.shard_range(n, world, r)
                cover += list(range(lo, hi))
                assert 0 <= hi - lo <= n // world + 1
            assert cover == list(range(n))


def test_epoch_permutation_is_rank_independent_and_changes_per_epoch():
    a = parallel.epoch_permutation(100, 1)
    assert np.array_equal(a, parallel.epoch_permutation(100, 1))
    assert not np.array_equal(a, parallel.epoch_permutation(100, 2))
    assert sorted(a.tolist()) == list(range(100))


def test_split_matches_sklearn():
    skl = pytest.importorskip('sklearn.model_selection')
    tr, va = split_indices(20000, 0.2, 398)
    tr2, va2 = skl.train_test_split(np.arange(20000), test_size=0.2, random_state=398)
    assert np.array_equal(tr, tr2) and np.array_equal(va, va2)
    assert len(tr) == 16000 and len(va) == 4000
