"""GPU tests of the batch producer (include/njode_producer.h) through the C ABI:
the Philox stream against the oracle restatement, the SDE recurrences and the observation
mask against the host generators (which tests/golden pins to the reference) when fed the
reference's own draws, the device collate against the host collate (bit-exact), and
size-independent statistics of the Philox-driven datasets."""
import ctypes as C

import numpy as np
import pytest
import torch

from hip_util import demo_cfg, hip_model
from njode_amd import _lib, data_utils, device_data, stock_model
from oracle import producer_oracle as po

pytestmark = pytest.mark.gpu
HP = dict(data_utils.hyperparam_default)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def test_device_philox_matches_the_oracle_word_for_word():
    rng = np.random.RandomState(0)
    n = 4096
    ctr = rng.randint(0, 2 ** 32, size=(n, 4), dtype=np.uint64).astype(np.uint32)
    key = rng.randint(0, 2 ** 32, size=(n, 2), dtype=np.uint64).astype(np.uint32)
    ctr[0], key[0] = 0, 0                       # known-answer vector rides along
    d_ctr = torch.from_numpy(ctr.view(np.int32)).cuda()
    d_key = torch.from_numpy(key.view(np.int32)).cuda()
    out = torch.empty((n, 4), dtype=torch.int32, device='cuda')
    _lib.check(_lib.lib().njode_philox4x32_10(n, d_ctr.data_ptr(), d_key.data_ptr(),
                                              out.data_ptr(), _stream()))
    got = out.cpu().numpy().view(np.uint32)
    np.testing.assert_array_equal(got, po.philox4x32_10(ctr, key))
    assert tuple(int(x) for x in got[0]) == (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)


@pytest.mark.parametrize('name', ['BlackScholes', 'OrnsteinUhlenbeck', 'Heston'])
@pytest.mark.parametrize('sine', [None, 2.0])
def test_recurrences_reproduce_the_host_generator_on_the_reference_draws(name, sine):
    """Same numpy draws (legacy seed protocol of create_dataset) -> the device recurrences
    must give the host generator's float64 paths and its observation mask."""
    hp = dict(HP, nb_paths=257, nb_steps=50, sine_coeff=sine)
    paths, observed, nb_obs, meta = data_utils.create_dataset(name, hp, seed=11)
    np.random.seed(11)
    per = (257, 50, 2, 1) if name == 'Heston' else (257, 50, 1)
    normals = np.random.normal(0, 1, per)
    uniforms = np.random.random(size=(257, 51))
    ds = device_data.DeviceDataset.generate(name, hp, seed=0, normals=normals, uniforms=uniforms)
    got_paths, got_obs, got_nb = ds.to_arrays()
    np.testing.assert_array_equal(got_obs, observed)
    np.testing.assert_array_equal(got_nb, nb_obs)
    if sine is None and name != 'Heston':
        np.testing.assert_array_equal(got_paths, paths)          # bit-exact float64
    else:  # sin / sqrt are not correctly rounded on either side: a few ulp per step
        np.testing.assert_allclose(got_paths, paths, rtol=1e-12, atol=0)
    assert ds.metadata['dt'] == meta['dt']


def test_philox_driven_generation_follows_the_oracle_stream():
    """Without supplied draws the kernels consume Philox(seed; path, step, dim): feeding the
    oracle's restatement of that stream through the host recurrences gives the same paths."""
    hp = dict(HP, nb_paths=64, nb_steps=30)
    ds = device_data.DeviceDataset.generate('BlackScholes', hp, seed=0x1234567890ab)
    got_paths, got_obs, _ = ds.to_arrays()
    z1 = po.step_normals(64, 30, 1, seed=0x1234567890ab)
    dt = hp['maturity'] / 30
    ref = np.empty((64, 1, 31))
    ref[:, :, 0] = hp['S0']
    for k in range(1, 31):
        prev = ref[:, :, k - 1]
        ref[:, :, k] = prev + hp['drift'] * prev * dt + hp['volatility'] * prev * (z1[:, k - 1, :] * np.sqrt(dt))
    np.testing.assert_allclose(got_paths, ref, rtol=1e-12)   # log / sincospi differ by ulps
    u = po.observation_uniforms(64, 30, seed=0x1234567890ab)
    np.testing.assert_array_equal(got_obs, (u < hp['obs_perc']) * 1)


@pytest.mark.parametrize('name', ['BlackScholes', 'OrnsteinUhlenbeck', 'Heston'])
def test_generated_datasets_have_the_models_moments(name):
    """100 000 paths: E[X_T] follows the model's conditional-expectation recursion (the Euler
    scheme's mean is exact for all three drifts), observation rate = obs_perc, seeds differ."""
    hp = dict(HP, nb_paths=100000, nb_steps=100)
    ds = device_data.DeviceDataset.generate(name, hp, seed=5)
    xT = ds.paths_tm[-1, 0].cpu().numpy()
    dt = hp['maturity'] / 100
    if name == 'OrnsteinUhlenbeck':
        mean = hp['mean'] + (hp['S0'] - hp['mean']) * (1 - hp['speed'] * dt) ** 100
    else:
        mean = hp['S0'] * (1 + hp['drift'] * dt) ** 100
    assert abs(xT.mean() - mean) < 5 * xT.std() / np.sqrt(len(xT))
    assert np.isfinite(xT).all() and xT.std() > 0
    obs = ds.observed_tm.float().mean().item()
    assert abs(obs - hp['obs_perc']) < 5 * np.sqrt(0.09 / ds.observed_tm.numel())
    np.testing.assert_array_equal(ds.nb_obs.cpu().numpy(),
                                  ds.observed_tm[1:].sum(0).cpu().numpy())
    ds2 = device_data.DeviceDataset.generate(name, hp, seed=6)
    assert not torch.equal(ds2.paths_tm[-1], ds.paths_tm[-1])
    ds3 = device_data.DeviceDataset.generate(name, hp, seed=5)
    assert torch.equal(ds3.paths_tm, ds.paths_tm) and torch.equal(ds3.observed_tm, ds.observed_tm)


def _assert_same_batch(got, ref):
    np.testing.assert_array_equal(got['times'], ref['times'])
    np.testing.assert_array_equal(got['time_ptr'], ref['time_ptr'])
    np.testing.assert_array_equal(got['obs_idx'].cpu().numpy(), ref['obs_idx'].numpy())
    np.testing.assert_array_equal(got['n_obs_ot'].cpu().numpy(), ref['n_obs_ot'].numpy())
    assert torch.equal(got['X'].cpu(), ref['X'])
    assert torch.equal(got['start_X'].cpu(), ref['start_X'])


@pytest.mark.parametrize('n_paths,dim,funcs', [(1000, 1, ()), (700, 3, ()), (333, 2, ('power-2',)),
                                               (64, 1, ('power-2', 'power-3'))])
def test_device_collate_is_bit_exact(n_paths, dim, funcs):
    hp = dict(HP, nb_paths=n_paths, nb_steps=60, S0=[1.0] * dim if dim > 1 else 1)
    paths, observed, nb_obs, meta = data_utils.create_dataset('BlackScholes', hp, seed=2)
    observed[:, 17] = 0           # a grid time without observations
    nb_obs = observed[:, 1:].sum(1)
    ds = device_data.DeviceDataset.from_arrays(paths, observed, nb_obs, meta)
    fns = [data_utils._get_func(f) for f in funcs]
    # whole dataset
    _assert_same_batch(ds.collate(func_names=funcs),
                       data_utils.collate_arrays(paths, observed, nb_obs, meta['dt'], fns))
    # shuffled sub-batches of awkward sizes (not multiples of the wave / workgroup)
    rng = np.random.RandomState(1)
    for B in (1, 63, 65, 257, min(n_paths, 513)):
        idx = rng.permutation(n_paths)[:B]
        _assert_same_batch(ds.collate(idx, func_names=funcs),
                           data_utils.collate_arrays(paths[idx], observed[idx], nb_obs[idx],
                                                     meta['dt'], fns))


def test_collate_of_a_batch_without_observations():
    hp = dict(HP, nb_paths=8, nb_steps=20)
    paths, observed, nb_obs, meta = data_utils.create_dataset('BlackScholes', hp, seed=2)
    observed[3, :] = 0
    ds = device_data.DeviceDataset.from_arrays(paths, observed, observed[:, 1:].sum(1), meta)
    b = ds.collate([3])
    assert len(b['times']) == 0 and b['time_ptr'].tolist() == [0] and b['X'].shape == (0, 1)
    assert b['start_X'].cpu().numpy().tolist() == [[np.float32(paths[3, 0, 0])]]


def test_c_abi_rejects_bad_arguments():
    L = _lib.lib()
    sde = _lib.NjodeSde()
    sde.model, sde.n_paths, sde.dim, sde.n_steps = 7, 4, 1, 4
    buf = torch.empty(64, dtype=torch.float64, device='cuda')
    assert L.njode_generate_paths(C.byref(sde), 0, None, buf.data_ptr(), _stream()) == _lib.E_UNSUPPORTED
    assert b'unknown SDE model' in L.njode_last_error()
    sde.model, sde.n_steps = 0, 0
    assert L.njode_generate_paths(C.byref(sde), 0, None, buf.data_ptr(), _stream()) == _lib.E_BADARG
    assert L.njode_collate_count(None, None, 4, 4, None, 4, None, None, _stream()) == _lib.E_BADARG
    pw = (C.c_int32 * 1)(-1)
    assert L.njode_collate_fill(buf.data_ptr(), buf.data_ptr(), 4, 1, 4, None, 4, buf.data_ptr(),
                                pw, 1, buf.data_ptr(), None, None, _stream()) == _lib.E_BADARG


def test_training_step_on_a_device_collated_batch_equals_the_host_collated_one():
    hp = dict(HP, nb_paths=500, nb_steps=100)
    paths, observed, nb_obs, meta = data_utils.create_dataset('BlackScholes', hp, seed=0)
    ds = device_data.DeviceDataset.from_arrays(paths, observed, nb_obs, meta)
    idx = np.random.RandomState(0).permutation(500)[:200]
    host = data_utils.collate_arrays(paths[idx], observed[idx], nb_obs[idx], meta['dt'])
    dev = ds.collate(idx)
    torch.manual_seed(0)
    m = hip_model(demo_cfg()).train()
    out = []
    for b in (host, dev):
        _, loss = m.loss_and_grad(b['times'], b['time_ptr'], b['X'].cuda(), b['obs_idx'].cuda().int(),
                                  meta['dt'], meta['maturity'], b['start_X'].cuda(),
                                  b['n_obs_ot'].cuda().int())
        out.append((float(loss), m.flat_grad().clone()))
    assert out[0][0] == out[1][0] and torch.equal(out[0][1], out[1][1])
