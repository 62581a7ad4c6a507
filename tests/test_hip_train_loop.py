"""GPU test of the build's training loop (njode_amd/train.py, the reproduction of
train.py:488-574): fused and autograd-driven steps agree, the loss goes down, metrics have
the reference's columns, and the loop matches the oracle's loop step for step."""
import copy

import numpy as np
import pytest
import torch

from njode_amd import data_utils, models, train
from oracle import njode_oracle

pytestmark = pytest.mark.gpu


def _dataset(n=600, name='BlackScholes'):
    hp = copy.deepcopy(data_utils.hyperparam_default)
    hp['nb_paths'] = n
    paths, obs, nb_obs, meta = data_utils.create_dataset(name, hp, seed=0)
    return (paths, obs, nb_obs), meta


def test_fused_and_autograd_loops_agree_and_learn():
    data, meta = _dataset()
    logs = []
    m1, met1 = train.train(data, meta, epochs=2, batch_size=100, dropout_rate=0.0, fused=True,
                           log=logs.append)
    m2, met2 = train.train(data, meta, epochs=2, batch_size=100, dropout_rate=0.0, fused=False,
                           log=logs.append)
    assert len(met1) == 2 and len(met1[0]) == len(train.METR_COLUMNS)
    # same batches, same init, same optimizer semantics => same trajectory
    np.testing.assert_allclose(m1.flat_parameters().cpu().numpy(),
                               m2.flat_parameters().cpu().numpy(), atol=5e-5, rtol=1e-3)
    assert met1[1][4] == pytest.approx(met2[1][4], rel=1e-3)         # eval loss
    assert met1[1][4] < met1[0][4]                                   # it learns
    assert met1[0][5] < met1[0][4]                                   # above the optimal loss
    assert m1.epoch == 3 and any('eval-loss' in s for s in logs)


def test_loop_matches_oracle_loop_for_a_few_steps():
    """4 optimizer steps of the harness (shuffled batches of 50, Adam lr 1e-3 wd 5e-4)
    against the oracle driven with the same batches."""
    data, meta = _dataset(n=250)
    paths, obs, nb_obs = data
    model, _ = train.train(data, meta, epochs=1, batch_size=50, dropout_rate=0.0, fused=True,
                           log=lambda s: None)
    # replay on the oracle
    torch.manual_seed(0)
    ref = models.NJODE(1, 10, 1, train.train.__defaults__[6], train.train.__defaults__[7],
                       train.train.__defaults__[8], False, options={})
    sd = {k: v.detach().clone() for k, v in ref.state_dict().items()}
    o = njode_oracle.make_oracle(dict(input_size=1, hidden_size=10, output_size=1,
                                      ode_nn=train.train.__defaults__[6],
                                      readout_nn=train.train.__defaults__[7],
                                      enc_nn=train.train.__defaults__[8], dropout_rate=0.0))
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    opt = torch.optim.Adam(list(params.values()), lr=1e-3, weight_decay=0.0005)
    from njode_amd import parallel
    train_idx, _ = train.split_indices(250, 0.2, 398)
    order = train_idx[parallel.epoch_permutation(len(train_idx), 1, 0)]
    for s in range(4):
        idx = order[s * 50:(s + 1) * 50]
        b = data_utils.collate_arrays(paths[idx], obs[idx], nb_obs[idx], meta['dt'])
        njode_oracle.train_step(o, params, opt, b, meta['dt'], meta['maturity'])
    flat_ref = torch.cat([params[k].detach().reshape(-1) for k in sd])
    np.testing.assert_allclose(model.flat_parameters().cpu().numpy(), flat_ref.numpy(),
                               atol=5e-5, rtol=1e-3)


def test_device_collated_loop_is_bit_identical():
    """The GPU batch producer's collate feeds the same batches as the host collate, so the
    whole training trajectory (dropout on) is identical."""
    data, meta = _dataset(n=400)
    m1, met1 = train.train(data, meta, epochs=2, batch_size=64, dropout_rate=0.1,
                           log=lambda s: None)
    m2, met2 = train.train(data, meta, epochs=2, batch_size=64, dropout_rate=0.1,
                           device_collate=True, log=lambda s: None)
    assert torch.equal(m1.flat_parameters(), m2.flat_parameters())
    assert met1[1][3] == met2[1][3] and met1[1][4] == met2[1][4]


def test_checkpoint_and_metric_file_policy(tmp_path):
    """train.py:400-418, 585-621: metric csv with the reference's columns, last / best
    checkpoints, and resuming continues from the saved epoch with the saved optimizer."""
    import csv
    import os
    data, meta = _dataset(n=300)
    kw = dict(batch_size=60, dropout_rate=0.0, log=lambda s: None, model_path=str(tmp_path),
              model_id=7)
    m1, met1 = train.train(data, meta, epochs=2, **kw)
    d = os.path.join(str(tmp_path), 'id-7')
    assert os.path.exists(os.path.join(d, 'last_checkpoint', 'checkpt.tar'))
    assert os.path.exists(os.path.join(d, 'best_checkpoint', 'checkpt.tar'))
    rows = list(csv.reader(open(os.path.join(d, 'metric_id-7.csv'))))
    assert rows[0] == [''] + train.METR_COLUMNS and len(rows) == 3
    assert float(rows[2][5]) == pytest.approx(met1[1][4])
    # three epochs in one go == two epochs, then resume for the third
    m2, met2 = train.train(data, meta, epochs=3, resume_training=True, **kw)
    m3, met3 = train.train(data, meta, epochs=3, batch_size=60, dropout_rate=0.0,
                           log=lambda s: None)
    assert len(met2) == 1 and met2[0][0] == 3
    np.testing.assert_allclose(m2.flat_parameters().cpu().numpy(),
                               m3.flat_parameters().cpu().numpy(), rtol=1e-6, atol=1e-7)
    ck = torch.load(os.path.join(d, 'last_checkpoint', 'checkpt.tar'), weights_only=False)
    assert set(ck) == {'epoch', 'weight', 'model_state_dict', 'optimizer_state_dict'}
