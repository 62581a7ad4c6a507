"""plot_one_path_with_pred (reference train.py:673-796) on the CPU: a stand-in model supplies
get_pred (the real one needs the GPU); the data, the collate and the true conditional
expectation are the build's own."""
import os

import numpy as np
import torch

from njode_amd import data_utils, plotting, stock_model


class _StubModel:
    weight = 0.5

    def eval(self):
        return self

    def get_pred(self, times, time_ptr, X, obs_idx, delta_t, T, start_X):
        n = int(round(T / delta_t)) + 1 + len(times)
        t = np.sort(np.concatenate([np.linspace(0., T, n - len(times)), np.asarray(times)]))
        B, d = start_X.shape[0], 2 * X.shape[1]      # as if trained on func_appl_X=['power-2']
        pred = torch.ones(len(t), B, d) * start_X.reshape(1, B, -1)[:, :, :1].repeat(1, 1, d)
        pred[:, :, X.shape[1]:] += 0.04               # E[X^2] - E[X]^2 = 0.04
        return {'pred': pred, 'pred_t': t}


def test_plot_is_written_and_optimal_loss_returned(tmp_path):
    hp = dict(data_utils.hyperparam_default, nb_paths=6)
    paths, obs, nb_obs, meta = data_utils.create_dataset('BlackScholes', hp, seed=3)
    b = data_utils.collate_arrays(paths, obs, nb_obs, meta['dt'])
    b['true_paths'], b['observed_dates'] = paths, obs
    sm = stock_model.STOCK_MODELS['BlackScholes'](**{k: meta[k] for k in
                                                    ('drift', 'volatility', 'nb_paths', 'nb_steps',
                                                     'S0', 'maturity', 'dimension')})
    opt = plotting.plot_one_path_with_pred(
        torch.device('cpu'), _StubModel(), b, sm, meta['dt'], meta['maturity'], path_to_plot=(0, 3),
        save_path=str(tmp_path), filename='p_{}.png', plot_variance=True, functions=['power-2'],
        ylabels=['x'])
    assert np.isfinite(opt) and opt > 0
    for i in (0, 3):
        f = tmp_path / 'p_{}.png'.format(i)
        assert f.exists() and os.path.getsize(f) > 1000
