"""Climate evaluation protocol (reference climate_train.py:508-566 +
GRU_ODE_Bayes/data_utils_gru_ode_bayes.py:379-408) against a golden produced by the reference's own
evaluate_model / extract_from_path (tests/golden/make_golden.py g11): first on the CPU oracle
(pins the protocol code), then (-m gpu) on the HIP model -- the climate shape (d = 5, H = 10,
masked) runs on the shape-generic kernels."""
import numpy as np
import pytest
import torch

from golden_util import Golden
from njode_amd import climate_eval
from oracle import njode_oracle


def _batches(g):
    out = []
    for i in range(int(g['n_batches'])):
        p = 'b{}/'.format(i)
        out.append({'times': g[p + 'times'], 'time_ptr': g[p + 'time_ptr'],
                    'X': torch.tensor(g[p + 'X']), 'M': torch.tensor(g[p + 'M']),
                    'obs_idx': torch.tensor(g[p + 'obs_idx'], dtype=torch.long),
                    'pat_idx': list(range(int(g[p + 'batch_size']))),
                    'X_val': torch.tensor(g[p + 'X_val']), 'M_val': torch.tensor(g[p + 'M_val']),
                    'times_val': g[p + 'times_val'], 'index_val': g[p + 'index_val']})
    return out


class _OracleModel:
    def __init__(self, g):
        self.o = njode_oracle.make_oracle(g.cfg)
        self.params = g.state_dict()

    def eval(self):
        self.o.training = False

    def __call__(self, times, time_ptr, X, obs_idx, delta_t, T, start_X, n_obs_ot, **kw):
        return self.o.forward(self.params, times, time_ptr, X, obs_idx, delta_t, T, start_X,
                              n_obs_ot, **kw)


def test_synthetic_climate_layout_is_reproducible():
    g = Golden('g8_climate_eval')
    for i, seed in enumerate((1, 2)):
        b = climate_eval.make_climate_batch(batch_size=7, T=20, T_val=15, n_obs_range=(5, 12), seed=seed)
        p = 'b{}/'.format(i)
        for k in ('times', 'time_ptr', 'times_val', 'index_val'):
            assert np.array_equal(np.asarray(b[k]), g[p + k]), k
        for k in ('X', 'M', 'obs_idx', 'X_val', 'M_val'):
            assert np.array_equal(b[k].numpy(), g[p + k]), k
        # observed part up to T_val, held-out rows after it, sorted by (station, time)
        assert b['times'].max() <= 15 and b['times_val'].min() > 15
        order = np.lexsort((b['times_val'], b['index_val']))
        assert np.array_equal(order, np.arange(len(order)))
        assert np.bincount(b['index_val'], minlength=7).max() <= 3


def test_extract_from_path_matches_reference():
    g = Golden('g8_climate_eval')
    for i in range(int(g['n_batches'])):
        p = 'b{}/'.format(i)
        # any array with the path's shape does: the function only selects
        path_t = g[p + 'path_t']
        rng = np.random.RandomState(i)
        path_y = rng.standard_normal((len(path_t), int(g[p + 'batch_size']), 5)).astype(np.float32)
        t_vec = np.around(path_t, 1).astype(np.float32)
        got = climate_eval.extract_from_path(t_vec, path_y, g[p + 'times_val'], g[p + 'index_val'])
        tu, first = np.unique(t_vec, return_index=True)
        for j, (t, b) in enumerate(zip(g[p + 'times_val'], g[p + 'index_val'])):
            k = first[np.abs(tu.astype(np.float64) - t).argmin()]
            assert np.array_equal(got[j], path_y[k, b])
    # a time between two path times goes to the closer one, a tie to the earlier one; of a
    # repeated time (jump) the first row is taken
    t = np.array([0.0, 0.125, 0.125, 0.25, 0.5], dtype=np.float32)
    y = np.arange(5, dtype=np.float32).reshape(5, 1, 1)
    out = climate_eval.extract_from_path(t, y, np.array([0.125, 0.2, 0.375, 0.38, 0.9]),
                                         np.zeros(5, dtype=int))
    assert out.reshape(-1).tolist() == [1.0, 3.0, 3.0, 4.0, 4.0]
    assert climate_eval.n_decimals(0.1) == 1 and climate_eval.n_decimals(0.05) == 2


def test_protocol_on_oracle_matches_reference():
    g = Golden('g8_climate_eval')
    batches = _batches(g)
    loss_val, mse_val = climate_eval.evaluate_model(_OracleModel(g), batches, 'cpu', g.delta_t, g.T)
    assert loss_val == pytest.approx(float(g['loss_val']), rel=1e-6)
    assert mse_val == pytest.approx(float(g['mse_val']), rel=1e-6)
    # and the per-batch selections the reference made
    m = _OracleModel(g)
    m.eval()
    for i, b in enumerate(batches):
        n_obs_ot = torch.tensor(np.bincount(b['obs_idx'].numpy(), minlength=len(b['pat_idx'])))
        with torch.no_grad():
            _, _, path_t, _, path_y = m(b['times'], b['time_ptr'], b['X'], b['obs_idx'], g.delta_t, g.T,
                                        torch.zeros(len(b['pat_idx']), 5), n_obs_ot, until_T=True,
                                        return_path=True, get_loss=True, M=b['M'])
        assert np.array_equal(np.asarray(path_t), g['b{}/path_t'.format(i)])
        got = climate_eval.extract_from_path(np.around(path_t, 1).astype(np.float32), path_y.numpy(),
                                             b['times_val'], b['index_val'])
        np.testing.assert_allclose(got, g['b{}/p_val'.format(i)], atol=1e-6, rtol=0)


@pytest.mark.gpu
def test_protocol_on_hip_model_matches_reference():
    from hip_util import LOSS_RTOL, hip_model
    g = Golden('g8_climate_eval')
    m = hip_model(g.cfg, g.state_dict())
    loss_val, mse_val = climate_eval.evaluate_model(m, _batches(g), 'cuda', g.delta_t, g.T)
    assert loss_val == pytest.approx(float(g['loss_val']), rel=LOSS_RTOL)
    assert mse_val == pytest.approx(float(g['mse_val']), rel=1e-4)
