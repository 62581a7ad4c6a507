"""PhysioNet evaluation protocol (reference physionet_train.py:411-510) against a
golden produced by the reference's own evaluate_model / get_comparison_times_ind
(tests/golden/make_golden.py g8): first on the CPU oracle (pins the protocol code), then
(-m gpu) on the HIP model."""
import numpy as np
import pytest
import torch

from golden_util import Golden
from njode_amd import physionet_eval
from oracle import njode_oracle


def _batches(g):
    out = []
    for i in range(int(g['n_batches'])):
        p = 'b{}/'.format(i)
        out.append({'times': g[p + 'times'], 'time_ptr': g[p + 'time_ptr'],
                    'X': torch.tensor(g[p + 'X']), 'M': torch.tensor(g[p + 'M']),
                    'obs_idx': torch.tensor(g[p + 'obs_idx'], dtype=torch.long),
                    'batch_size': int(g[p + 'batch_size']), 'times_val': g[p + 'times_val'],
                    'vals_val': g[p + 'vals_val'], 'mask_val': g[p + 'mask_val']})
    return out


class _OracleModel:
    """The oracle behind the model call signature the protocol uses."""

    def __init__(self, g):
        self.o = njode_oracle.make_oracle(g.cfg)
        self.params = g.state_dict()

    def eval(self):
        self.o.training = False

    def __call__(self, times, time_ptr, X, obs_idx, delta_t, T, start_X, n_obs_ot, **kw):
        return self.o.forward(self.params, times, time_ptr, X, obs_idx, delta_t, T, start_X,
                              n_obs_ot, **kw)


def test_synthetic_test_layout_is_reproducible():
    g = Golden('g8_physionet_eval')
    for i, seed in enumerate((3, 4)):
        b = physionet_eval.make_eval_batch(batch_size=6, n_grid=240, n_obs_range=(6, 16), seed=seed)
        p = 'b{}/'.format(i)
        assert np.array_equal(b['times'], g[p + 'times'])
        assert np.array_equal(b['time_ptr'], g[p + 'time_ptr'])
        assert np.array_equal(b['X'].numpy(), g[p + 'X'])
        assert np.array_equal(b['vals_val'], g[p + 'vals_val'])
        assert np.array_equal(b['mask_val'], g[p + 'mask_val'])
        # observe the first half of the union times, hold out the second half
        n_all = len(b['times']) + len(b['times_val'])
        assert len(b['times']) == n_all // 2 and b['times'][-1] < b['times_val'][0]


def test_comparison_indices_match_reference():
    g = Golden('g8_physionet_eval')
    for i in range(int(g['n_batches'])):
        p = 'b{}/'.format(i)
        got = physionet_eval.get_comparison_times_ind(g[p + 'path_t'], g[p + 'times_val'])
        assert np.array_equal(got, g[p + 'cmp_ind'])
    # nearest neighbour with ties to the left, and the clamp past the last interval
    t = np.array([0.0, 0.1, 0.1, 0.2, 0.4])
    assert physionet_eval.get_comparison_times_ind(t, [0.1, 0.15, 0.16, 0.31, 0.4]).tolist() == \
        [1, 2, 3, 4, 4]
    with pytest.raises(AssertionError):
        physionet_eval.get_comparison_times_ind(t, [0.0, 0.5])


def test_protocol_on_oracle_matches_reference():
    g = Golden('g8_physionet_eval')
    loss_val, mse_val, mse_val_2 = physionet_eval.evaluate_model(
        _OracleModel(g), _batches(g), 'cpu', g.delta_t, g.T)
    assert loss_val == pytest.approx(float(g['loss_val']), rel=1e-6)
    assert mse_val == pytest.approx(float(g['mse_val']), rel=1e-6)
    assert mse_val_2 == pytest.approx(float(g['mse_val_2']), rel=1e-5)


@pytest.mark.gpu
def test_protocol_on_hip_model_matches_reference():
    from hip_util import LOSS_RTOL, hip_model
    g = Golden('g8_physionet_eval')
    m = hip_model(g.cfg, g.state_dict())
    loss_val, mse_val, mse_val_2 = physionet_eval.evaluate_model(
        m, _batches(g), 'cuda', g.delta_t, g.T)
    assert loss_val == pytest.approx(float(g['loss_val']), rel=LOSS_RTOL)
    assert mse_val == pytest.approx(float(g['mse_val']), rel=1e-4)
    assert mse_val_2 == pytest.approx(float(g['mse_val_2']), rel=1e-4)
