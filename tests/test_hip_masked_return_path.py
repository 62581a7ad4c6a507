"""Masked training calls that also ask for the prediction path (return_path=True) take the
one-wave lockstep forward, which does not store the hidden activations the four-wave adjoint
sweep reads (njode_mfma_lock4.h): the backward must then fall back to the recomputing sweep.
Loss and gradients have to agree with the plain training call."""
import pytest
import torch

from njode_amd import models, synthetic_physionet

pytestmark = pytest.mark.gpu
NN = ((50, 'tanh'), (50, 'tanh'))


def _grads(return_path):
    cfg = dict(input_size=41, hidden_size=41, output_size=41, ode_nn=NN, readout_nn=NN, enc_nn=NN,
               use_rnn=False, bias=True, dropout_rate=0.0, options={'masked': True})
    b = synthetic_physionet.make_batch(batch_size=20, dim=41, n_grid=120, n_obs_range=(3, 9), seed=3)
    torch.manual_seed(1)
    m = models.NJODE(**cfg).cuda().train()
    out = m(b['times'], b['time_ptr'], b['X'].cuda(), b['obs_idx'], b['delta_t'], b['T'],
            b['start_X'].cuda(), b['n_obs_ot'].cuda(), return_path=return_path, get_loss=True,
            M=b['M'].cuda())
    loss = out[1]
    loss.backward()
    return float(loss), torch.cat([q.grad.reshape(-1) for q in m.parameters()]).cpu().clone()


def test_masked_gradients_do_not_depend_on_return_path():
    l0, g0 = _grads(False)
    l1, g1 = _grads(True)
    assert l1 == pytest.approx(l0, rel=1e-5)
    assert float((g1 - g0).norm() / g0.norm()) < 1e-4
