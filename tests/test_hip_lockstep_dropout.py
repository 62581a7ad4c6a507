"""GPU tests of the lockstep plan's training mode: the backward regenerates the dropout masks
its saving forward drew (matrix-core keying or VALU keying, depending on which sweep the
shape has), so with the masks fixed the analytic gradient must match central finite
differences of the loss -- for the masked PhysioNet-shaped model (self-imputation feeds the
prediction back) and for an unmasked model on a schedule with a tail."""
import numpy as np
import pytest
import torch

from hip_util import (GRAD_REL_L2, LOSS_RTOL, bs_batch, demo_cfg, hip_model, oracle_forward,
                      rel_l2, to_dev)
from njode_amd import models, synthetic_physionet

pytestmark = pytest.mark.gpu
NN = ((50, 'tanh'), (50, 'tanh'))


def _fd_check(m, args, kw, eps, n_dir=2, rel=4e-2, autograd=False):
    m._step_counter = 5
    if autograd:
        m.flat_grad().zero_()
        _, loss = m(*args, **kw)
        loss.backward()
    else:
        _, loss = m.loss_and_grad(*args, **kw)
    g = m.flat_grad().clone()
    assert torch.isfinite(g).all() and float(g.abs().max()) > 0
    flat = m.flat_parameters()
    base = flat.clone()
    gen = torch.Generator(device='cpu').manual_seed(1)
    for _ in range(n_dir):
        v = torch.randn(flat.shape, generator=gen).to(flat.device)
        v /= v.norm()
        vals = []
        for s in (+1, -1):
            flat.copy_(base + s * eps * v)
            m._step_counter = 5
            with torch.no_grad():
                vals.append(float(m(*args, **kw)[1].double()))
        flat.copy_(base)
        fd = (vals[0] - vals[1]) / (2 * eps)
        an = float((g * v).sum())
        assert fd == pytest.approx(an, rel=rel, abs=1e-4), (fd, an)
    return float(loss)


@pytest.mark.parametrize('hidden,residual', [(41, True), (50, False)])
def test_masked_dropout_gradient_matches_finite_differences(hidden, residual):
    cfg = dict(input_size=41, hidden_size=hidden, output_size=41, ode_nn=NN, readout_nn=NN,
               enc_nn=NN, use_rnn=False, bias=True, dropout_rate=0.1,
               options={'masked': True, 'device_outputs': True, 'residual_enc_dec': residual})
    b = synthetic_physionet.make_batch(batch_size=37, n_grid=60, n_obs_range=(3, 9), seed=3)
    torch.manual_seed(0)
    m = models.NJODE(**cfg).cuda().train()
    args = (b['times'], b['time_ptr'], b['X'].cuda(), b['obs_idx'].cuda().int(), b['delta_t'],
            b['T'], b['start_X'].cuda(), b['n_obs_ot'].cuda().int())
    _fd_check(m, args, {'M': b['M'].cuda()}, eps=1e-2)


@pytest.mark.parametrize('hidden,residual', [(41, True), (50, False)])
def test_masked_gradients_match_oracle(hidden, residual):
    """Both PhysioNet shapes (the reference's H = 41 residual one and BASELINE's H = 50
    non-residual wording) against the oracle's autograd gradients, dropout off."""
    cfg = dict(input_size=41, hidden_size=hidden, output_size=41, ode_nn=NN, readout_nn=NN,
               enc_nn=NN, use_rnn=False, bias=True, dropout_rate=0.0,
               options={'masked': True, 'residual_enc_dec': residual})
    b = synthetic_physionet.make_batch(batch_size=21, n_grid=50, n_obs_range=(3, 8), seed=5)
    torch.manual_seed(1)
    m = hip_model(cfg).train()
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    (_, l_o), params = oracle_forward(cfg, sd, b, b['delta_t'], b['T'], training=True, grads=True)
    l_o.backward()
    args = (b['times'], b['time_ptr'], b['X'].cuda(), b['obs_idx'].cuda().int(), b['delta_t'],
            b['T'], b['start_X'].cuda(), b['n_obs_ot'].cuda().int())
    _, loss = m.loss_and_grad(*args, M=b['M'].cuda())
    assert float(loss) == pytest.approx(float(l_o), rel=LOSS_RTOL)
    ref = np.concatenate([params[k].grad.numpy().reshape(-1) for k in sd])
    assert rel_l2(m.flat_grad().cpu().numpy(), ref) < GRAD_REL_L2


def test_masked_dropout_is_reproducible_and_seeded():
    cfg = dict(input_size=41, hidden_size=41, output_size=41, ode_nn=NN, readout_nn=NN,
               enc_nn=NN, use_rnn=False, bias=True, dropout_rate=0.1,
               options={'masked': True, 'device_outputs': True})
    b = synthetic_physionet.make_batch(batch_size=20, n_grid=40, n_obs_range=(3, 6), seed=4)
    torch.manual_seed(0)
    m = models.NJODE(**cfg).cuda().train()
    args = (b['times'], b['time_ptr'], b['X'].cuda(), b['obs_idx'].cuda().int(), b['delta_t'],
            b['T'], b['start_X'].cuda(), b['n_obs_ot'].cuda().int())
    kw = {'M': b['M'].cuda()}
    out = []
    for step in (7, 7, 8):
        m._step_counter = step
        _, loss = m.loss_and_grad(*args, **kw)
        out.append((float(loss), m.flat_grad().clone()))
    assert out[0][0] == out[1][0] and torch.equal(out[0][1], out[1][1])
    assert out[0][0] != out[2][0]


def test_unmasked_lockstep_dropout_gradient_matches_finite_differences():
    cfg = demo_cfg(dropout=0.1)
    b, meta = bs_batch(300, seed=8)
    torch.manual_seed(0)
    m = hip_model(cfg).train()
    d = to_dev(b)
    args = (d['times'], d['time_ptr'], d['X'], d['obs_idx'], meta['dt'], meta['maturity'] + 0.05,
            d['start_X'], d['n_obs_ot'])
    _fd_check(m, args, {'until_T': True}, eps=2e-2, autograd=True)
