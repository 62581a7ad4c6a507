"""torch.library route (njode_amd/ops.py): torch.ops.njode_amd.forward is registered with a
fake implementation and an autograd formula; it returns the numbers of the default
torch.autograd.Function route (same two library calls)."""
import pytest
import torch

from hip_util import bs_batch, demo_cfg, to_dev
from njode_amd import models

pytestmark = pytest.mark.gpu


def _run(use_op, train):
    cfg = demo_cfg()
    cfg['options'] = dict(cfg.get('options', {}), torch_library_op=use_op, device_outputs=True)
    torch.manual_seed(0)
    m = models.NJODE(**cfg).cuda()
    m.train(train)
    b, meta = bs_batch(64, seed=1)
    b = to_dev(b)
    hT, loss = m(b['times'], b['time_ptr'], b['X'], b['obs_idx'], meta['dt'], meta['maturity'],
                 b['start_X'], b['n_obs_ot'])
    g = None
    if train:
        loss.backward()
        g = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone()
    return hT.clone(), float(loss), g


def test_operator_is_registered():
    import njode_amd.ops  # noqa: F401
    assert hasattr(torch.ops.njode_amd, 'forward')
    schema = str(torch.ops.njode_amd.forward.default._schema)
    assert 'Tensor[] params' in schema and 'model_id' in schema


@pytest.mark.parametrize('train', [False, True])
def test_operator_route_matches_autograd_function_route(train):
    h0, l0, g0 = _run(False, train)
    h1, l1, g1 = _run(True, train)
    assert l1 == pytest.approx(l0, rel=1e-6)
    assert torch.allclose(h1, h0, rtol=1e-6, atol=1e-7)
    if train:
        assert float((g1 - g0).norm() / g0.norm()) < 1e-6


def test_second_backward_through_the_operator_raises():
    cfg = demo_cfg()
    cfg['options'] = dict(cfg.get('options', {}), torch_library_op=True, device_outputs=True)
    m = models.NJODE(**cfg).cuda().train()
    b, meta = bs_batch(16, seed=2)
    b = to_dev(b)
    _, loss = m(b['times'], b['time_ptr'], b['X'], b['obs_idx'], meta['dt'], meta['maturity'],
                b['start_X'], b['n_obs_ot'])
    loss.backward(retain_graph=True)
    with pytest.raises(RuntimeError):
        loss.backward()


@pytest.mark.parametrize('use_op', [False, True])
def test_gradient_through_hT_is_refused_not_dropped(use_op):
    """The reference returns hT inside its autograd graph (models.py:414-518); the library
    differentiates the loss only.  A loss that touches hT must raise on both routes -- round 4's
    custom-op route received grad_hT and ignored it -- while the loss alone still trains."""
    cfg = demo_cfg()
    cfg['options'] = dict(cfg.get('options', {}), torch_library_op=use_op, device_outputs=True)
    torch.manual_seed(0)
    m = models.NJODE(**cfg).cuda().train()
    b, meta = bs_batch(16, seed=3)
    b = to_dev(b)
    args = (b['times'], b['time_ptr'], b['X'], b['obs_idx'], meta['dt'], meta['maturity'],
            b['start_X'], b['n_obs_ot'])
    hT, loss = m(*args)
    assert hT.requires_grad                       # part of the graph, as in the reference
    with pytest.raises(NotImplementedError, match='hT'):
        (loss + hT.sum()).backward()
    hT, loss = m(*args)
    with pytest.raises(NotImplementedError, match='hT'):
        hT.sum().backward()
    hT, loss = m(*args)
    (2.0 * loss + hT.detach().sum()).backward()   # detached: fine
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
