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


HT_CASES = ['g16_hT_demo', 'g16_hT_rnn', 'g16_hT_w100', 'g16_hT_masked']


@pytest.mark.parametrize('use_op', [False, True])
@pytest.mark.parametrize('name', HT_CASES)
def test_gradient_through_hT_matches_reference(name, use_op):
    """The reference returns hT inside its autograd graph (models.py:414-518); round 4's custom-op
    route received grad_hT and dropped it.  Now a backward pass that reaches hT runs a second pass
    on the lockstep plan seeded with that gradient (models._hT_and_loss_grads): the gradients of
    loss + <W, hT> and of <W, hT> alone against the reference's autograd (make_golden.py:g16), on
    the segment plan (demo shape), the GRU jump, the shape-generic kernels and a masked model."""
    import numpy as np
    from golden_util import Golden
    from hip_util import GRAD_REL_L2, grads_by_name, hip_forward, hip_model, rel_l2
    g = Golden(name)
    cfg = dict(g.cfg)
    cfg['options'] = dict(cfg.get('options', {}), torch_library_op=use_op)
    W = torch.tensor(g['W']).cuda()
    for tag, with_loss in (('both', True), ('hT', False)):
        m = hip_model(cfg, g.state_dict()).train()
        hT, loss = hip_forward(m, g.batch(), g.delta_t, g.T)
        assert hT.requires_grad
        np.testing.assert_allclose(hT.detach().cpu().numpy(), g['train_hT'], atol=3e-5, rtol=1e-4)
        obj = (hT * W).sum() + (loss if with_loss else 0.0)
        assert float(obj.detach()) == pytest.approx(float(g[tag + '/objective']), rel=1e-4, abs=1e-4)
        obj.backward()
        got = grads_by_name(m)
        for k, ref in g.group(tag + '/grad').items():
            if not np.any(ref):                      # (hT does not depend on this tensor)
                assert got[k] is None or not np.any(got[k]) or np.abs(got[k]).max() < 1e-30, k
            else:
                assert rel_l2(got[k], ref) < GRAD_REL_L2, (tag, k, rel_l2(got[k], ref))


def test_hT_detached_costs_no_second_pass():
    """The loss alone (and hT.detach()) still takes the single backward."""
    cfg = demo_cfg()
    cfg['options'] = dict(cfg.get('options', {}), device_outputs=True)
    torch.manual_seed(0)
    m = models.NJODE(**cfg).cuda().train()
    b, meta = bs_batch(16, seed=3)
    b = to_dev(b)
    calls = []
    orig = m._grad_through_hT
    m._grad_through_hT = lambda *a, **k: calls.append(1) or orig(*a, **k)
    hT, loss = m(b['times'], b['time_ptr'], b['X'], b['obs_idx'], meta['dt'], meta['maturity'],
                 b['start_X'], b['n_obs_ot'])
    (2.0 * loss + hT.detach().sum()).backward()
    assert not calls
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())


@pytest.mark.parametrize('masked', [False, True])
def test_hT_pass_replays_the_same_dropout_masks(masked):
    """The second pass of a gradient through hT re-runs the step on the lockstep plan with the
    call's dropout seed: with dropout ON its hT must be the call's own hT (the segment plan's tails
    resp. the masked lockstep forward) -- masks are keyed by (seed, path, Euler step / jump time,
    network), not by the plan -- and the gradient must be finite and repeatable."""
    import numpy as np
    from hip_util import hip_model
    from njode_amd import synthetic_physionet
    if masked:
        nn = ((50, 'tanh'), (50, 'tanh'))
        cfg = dict(input_size=41, hidden_size=41, output_size=41, ode_nn=nn, readout_nn=nn, enc_nn=nn,
                   use_rnn=False, bias=True, dropout_rate=0.1, options={'masked': True})
        b = synthetic_physionet.make_batch(batch_size=11, n_grid=80, n_obs_range=(4, 10), seed=6)
        dt, T = b['delta_t'], b['T']
    else:
        cfg = demo_cfg(dropout=0.1)
        b, meta = bs_batch(48, seed=8)
        dt, T = meta['dt'], meta['maturity']
    torch.manual_seed(0)
    m = hip_model(cfg).train()
    d = to_dev(b)
    grads = []
    for _ in range(2):
        m._step_counter = 5
        m.zero_grad()
        hT, loss = m(d['times'], d['time_ptr'], d['X'], d['obs_idx'], dt, T, d['start_X'], d['n_obs_ot'],
                     M=d.get('M'))
        (loss + 0.37 * hT.sum()).backward()
        np.testing.assert_allclose(m._last_hT_replay.cpu().numpy(), hT.detach().cpu().numpy(),
                                   atol=2e-5, rtol=1e-4)
        g = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
        assert torch.isfinite(g).all()
        grads.append(g.clone())
    assert float((grads[0] - grads[1]).norm() / grads[0].norm()) < 1e-6


@pytest.mark.parametrize('name', HT_CASES)
def test_hT_of_a_get_loss_false_call_is_differentiable(name):
    """models.py:414-518: hT is in the graph whether the loss was asked for or not.  A
    get_loss=False call saves nothing; a backward that reaches its hT replays the step (default
    route; the custom-op route raises for such a call)."""
    import numpy as np
    from golden_util import Golden
    from hip_util import GRAD_REL_L2, grads_by_name, hip_forward, hip_model, rel_l2
    g = Golden(name)
    W = torch.tensor(g['W']).cuda()
    m = hip_model(g.cfg, g.state_dict()).train()
    hT, loss = hip_forward(m, g.batch(), g.delta_t, g.T, get_loss=False)
    assert loss == 0 and hT.requires_grad
    (hT * W).sum().backward()
    got = grads_by_name(m)
    for k, ref in g.group('hT/grad').items():
        if np.any(ref):
            assert rel_l2(got[k], ref) < GRAD_REL_L2, (k, rel_l2(got[k], ref))
        else:
            assert not np.any(got[k]) or np.abs(got[k]).max() < 1e-30, k
    with torch.no_grad():                      # (and without autograd: a plain tensor, as before)
        hT2, _ = hip_forward(m, g.batch(), g.delta_t, g.T, get_loss=False)
    assert not hT2.requires_grad


def test_gradient_through_hT_matches_finite_differences_with_dropout_on():
    """loss + <W, hT> with dropout 0.1 (masks fixed by the step counter): the gradient of the two
    passes -- the call's own backward + the replay seeded with W -- against central finite
    differences of the objective along random directions (segment plan, demo shape)."""
    from hip_util import hip_model
    torch.manual_seed(0)
    m = hip_model(demo_cfg(dropout=0.1)).train()
    b, meta = bs_batch(40, seed=9)
    d = to_dev(b)
    args = (d['times'], d['time_ptr'], d['X'], d['obs_idx'], meta['dt'], meta['maturity'], d['start_X'],
            d['n_obs_ot'])
    W = torch.randn(40, 10, generator=torch.Generator().manual_seed(3)).cuda()

    def objective():
        m._step_counter = 5
        hT, loss = m(*args)
        return loss + (hT * W).sum()

    m.zero_grad()
    objective().backward()
    g = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone()
    assert torch.isfinite(g).all() and float(g.abs().max()) > 0
    flat = m.flat_parameters()
    base = flat.clone()
    gen = torch.Generator(device='cpu').manual_seed(1)
    eps = 1e-2
    for _ in range(3):
        v = torch.randn(flat.shape, generator=gen).to(flat.device)
        v /= v.norm()
        vals = []
        for s in (+1, -1):
            flat.copy_(base + s * eps * v)
            with torch.no_grad():
                vals.append(float(objective().double()))
        flat.copy_(base)
        fd = (vals[0] - vals[1]) / (2 * eps)
        an = float((g * v).sum())
        assert fd == pytest.approx(an, rel=4e-2, abs=1e-3), (fd, an)
