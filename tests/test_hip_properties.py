"""GPU tests of size-independent properties at BASELINE's full size (20 000
Black-Scholes paths) and of the training-mode (dropout) path."""
import numpy as np
import pytest
import torch

from hip_util import (GRAD_REL_L2, LOSS_RTOL, bs_batch, demo_cfg, hip_forward, hip_model,
                      oracle_forward, rel_l2, to_dev)
from njode_amd import data_utils, models

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def big():
    b, meta = bs_batch(20000, seed=0)
    torch.manual_seed(0)
    m = hip_model(demo_cfg()).eval()
    return b, meta, m


def _sub_batch(b, meta, idx):
    return data_utils.collate_arrays(b['true_paths'][idx], b['observed_dates'][idx],
                                     b['observed_dates'][idx][:, 1:].sum(1), meta['dt'])


def test_two_plans_agree_at_full_size(big):
    """segment plan and lockstep plan are independent kernels: same loss and hT."""
    b, meta, m = big
    with torch.no_grad():
        hT_s, loss_s = hip_forward(m, b, meta['dt'], meta['maturity'])
        hT_l, loss_l, _, _, _ = hip_forward(m, b, meta['dt'], meta['maturity'],
                                            return_path=True, get_loss=True)
    assert float(loss_s) == pytest.approx(float(loss_l), rel=2e-5)
    np.testing.assert_allclose(hT_s.cpu().numpy(), hT_l.cpu().numpy(), atol=1e-5, rtol=1e-4)


def test_autograd_route_equals_the_fused_step_at_full_size(big):
    """The literal reference sequence -- hT, loss = model(...); loss.backward() -- against
    loss_and_grad at 20 000 paths, where the call has helper streams: the forward runs the
    backward's row pass itself (NJODE_C_ROWS_IN_FWD), the tail order and the tails (hT) run on a
    stream of their own beside the ODE forward.  Same loss, same gradient, and the hT of the
    lockstep plan."""
    b, meta, m = big
    m.train()                     # dropout_rate = 0 here
    try:
        args = _args(b, meta)
        _, loss_f = m.loss_and_grad(*args)
        g_f = m.flat_grad().clone()
        m.zero_grad()
        for _ in range(2):        # (twice: helper-stream events are reused from call to call)
            hT, loss = m(*args, return_path=False, get_loss=True)
            loss.backward()
        g_a = torch.cat([p.grad.reshape(-1) for p in m.parameters()]) / 2.0
        with torch.no_grad():
            hT_l = hip_forward(m, b, meta['dt'], meta['maturity'], return_path=True, get_loss=True)[0]
    finally:
        m.eval()
    assert float(loss) == pytest.approx(float(loss_f), rel=1e-6)
    assert rel_l2(g_a.cpu().numpy(), g_f.cpu().numpy()) < 1e-6
    np.testing.assert_allclose(hT.detach().cpu().numpy(), hT_l.cpu().numpy(), atol=1e-5, rtol=1e-4)


def test_permutation_invariance_at_full_size(big):
    b, meta, m = big
    perm = np.random.RandomState(0).permutation(20000)
    bp = _sub_batch(b, meta, perm)
    with torch.no_grad():
        hT, loss = hip_forward(m, b, meta['dt'], meta['maturity'])
        hTp, lossp = hip_forward(m, bp, meta['dt'], meta['maturity'])
    assert float(lossp) == pytest.approx(float(loss), rel=2e-5)
    np.testing.assert_allclose(hTp.cpu().numpy(), hT.cpu().numpy()[perm], atol=1e-6, rtol=1e-5)


def test_shards_add_up_to_the_full_batch(big):
    """The data-parallel contract: with the global batch size in the loss
    denominator, per-shard losses and gradients SUM to the single-GPU values
    (what the RCCL all-reduce computes)."""
    b, meta, m = big
    m.train()                     # dropout_rate = 0 here
    try:
        _, loss = m.loss_and_grad(*_args(b, meta))
        g_full = m.flat_grad().clone()
        total, g_sum = 0.0, torch.zeros_like(g_full)
        n_shards = 4
        for r in range(n_shards):
            idx = np.arange(r * 5000, (r + 1) * 5000)
            bs = _sub_batch(b, meta, idx)
            m.dp_global_batch, m.dp_path_offset = 20000, r * 5000
            _, l = m.loss_and_grad(*_args(bs, meta))
            total += float(l)
            g_sum += m.flat_grad()
        assert total == pytest.approx(float(loss), rel=2e-5)
        assert rel_l2(g_sum.cpu().numpy(), g_full.cpu().numpy()) < 1e-4
    finally:
        m.dp_global_batch, m.dp_path_offset = None, 0
        m.eval()


def _args(b, meta):
    d = to_dev(b)
    return (d['times'], d['time_ptr'], d['X'], d['obs_idx'], meta['dt'], meta['maturity'],
            d['start_X'], d['n_obs_ot'])


def test_loss_close_to_oracle_on_a_2000_path_slice(big):
    b, meta, m = big
    bs = _sub_batch(b, meta, np.arange(2000))
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    with torch.no_grad():
        _, loss = hip_forward(m, bs, meta['dt'], meta['maturity'])
        (_, l_o), _ = oracle_forward(demo_cfg(), sd, bs, meta['dt'], meta['maturity'])
    assert float(loss) == pytest.approx(float(l_o), rel=LOSS_RTOL)


# ---- dropout -----------------------------------------------------------------------------
def test_dropout_is_deterministic_per_seed_and_differs_across_steps():
    cfg = demo_cfg(dropout=0.1)
    b, meta = bs_batch(256, seed=2)
    torch.manual_seed(0)
    m = hip_model(cfg).train()
    losses = []
    for _ in range(3):
        m._step_counter = 7
        with torch.no_grad():
            losses.append(float(hip_forward(m, b, meta['dt'], meta['maturity'])[1]))
    assert losses[0] == losses[1] == losses[2]
    with torch.no_grad():
        other = float(hip_forward(m, b, meta['dt'], meta['maturity'])[1])
    assert other != losses[0]
    m.eval()
    with torch.no_grad():
        ev = float(hip_forward(m, b, meta['dt'], meta['maturity'])[1])
    assert ev != losses[0]


def test_dropout_loss_distribution_matches_oracle():
    """Bitwise parity is impossible with dropout (different RNG); the train-mode loss
    must have the same mean as the oracle's (torch dropout) within 4 standard errors."""
    cfg = demo_cfg(dropout=0.1)
    b, meta = bs_batch(96, seed=4)
    torch.manual_seed(0)
    m = hip_model(cfg).train()
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    n = 80          # (each oracle draw is ~1 s of CPU: 200 draws were a third of the GPU suite)
    hip = []
    with torch.no_grad():
        for _ in range(n):
            hip.append(float(hip_forward(m, b, meta['dt'], meta['maturity'])[1]))
    torch.manual_seed(123)
    ora = []
    with torch.no_grad():
        for _ in range(n):
            (_, l), _ = oracle_forward(cfg, sd, b, meta['dt'], meta['maturity'], training=True)
            ora.append(float(l))
    hip, ora = np.array(hip), np.array(ora)
    se = np.sqrt(hip.var(ddof=1) / n + ora.var(ddof=1) / n)
    assert abs(hip.mean() - ora.mean()) < 4 * se, (hip.mean(), ora.mean(), se)
    assert 0.5 < hip.std() / ora.std() < 2.0


def test_dropout_gradient_matches_finite_differences():
    """With the dropout masks fixed (same seed/step) the analytic gradient must match
    a central finite difference of the loss along a random direction."""
    cfg = demo_cfg(dropout=0.1)
    b, meta = bs_batch(512, seed=6)
    torch.manual_seed(0)
    m = hip_model(cfg).train()
    args = _args(b, meta)
    m._step_counter = 3
    _, loss = m.loss_and_grad(*args)
    g = m.flat_grad().clone()
    flat = m.flat_parameters()
    v = torch.randn_like(flat)
    v /= v.norm()
    base = flat.clone()
    eps = 2e-2
    vals = []
    for s in (+1, -1):
        flat.copy_(base + s * eps * v)
        m._step_counter = 3
        with torch.no_grad():
            vals.append(float(m(*args)[1].double()))
    flat.copy_(base)
    fd = (vals[0] - vals[1]) / (2 * eps)
    an = float((g * v).sum())
    assert fd == pytest.approx(an, rel=3e-2, abs=1e-4)


_ROLE_SNIPPET = r'''
import json, sys, numpy as np, torch
sys.path.insert(0, {tests!r}); sys.path.insert(0, {repo!r})
from hip_util import bs_batch, demo_cfg, hip_model
b, meta = bs_batch(20000, seed=0)
torch.manual_seed(0)
m = hip_model(demo_cfg(dropout=0.1)).train()
m._step_counter = 11
_, loss = m.loss_and_grad(b['times'], b['time_ptr'], b['X'].cuda(), b['obs_idx'].cuda().int(),
                          meta['dt'], meta['maturity'], b['start_X'].cuda(), b['n_obs_ot'].cuda().int())
np.save({out!r}, np.concatenate([[float(loss)], m.flat_grad().cpu().numpy().astype(np.float64)]))
'''


def test_dropout_masks_do_not_depend_on_a_tiles_role(tmp_path):
    """Full-size training step with dropout: the mixed ODE kernels (longest tiles four waves
    wide, different split points forward and backward) must see exactly the masks of the
    one-wave kernels (NJODE_ODE=mfma1), i.e. the same loss and gradient up to summation order."""
    import os, subprocess, sys
    tests = os.path.dirname(os.path.abspath(__file__))
    repo = os.path.dirname(tests)
    res = {}
    for mode in ('mfma', 'mfma1'):
        out = str(tmp_path / (mode + '.npy'))
        env = dict(os.environ, NJODE_ODE=mode)
        p = subprocess.run([sys.executable, '-c', _ROLE_SNIPPET.format(tests=tests, repo=repo, out=out)],
                           env=env, cwd=repo, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert p.returncode == 0, p.stdout[-3000:]
        res[mode] = np.load(out)
    a, b = res['mfma'], res['mfma1']
    assert a[0] == pytest.approx(b[0], rel=1e-5)
    assert rel_l2(a[1:], b[1:]) < 1e-4
