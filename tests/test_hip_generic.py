"""The shape-generic matrix-core kernels (njode_amd/csrc/njode_gen.h): every model shape the
build table has no specialisation for -- widths >= 64, nn_desc=None with wide hidden states,
networks that differ from each other, deeper networks, the climate shape.

* NJODE_GENERIC=1 routes EVERY model without a GRU jump to them: the whole parity suite (all
  reference goldens: eval paths, losses, gradients, Adam steps, masked configs incl. the
  3 000-step one) is re-run that way in a child process;
* shapes of the reference's grids beyond the goldens against the CPU oracle;
* dropout: gradient against central finite differences, mean loss against the oracle's;
* ragged tiles, no observations, data-parallel shards."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from hip_util import (ATOL, GRAD_REL_L2, LOSS_RTOL, RTOL, bs_batch, grads_by_name, hip_forward,
                      hip_model, oracle_forward, rel_l2, to_dev)
from njode_amd import synthetic_physionet

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(2400)
def test_parity_suite_on_the_generic_kernels():
    env = dict(os.environ, NJODE_GENERIC='1')
    cmd = [sys.executable, '-m', 'pytest', os.path.join(REPO, 'tests', 'test_hip_parity.py'),
           os.path.join(REPO, 'tests', 'test_hip_masked_return_path.py'),
           '-m', 'gpu', '-q', '-x', '--timeout', '900', '-p', 'no:cacheprovider']
    p = subprocess.run(cmd, cwd=REPO, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True)
    assert p.returncode == 0, p.stdout[-6000:]


def _cfg(d, H, ode, enc, dec, dropout=0.0, use_rnn=False, **options):
    return dict(input_size=d, hidden_size=H, output_size=d, ode_nn=ode, readout_nn=dec, enc_nn=enc,
                use_rnn=use_rnn, bias=True, dropout_rate=dropout, options=options)


def _w(n, act='tanh', layers=2):
    return tuple((n, act) for _ in range(layers))


UNMASKED = {
    # convergence study / sine / width-400 grids (parallel_train.py:304-305, 609, 712)
    'w80': (_cfg(1, 10, _w(80), _w(80), _w(80)), 37),
    'w160': (_cfg(1, 10, _w(160), _w(160), _w(160)), 21),
    'w400': (_cfg(1, 10, _w(400), _w(400), _w(400)), 17),
    # nn_desc = None with hidden_size 100 (parallel_train.py:366-371)
    'none_h100': (_cfg(1, 100, None, None, None), 33),
    # four hidden layers, relu; no residual; one-layer nets
    'deep_relu': (_cfg(1, 12, _w(24, 'relu', 4), _w(24, 'relu', 4), _w(24, 'relu', 4),
                       residual_enc_dec=False), 40),
    # six hidden layers of three widths, tanh / relu mixed per network (round 4: depth <= 8)
    'deep6_mixed': (_cfg(1, 10, ((40, 'tanh'), (72, 'relu'), (40, 'tanh'), (24, 'tanh'), (72, 'relu'), (40, 'tanh')),
                         _w(33, 'tanh', 5), _w(20, 'relu', 8)), 19),
    'one_layer_easy': (_cfg(2, 6, _w(70, 'tanh', 1), _w(18, 'relu', 1), _w(65, 'tanh', 1),
                            which_loss='easy'), 23),
    # round 4: the GRU jump (models.py:202-217, 353-356, 460-461) on shapes outside the build table
    'gru_w80': (_cfg(1, 10, _w(80), _w(80), _w(80), use_rnn=True), 37),
    'gru_d2_h24': (_cfg(2, 24, _w(33), _w(20, 'relu', 1), _w(72, 'tanh', 3), use_rnn=True,
                        residual_enc_dec=False), 21),
}


@pytest.mark.parametrize('name', sorted(UNMASKED))
def test_unmasked_shapes_against_the_oracle(name):
    cfg, B = UNMASKED[name]
    torch.manual_seed(3)
    m = hip_model(cfg).train()
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    b, meta = bs_batch(B, seed=7)
    if cfg['input_size'] == 2:
        b['X'] = torch.cat([b['X'], b['X'] ** 2], 1)
        b['start_X'] = torch.cat([b['start_X'], b['start_X'] ** 2], 1)
    dt, T = meta['dt'], meta['maturity']
    (h_o, l_o), params = oracle_forward(cfg, sd, b, dt, T, grads=True)
    l_o.backward()
    hT, loss = hip_forward(m, b, dt, T)
    loss.backward()
    assert float(loss) == pytest.approx(float(l_o), rel=LOSS_RTOL)
    np.testing.assert_allclose(hT.detach().cpu().numpy(), h_o.detach().numpy(), atol=ATOL, rtol=RTOL)
    got = grads_by_name(m)
    for k, p in params.items():
        assert rel_l2(got[k], p.grad.numpy()) < GRAD_REL_L2, (k, rel_l2(got[k], p.grad.numpy()))
    # prediction path with the until_T tail, eval mode
    m.eval()
    with torch.no_grad():
        hT2, loss2, path_t, path_h, path_y = hip_forward(m, b, dt, T + 0.03, return_path=True,
                                                         get_loss=True, until_T=True)
        (h2, l2, pt, ph, py), _ = oracle_forward(cfg, sd, b, dt, T + 0.03, return_path=True,
                                                 until_T=True)
    assert np.array_equal(path_t, pt)
    np.testing.assert_allclose(path_y.cpu().numpy(), py.numpy(), atol=ATOL, rtol=RTOL)
    np.testing.assert_allclose(path_h.cpu().numpy(), ph.numpy(), atol=ATOL, rtol=RTOL)
    assert float(loss2) == pytest.approx(float(l2), rel=LOSS_RTOL)


def _masked_batch(dim, B, seed):
    return synthetic_physionet.make_batch(batch_size=B, dim=dim, n_grid=90, n_obs_range=(3, 11),
                                          p_feature=0.3, seed=seed)


MASKED = {
    # climate grids (parallel_train.py:433-470): d = 5, H = 10 / 50, widths 50 / 400
    'climate_w400_h50': (_cfg(5, 50, _w(400), _w(400), _w(400), masked=True), 9),
    # PhysioNet H = 50 non-residual (BASELINE config 5 wording) at width 100
    'physio_h50_w100': (_cfg(41, 50, _w(100), _w(100), _w(100), masked=True,
                             residual_enc_dec=False), 6),
    # residual case 2 in the encoder (input larger than the hidden state)
    'd8_h4': (_cfg(8, 4, _w(20), _w(33, 'relu', 1), None, masked=True), 21),
}


@pytest.mark.parametrize('name', sorted(MASKED))
def test_masked_shapes_against_the_oracle(name):
    cfg, B = MASKED[name]
    torch.manual_seed(5)
    m = hip_model(cfg).train()
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    b = _masked_batch(cfg['input_size'], B, seed=4)
    dt, T = b['delta_t'], b['T']
    (h_o, l_o), params = oracle_forward(cfg, sd, b, dt, T, grads=True)
    l_o.backward()
    hT, loss = hip_forward(m, b, dt, T)
    loss.backward()
    assert float(loss) == pytest.approx(float(l_o), rel=LOSS_RTOL)
    np.testing.assert_allclose(hT.detach().cpu().numpy(), h_o.detach().numpy(), atol=2e-5, rtol=RTOL)
    got = grads_by_name(m)
    for k, p in params.items():
        assert rel_l2(got[k], p.grad.numpy()) < GRAD_REL_L2, (k, rel_l2(got[k], p.grad.numpy()))


def _args(b, dt, T):
    d = to_dev(b)
    return (d['times'], d['time_ptr'], d['X'], d['obs_idx'], dt, T, d['start_X'], d['n_obs_ot'])


@pytest.mark.parametrize('use_rnn', [False, True])
def test_generic_dropout_gradient_matches_finite_differences(use_rnn):
    """(use_rnn: the GRU jump on the lockstep plan, dropout in the three networks around it)"""
    cfg = _cfg(1, 10, _w(100), _w(100), _w(100), dropout=0.1, use_rnn=use_rnn)
    b, meta = bs_batch(200, seed=6)
    torch.manual_seed(0)
    m = hip_model(cfg).train()
    args = _args(b, meta['dt'], meta['maturity'])
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
    # deterministic per (seed, step), different across steps, off in eval mode
    m._step_counter = 3
    with torch.no_grad():
        l3 = float(m(*args)[1])
        l4 = float(m(*args)[1])
    assert l3 == pytest.approx(float(loss), rel=1e-6) and l4 != l3


def test_generic_dropout_loss_distribution_matches_oracle():
    cfg = _cfg(1, 10, _w(100), _w(100), _w(100), dropout=0.1)
    b, meta = bs_batch(64, seed=4)
    torch.manual_seed(0)
    m = hip_model(cfg).train()
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    n = 60          # (each oracle draw at width 100 is ~2 s of CPU)
    with torch.no_grad():
        hip = np.array([float(hip_forward(m, b, meta['dt'], meta['maturity'])[1]) for _ in range(n)])
    torch.manual_seed(123)
    ora = []
    with torch.no_grad():
        for _ in range(n):
            (_, l), _ = oracle_forward(cfg, sd, b, meta['dt'], meta['maturity'], training=True)
            ora.append(float(l))
    ora = np.array(ora)
    se = np.sqrt(hip.var(ddof=1) / n + ora.var(ddof=1) / n)
    assert abs(hip.mean() - ora.mean()) < 4 * se, (hip.mean(), ora.mean(), se)
    assert 0.5 < hip.std() / ora.std() < 2.0


def test_generic_shards_add_up_and_edge_cases():
    cfg = _cfg(1, 10, _w(100), _w(100), _w(100))
    torch.manual_seed(1)
    m = hip_model(cfg).train()
    b, meta = bs_batch(83, seed=9)           # 6 tiles, the last one with 3 paths
    dt, T = meta['dt'], meta['maturity']
    from njode_amd import data_utils
    _, loss = m.loss_and_grad(*_args(b, dt, T))
    g_full = m.flat_grad().clone()
    total, g_sum = 0.0, torch.zeros_like(g_full)
    for lo, hi in ((0, 1), (1, 30), (30, 83)):
        idx = np.arange(lo, hi)
        bs = data_utils.collate_arrays(b['true_paths'][idx], b['observed_dates'][idx],
                                       b['observed_dates'][idx][:, 1:].sum(1), dt)
        m.dp_global_batch, m.dp_path_offset = 83, lo
        _, l = m.loss_and_grad(*_args(bs, dt, T))
        total += float(l)
        g_sum += m.flat_grad()
    m.dp_global_batch, m.dp_path_offset = None, 0
    assert total == pytest.approx(float(loss), rel=2e-5)
    assert rel_l2(g_sum.cpu().numpy(), g_full.cpu().numpy()) < 1e-4
    # nobody observed: loss 0, zero gradient except nothing; hT = encoder(start_X) evolved
    empty = dict(b, times=np.zeros(0), time_ptr=np.zeros(1, dtype=np.int64), X=torch.zeros(0, 1),
                 obs_idx=torch.zeros(0, dtype=torch.long), n_obs_ot=torch.zeros(83, dtype=torch.long))
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    m.eval()
    with torch.no_grad():
        hT, loss0 = hip_forward(m, empty, dt, T, until_T=True)
        (h_o, _), _ = oracle_forward(cfg, sd, empty, dt, T, get_loss=False, until_T=True)
    assert float(loss0) == 0.0
    np.testing.assert_allclose(hT.cpu().numpy(), h_o.numpy(), atol=ATOL, rtol=RTOL)


def test_harness_trains_a_width_100_model():
    """njode_amd.train.train (fused loop, device collate, dropout 0.1) on the shape of
    parallel_train.py:609 -- three hidden-100 networks -- for a few epochs: the validation loss
    falls towards the optimal loss, as it does for the demo shape on the specialised kernels."""
    from njode_amd import data_utils, train
    hp = dict(data_utils.hyperparam_default, nb_paths=2000)
    paths, obs, nb_obs, meta = data_utils.create_dataset('BlackScholes', hp, seed=0)
    nn = _w(100)
    _, metrics = train.train((paths, obs, nb_obs), meta, epochs=6, batch_size=100, ode_nn=nn,
                             readout_nn=nn, enc_nn=nn, dropout_rate=0.1, log=lambda s: None,
                             device_collate=True)
    ev = [m[4] for m in metrics]
    opt = metrics[0][5]
    assert all(np.isfinite(ev))
    assert ev[-1] < 0.8 * ev[0] and ev[-1] > 0.9 * opt


PLAN_CASES = {
    'w100_dropout': (_cfg(1, 10, _w(100), _w(100), _w(100), dropout=0.1), 83),
    'none_h100': (_cfg(1, 100, None, None, None), 33),
    'one_layer_easy_d2': (_cfg(2, 6, _w(70, 'tanh', 1), _w(18, 'relu', 1), _w(65, 'tanh', 1),
                               dropout=0.2, which_loss='easy'), 23),
    'deep_relu_no_residual_current_t': (_cfg(1, 12, _w(24, 'relu', 4), _w(24, 'relu', 4),
                                             _w(24, 'relu', 4), dropout=0.1, residual_enc_dec=False,
                                             input_current_t=True), 40),
}


@pytest.mark.parametrize('name', sorted(PLAN_CASES))
def test_segment_plan_agrees_with_the_lockstep_plan(name, monkeypatch):
    """Unmasked loss calls run the segment plan (njode_gen_seg.h: one work item per
    inter-observation segment); NJODE_GEN_PLAN=lock keeps them on the lockstep plan.  Both draw
    the same dropout masks (keys: seed, global path, Euler step of the event, network, layer,
    unit), so loss, hT and the gradient agree to rounding -- in train mode, dropout on."""
    cfg, B = PLAN_CASES[name]
    torch.manual_seed(11)
    m = hip_model(cfg).train()
    b, meta = bs_batch(B, seed=5)
    if cfg['input_size'] == 2:
        b['X'] = torch.cat([b['X'], b['X'] ** 2], 1)
        b['start_X'] = torch.cat([b['start_X'], b['start_X'] ** 2], 1)
    dt, T = meta['dt'], meta['maturity']
    out = {}
    for plan in ('seg', 'lock'):
        if plan == 'lock':
            monkeypatch.setenv('NJODE_GEN_PLAN', 'lock')
        else:
            monkeypatch.delenv('NJODE_GEN_PLAN', raising=False)
        m._step_counter = 5
        _, loss = m.loss_and_grad(*_args(b, dt, T))
        g = m.flat_grad().clone().cpu().numpy()
        m._step_counter = 5
        with torch.no_grad():
            hT, loss2 = hip_forward(m, b, dt, T + 0.02, until_T=True)      # tails: hT past the last row
        out[plan] = (float(loss), g, hT.cpu().numpy(), float(loss2))
    monkeypatch.delenv('NJODE_GEN_PLAN', raising=False)
    (l_s, g_s, h_s, l2_s), (l_l, g_l, h_l, l2_l) = out['seg'], out['lock']
    assert l_s == pytest.approx(l_l, rel=2e-6)
    assert l2_s == pytest.approx(l2_l, rel=2e-6)
    np.testing.assert_allclose(h_s, h_l, atol=2e-6, rtol=1e-5)
    assert rel_l2(g_s, g_l) < 2e-6, rel_l2(g_s, g_l)
    assert np.abs(g_s).max() > 0


def test_segment_plan_survives_a_malformed_batch():
    """A path listed twice in one time slice breaks the batch layout's contract (a path has at most
    one row per slice; `NJODE_VALIDATE=1` reports it).  The segment plan then loses one of the two
    observations -- it must not read or write out of bounds: the call completes with finite results."""
    cfg = _cfg(1, 10, _w(100), _w(100), _w(100))
    torch.manual_seed(2)
    m = hip_model(cfg).train()
    b, meta = bs_batch(40, seed=3)
    tp = np.asarray(b['time_ptr'])
    sizes = np.diff(tp)
    i = int(np.argmax(sizes >= 2))                      # a slice with at least two rows
    obs = b['obs_idx'].clone()
    obs[tp[i] + 1] = obs[tp[i]]                         # the same path twice
    bad = dict(b, obs_idx=obs)
    _, loss = m.loss_and_grad(*_args(bad, meta['dt'], meta['maturity']))
    torch.cuda.synchronize()
    assert np.isfinite(float(loss))
    assert torch.isfinite(m.flat_grad()).all()
    # and the model is still usable
    _, loss2 = m.loss_and_grad(*_args(b, meta['dt'], meta['maturity']))
    assert np.isfinite(float(loss2)) and float(loss2) > 0


@pytest.mark.timeout(900)
def test_generic_segment_plan_at_config4_shard_size():
    """125 000 paths (BASELINE config 4's per-rank shard) of the width-100 model on the generic
    segment plan: ~1.2 M work items, ~0.8 M ODE records, a 35 GB workspace.  Size-independent
    property: two data-parallel shards add up to the whole batch (loss and gradient), dropout on."""
    sys.path.insert(0, REPO)
    import bench
    N = 125000
    cfg = _cfg(1, 10, _w(100), _w(100), _w(100), dropout=0.1)
    torch.manual_seed(0)
    m = hip_model(cfg).train()
    dev = torch.device('cuda', 0)

    def args_of(lo, hi):
        b, meta = bench.make_global_slice(lo, hi)
        return (b['times'], b['time_ptr'], b['X'].to(dev), b['obs_idx'].to(dev, torch.int32), meta['dt'],
                meta['maturity'], b['start_X'].to(dev), b['n_obs_ot'].to(dev, torch.int32))

    m._step_counter = 1
    m.dp_global_batch, m.dp_path_offset = N, 0
    _, loss = m.loss_and_grad(*args_of(0, N))
    g_full = m.flat_grad().clone()
    assert np.isfinite(float(loss)) and float(loss) > 0
    total, g_sum = 0.0, torch.zeros_like(g_full)
    for lo, hi in ((0, 60000), (60000, N)):
        m._step_counter = 1
        m.dp_global_batch, m.dp_path_offset = N, lo
        _, l = m.loss_and_grad(*args_of(lo, hi))
        total += float(l)
        g_sum += m.flat_grad()
    m.dp_global_batch, m.dp_path_offset = None, 0
    assert total == pytest.approx(float(loss), rel=1e-5)
    assert float((g_sum - g_full).norm() / g_full.norm()) < 1e-4


@pytest.mark.parametrize('name', ['g14_out5', 'g14_out20'])
def test_output_size_other_than_input_size_predicts_and_refuses_the_loss(name):
    """Round 5 (VERDICT r4 missing 3): `models.py:350-352` builds a readout to any output_size.
    The loss compares X with the readout, so such a model runs prediction calls only: the HIP
    path against the reference's own prediction path (make_golden.py:g14), and a loud error --
    not a wrong number -- when the loss is asked for."""
    from golden_util import Golden
    from hip_util import ATOL, RTOL, hip_forward, hip_model
    g = Golden(name)
    m = hip_model(g.cfg, g.state_dict()).eval()
    with torch.no_grad():
        hT, loss, path_t, path_h, path_y = hip_forward(m, g.batch(), g.delta_t, g.T, return_path=True,
                                                       get_loss=False, until_T=True)
        hT2, loss2 = hip_forward(m, g.batch(), g.delta_t, g.T, get_loss=False)
    assert loss == 0 and loss2 == 0
    assert np.array_equal(path_t, g['path_t'])
    assert tuple(path_y.shape) == g['path_y'].shape
    np.testing.assert_allclose(path_y.cpu().numpy(), g['path_y'], atol=ATOL, rtol=RTOL)
    np.testing.assert_allclose(path_h.cpu().numpy(), g['path_h'], atol=ATOL, rtol=RTOL)
    np.testing.assert_allclose(hT.cpu().numpy(), g['hT'], atol=ATOL, rtol=RTOL)
    np.testing.assert_allclose(hT2.cpu().numpy(), g['hT_lastobs'], atol=ATOL, rtol=RTOL)
    with pytest.raises(Exception, match='input_size != output_size'):
        hip_forward(m, g.batch(), g.delta_t, g.T)          # get_loss=True


def test_use_rnn_with_masked_data_matches_reference():
    """Round 5: the GRU jump with masked data (reference: models.py:353 TODO, :460-461 -- it runs):
    prediction path, loss and every gradient against the reference's own run (make_golden.py:g15)."""
    from golden_util import Golden
    from hip_util import hip_forward, hip_model
    g = Golden('g15_rnn_masked')
    m = hip_model(g.cfg, g.state_dict()).eval()
    with torch.no_grad():
        hT, loss, path_t, path_h, path_y = hip_forward(m, g.batch(), g.delta_t, g.T, return_path=True,
                                                       get_loss=True, until_T=True)
    assert np.array_equal(path_t, g['path_t'])
    np.testing.assert_allclose(path_y.cpu().numpy(), g['path_y'], atol=2e-5, rtol=RTOL)
    np.testing.assert_allclose(hT.cpu().numpy(), g['hT'], atol=2e-5, rtol=RTOL)
    assert float(loss) == pytest.approx(float(g['loss']), rel=LOSS_RTOL)
    m.train()
    _, loss = hip_forward(m, g.batch(), g.delta_t, g.T)
    loss.backward()
    assert float(loss) == pytest.approx(float(g['train_loss']), rel=LOSS_RTOL)
    got = grads_by_name(m)
    for k, ref in g.group('grad').items():
        assert rel_l2(got[k], ref) < GRAD_REL_L2, (k, rel_l2(got[k], ref))
