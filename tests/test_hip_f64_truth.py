"""float64 truth test of the masked fixtures (round 5).

tests/golden/g13_f64_truth.npz holds the REFERENCE's model evaluated in float64 on the stored
parameters and inputs of g5_masked / g5_full / g5_w200 (make_golden.py:g13).  Masked mode feeds
every prediction back as an input (models.py:465-467, 483-484), so two correct fp32 evaluations
drift apart by more than a plain 1e-5; what a tolerance against the reference's fp32 output
cannot tell is WHICH of the two is off.  Here both are measured against the float64 result:

    err(HIP fp32, f64)  <=  1.5 x err(reference fp32, f64)      (max-abs and L2; round 5: 2 x)

on the prediction path, hT, the loss and every gradient tensor -- i.e. the HIP result is no
further from the truth than the reference itself is (factor 2: two fp32 evaluations of the same
recursion round differently, neither is privileged).  Round 6: the prediction calls of the specialised
masked shapes run the wave-per-path forward (njode_chain.h; two-accumulator fma chains), which ends
0.65 x (g5_masked) / 0.98 x (g5_full) the reference's distance from float64 where the matrix-core
lockstep forward ended 1.84 x / 1.40 x (profiles/r06_f64_truth.txt): FACTOR 2 -> 1.5.  The CPU half (not gpu) checks the fixture
against the fp32 goldens, so the truth cannot silently be something else.
"""
import numpy as np
import pytest
import torch

from golden_util import GOLDEN_DIR, Golden

CASES = ['g5_masked', 'g5_full', 'g5_w200']
FACTOR = 1.5      # (round 6: 2.0 -> 1.5; measured 0.65 / 0.75 (g5_masked), 0.98 / 1.05 (g5_full), 0.81 / 0.98 (g5_w200))
# Gradients: where the exact discrete adjoint on stored activations is CLOSER to
# float64 than the reference's fp32 autograd, the bound says so.  Round 6, wave-per-path kernels with
# two-accumulator dot products: g5_masked 0.47 - 0.70, g5_full 0.93 - 0.96 (profiles/r06_f64_truth.txt);
# g5_w200 runs the shape-generic kernels (0.85 - 1.67, round 5).
GRAD_FACTOR = {'g5_masked': 1.0, 'g5_full': 1.25, 'g5_w200': 2.0}


def _truth():
    import os
    return np.load(os.path.join(GOLDEN_DIR, 'g13_f64_truth.npz'), allow_pickle=False)


def _errs(x, truth):
    d = np.asarray(x, dtype=np.float64) - np.asarray(truth, dtype=np.float64)
    return float(np.abs(d).max()), float(np.linalg.norm(d))


@pytest.mark.parametrize('name', CASES)
def test_truth_fixture_is_the_f64_twin_of_the_fp32_golden(name):
    """CPU: same shapes, float64, and the reference's fp32 output sits within fp32 drift of it."""
    g, t = Golden(name), _truth()
    y32, y64 = g['path_y'], t[name + '/path_y']
    assert y64.dtype == np.float64 and y64.shape == y32.shape
    e_max, _ = _errs(y32, y64)
    assert 0.0 < e_max < 1e-3 * max(1.0, float(np.abs(y64).max()))
    assert float(g['loss']) == pytest.approx(float(t[name + '/loss']), rel=1e-5)
    for k, ref in g.group('grad').items():
        g64 = t[name + '/grad/' + k]
        assert g64.dtype == np.float64 and g64.shape == ref.shape
        assert np.linalg.norm(ref - g64) <= 1e-3 * np.linalg.norm(g64) + 1e-12, k


@pytest.mark.gpu
@pytest.mark.parametrize('name', CASES)
def test_hip_is_as_close_to_float64_as_the_reference_is(name):
    import json
    import os
    from hip_util import grads_by_name, hip_forward, hip_model
    g, t = Golden(name), _truth()
    m = hip_model(g.cfg, g.state_dict()).eval()
    with torch.no_grad():
        hT, loss, path_t, path_h, path_y = hip_forward(
            m, g.batch(), g.delta_t, g.T, return_path=True, get_loss=True, until_T=True)
    rows = g['path_rows'] if 'path_rows' in g else slice(None)
    rep, bad = {'fixture': name}, []

    def check(key, got, ref32, truth, floor_max=0.0, floor_l2=0.0, factor=FACTOR, max_too=True):
        h_max, h_l2 = _errs(got, truth)
        r_max, r_l2 = _errs(ref32, truth)
        rep[key] = {'hip_max': h_max, 'ref_max': r_max, 'hip_l2': h_l2, 'ref_l2': r_l2}
        if max_too and h_max > factor * r_max + floor_max:
            bad.append((key, 'max', h_max, r_max))
        if h_l2 > factor * r_l2 + floor_l2:
            bad.append((key, 'l2', h_l2, r_l2))

    check('path_y', path_y.cpu().numpy()[rows], g['path_y'], t[name + '/path_y'])
    check('hT', hT.cpu().numpy(), g['hT'], t[name + '/hT'])
    eps = float(np.finfo(np.float32).eps)
    # (the loss is ONE fp32 number: its error against float64 is a handful of ulps for either
    # side; a floor of 4 ulp keeps the comparison from being a coin toss)
    l64 = float(t[name + '/loss'])
    check('loss', float(loss), float(g['loss']), l64, floor_max=4 * eps * abs(l64), floor_l2=4 * eps * abs(l64))
    # gradients
    m.train()
    _, tl = hip_forward(m, g.batch(), g.delta_t, g.T)
    tl.backward()
    got = grads_by_name(m)
    tl64 = float(t[name + '/train_loss'])
    check('train_loss', float(tl.detach()), float(g['train_loss']), tl64, floor_max=4 * eps * abs(tl64),
          floor_l2=4 * eps * abs(tl64))
    worst = 0.0
    for k, ref32 in g.group('grad').items():
        truth = t[name + '/grad/' + k]
        # per tensor, with a floor of one fp32 ulp of the tensor's largest entry (tiny bias
        # gradients: either side is then exact to rounding)
        ulp = eps * float(np.abs(truth).max())
        # (L2 against the tightened per-fixture factor; the single worst entry against the general one)
        check('grad/' + k, got[k], ref32, truth, floor_max=4 * ulp, floor_l2=ulp * np.sqrt(truth.size))
        check('grad/' + k, got[k], ref32, truth, floor_l2=ulp * np.sqrt(truth.size), factor=GRAD_FACTOR[name],
              max_too=False)
        r = rep['grad/' + k]
        worst = max(worst, r['hip_l2'] / max(r['ref_l2'], 1e-300))
    rep['worst_gradient_l2_ratio'] = worst
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    try:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, 'f64_truth_report.jsonl'), 'a') as f:
            f.write(json.dumps(rep) + '\n')
    except OSError:
        pass
    p = rep['path_y']
    print('{}: path_y max err HIP {:.3e} / reference {:.3e}; L2 {:.3e} / {:.3e}; worst gradient '
          'L2 ratio {:.2f}'.format(name, p['hip_max'], p['ref_max'], p['hip_l2'], p['ref_l2'], worst))
    assert not bad, bad
