"""End-to-end training with dropout ON, pinned to the reference's shipped training curves
(VERDICT r2 item 6).  The build's harness (njode_amd.train.train = reference train.py:488-624)
trains the demo model for 30 epochs on the reference's own recipe -- seed-0 20 000-path dataset,
split seed 398, batch 200, Adam lr 1e-3 / weight decay 5e-4, dropout 0.1 -- and its validation
loss must track the curve the reference logged for the same recipe
(data/saved_models/id-{1,2,3}/metric_id-N.csv, stored as data in
tests/golden/g9_ref_training_curves.npz).

What is comparable: the reference's runs used another realisation of the 20 000 paths (its
logged optimal_eval_loss differs from the one of the seed-0 dataset by 1.5-2.5 %) and another
weight initialisation, so the quantity compared is the EXCESS of the validation loss over the
optimal loss of the respective validation set, (eval - optimal) / |optimal|.  Bands (stated,
not tuned per epoch): the build's excess may exceed the reference's at the same epoch by at most
BAND[dataset]; it must fall over training; and the final loss must not undercut the optimal loss
by more than the reference's own curve does (Heston: the logged curve ends 2 % below its
'optimal' loss)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'tools'))

EPOCHS = 30
# absolute band on the excess ratio; observed gaps (profiles/r03_convergence_*.jsonl):
# BS -0.02 ... -0.22 (the build converges faster), OU +0.04 ... +0.14, Heston -0.013 ... -0.025
BAND = {'BlackScholes': 0.10, 'OrnsteinUhlenbeck': 0.18, 'Heston': 0.03}
UNDERCUT = {'BlackScholes': 0.02, 'OrnsteinUhlenbeck': 0.02, 'Heston': 0.05}


@pytest.mark.parametrize('name', ['BlackScholes', 'OrnsteinUhlenbeck', 'Heston'])
def test_training_tracks_the_reference_curve(name):
    import convergence_run
    rows, ref = convergence_run.run(name, EPOCHS)
    assert len(rows) == EPOCHS
    ev = np.array([r['eval_loss'] for r in rows])
    ex = np.array([r['excess'] for r in rows])
    opt = rows[0]['optimal']
    assert np.isfinite(ev).all()
    # same data distribution: the optimal losses of the two validation sets agree within 5 %
    assert opt == pytest.approx(float(ref['optimal'][0]), rel=0.05)
    # the first epoch starts where the reference's does (same model, same data distribution)
    assert 0.5 * ref['eval_loss'][0] < ev[0] < 1.5 * ref['eval_loss'][0]
    # learning happens: the loss falls over training
    assert ev[9] < ev[0] and ev[-5:].mean() < ev[:5].mean()
    # tracks the shipped curve at the same epochs, within the stated band
    for e in (10, 20, 30):
        assert ex[e - 1] <= ref['excess'][e - 1] + BAND[name], (e, ex[e - 1], ref['excess'][e - 1])
    # ends between the optimal loss and the reference's value at this epoch + band
    assert ev[-1] >= opt * (1.0 - UNDERCUT[name]), (ev[-1], opt)
    assert ex[-1] <= ref['excess'][EPOCHS - 1] + BAND[name]


@pytest.mark.parametrize('name', ['OrnsteinUhlenbeck', 'BlackScholes'])
def test_training_matches_the_reference_trained_like_for_like(name):
    """VERDICT r3 item 3a.  The shipped curves above come from another realisation of the data and
    another initialisation; `g9b_ref_seeded_curves` is the REFERENCE's own model and loop
    (train.py:488-574) run on exactly what this harness feeds its model -- seed-0 dataset, split
    398, batch 200, the same epoch orders, the same initial weights, Adam lr 1e-3 / wd 5e-4,
    dropout 0.1 -- for 12 epochs and three dropout seeds.  The only thing that differs between the
    two runs is the dropout stream (torch's generator there, the kernels' counter-based stream
    here), so the seed AVERAGES must agree: per epoch, the build's mean excess over the optimal
    loss within the reference's seed spread + 10 % of the reference's mean excess."""
    import convergence_run
    mine, ref, _ = convergence_run.run_seeded(name)
    assert mine.shape == ref.shape and np.isfinite(mine).all()
    m_b, m_r = mine.mean(axis=0), ref.mean(axis=0)
    spread = ref.max(axis=0) - ref.min(axis=0)
    band = spread + 0.10 * np.abs(m_r)
    worst = np.abs(m_b - m_r) / band
    assert (worst[1:] <= 1.0).all(), (name, np.round(m_b, 3), np.round(m_r, 3), np.round(band, 3))
    # epoch 1 (one pass over a fresh model, the steepest part of the curve): 20 %
    assert abs(m_b[0] - m_r[0]) <= 0.2 * m_r[0]
