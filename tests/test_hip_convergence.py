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
BAND = {'BlackScholes': 0.10, 'OrnsteinUhlenbeck': 0.25, 'Heston': 0.03}
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
