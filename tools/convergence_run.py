"""End-to-end training of the demo model with the build's harness (njode_amd.train.train =
reference train.py:488-624 semantics) on the reference's own recipe: seed-0 20 000-path dataset,
split seed 398 (16 000 / 4 000), batch 200, Adam lr 1e-3 wd 5e-4, dropout 0.1, weight 0.5 --
the setting of the three models the reference ships (data/saved_models/model_overview.csv).
Prints one JSON line per epoch with the reference's shipped curve (tests/golden/
g9_ref_training_curves.npz) beside it, as excess over the optimal loss, the quantity that is
comparable across dataset realisations:

    python tools/convergence_run.py [BlackScholes|Heston|OrnsteinUhlenbeck] [epochs]
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def run(name='BlackScholes', epochs=30, batch_size=200, device_collate=True, log=None, **kw):
    """Returns (rows, ref): rows[e] = dict(epoch, train_loss, eval_loss, optimal, excess) of the
    build's run; ref = the same columns of the reference's shipped curve."""
    from golden_util import Golden
    from njode_amd import data_utils, train
    g = Golden('g9_ref_training_curves')
    hp = dict(data_utils.hyperparam_default, nb_paths=20000)
    paths, obs, nb_obs, meta = data_utils.create_dataset(name, hp, seed=0)
    _, metrics = train.train((paths, obs, nb_obs), meta, epochs=epochs, batch_size=batch_size,
                             learning_rate=1e-3, dropout_rate=0.1, seed=398, test_size=0.2,
                             log=log or (lambda s: None), device_collate=device_collate, **kw)
    rows = []
    for m in metrics:
        ep, _, _, tl, el, opt = m[:6]
        rows.append({'epoch': int(ep), 'train_loss': tl, 'eval_loss': el, 'optimal': opt,
                     'excess': (el - opt) / abs(opt)})
    r_opt = g[name + '/optimal_eval_loss']
    ref = {'eval_loss': g[name + '/eval_loss'], 'optimal': r_opt,
           'excess': (g[name + '/eval_loss'] - r_opt) / np.abs(r_opt)}
    return rows, ref


def run_seeded(name='OrnsteinUhlenbeck', seeds=(0, 1, 2), log=None):
    """Like-for-like against tests/golden/g9b_ref_seeded_curves.npz (the REFERENCE model trained
    on this harness' own recipe: same dataset, split, batches, epoch orders and initial weights,
    dropout masks from three torch seeds).  Returns (excess[seed][epoch] of the build, the same
    of the reference, optimal loss): dropout streams differ (torch's generator vs the kernels'
    counter-based one), so the comparison is between seed averages."""
    import torch
    from golden_util import Golden
    from njode_amd import data_utils, train
    g = Golden('g9b_ref_seeded_curves')
    epochs = int(g.cfg['epochs'])
    hp = dict(data_utils.hyperparam_default, nb_paths=20000)
    paths, obs, nb_obs, meta = data_utils.create_dataset(name, hp, seed=0)
    init = {k[len(name) + 6:]: torch.from_numpy(g[k]) for k in g.z.files if k.startswith(name + '/init/')}
    ref_opt = float(g[name + '/optimal_eval_loss'])
    out = []
    for sd in seeds:
        _, metrics = train.train((paths, obs, nb_obs), meta, epochs=epochs, batch_size=int(g.cfg['batch_size']),
                                 learning_rate=1e-3, dropout_rate=0.1, seed=398, test_size=0.2,
                                 log=log or (lambda s: None), device_collate=True, init_state=init,
                                 dropout_seed=1000 + int(sd))
        opt = metrics[0][5]
        assert abs(opt - ref_opt) <= 1e-6 * abs(ref_opt), (opt, ref_opt)     # the same validation set
        out.append([(m[4] - opt) / abs(opt) for m in metrics])
    ref = (g[name + '/eval_loss'] - ref_opt) / abs(ref_opt)
    return np.array(out), ref, ref_opt


if __name__ == '__main__':
    name = sys.argv[1] if len(sys.argv) > 1 else 'BlackScholes'
    epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    rows, ref = run(name, epochs)
    for r in rows:
        e = r['epoch'] - 1
        print(json.dumps(dict(r, dataset=name, ref_eval_loss=float(ref['eval_loss'][e]),
                              ref_optimal=float(ref['optimal'][e]),
                              ref_excess=float(ref['excess'][e]))), flush=True)
