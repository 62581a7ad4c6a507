"""Diagnostic micro-benchmarks of the training step (not part of bench.py's contract):
per-kernel HIP-event times for workload variants that separate throughput from
tail / imbalance effects.  Usage: python tools/bench_variants.py [--steps 5]"""
import argparse
import copy
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from njode_amd import _lib, data_utils, models  # noqa: E402

NN = ((50, 'tanh'), (50, 'tanh'))


def make(n_paths, seed=0, balanced=None):
    hp = copy.deepcopy(data_utils.hyperparam_default)
    hp['nb_paths'] = n_paths
    paths, obs, nb_obs, meta = data_utils.create_dataset('BlackScholes', hp, seed=seed)
    if balanced:
        obs = np.zeros_like(obs)
        obs[:, balanced::balanced] = 1           # every path observed every `balanced` steps
        nb_obs = obs[:, 1:].sum(1)
    return data_utils.collate_arrays(paths, obs, nb_obs, meta['dt']), meta


def run(name, n_paths, dropout, steps, balanced=None, train=True, warmup=2):
    b, meta = make(n_paths, balanced=balanced)
    cfg = dict(input_size=1, hidden_size=10, output_size=1, ode_nn=NN, readout_nn=NN, enc_nn=NN,
               use_rnn=False, bias=True, dropout_rate=dropout, options={'device_outputs': True})
    torch.manual_seed(0)
    m = models.NJODE(**cfg).cuda()
    m.train(train)
    opt = models.FusedAdam(m, lr=1e-3, weight_decay=0.0005)
    args = (b['times'], b['time_ptr'], b['X'].cuda(), b['obs_idx'].cuda().int(), meta['dt'],
            meta['maturity'], b['start_X'].cuda(), b['n_obs_ot'].cuda().int())

    def step():
        if train:
            m.loss_and_grad(*args)
            opt.step()
        else:
            with torch.no_grad():
                m(*args)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    _lib.profile_enable(False)
    k = {n: round(v[1] / v[0], 4) for n, v in _lib.profile_read().items()}
    print(json.dumps({'case': name, 'paths': n_paths, 'dropout': dropout, 'balanced': balanced,
                      'train': train, 'ms_per_step': round(1e3 * el / steps, 4),
                      'paths_per_s': round(n_paths * steps / el, 1), 'kernel_ms': k}), flush=True)


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--quick', action='store_true')
    a = ap.parse_args()
    print(json.dumps({'NJODE_WSRC': os.environ.get('NJODE_WSRC', 'lds(default)')}), flush=True)
    if a.quick:
        run('default', 20000, 0.1, a.steps)
        run('balanced-10', 20000, 0.0, a.steps, balanced=10)
        run('eval-forward', 20000, 0.0, a.steps, train=False)
        run('B=100', 100, 0.1, 10)
        sys.exit(0)
    run('default', 20000, 0.1, a.steps)
    run('no-dropout', 20000, 0.0, a.steps)
    run('balanced-10', 20000, 0.0, a.steps, balanced=10)
    run('balanced-10-dropout', 20000, 0.1, a.steps, balanced=10)
    run('balanced-50', 20000, 0.0, a.steps, balanced=50)
    run('eval-forward', 20000, 0.0, a.steps, train=False)
    run('B=100', 100, 0.1, 20)
    run('B=200', 200, 0.1, 20)
    run('B=2000', 2000, 0.1, 10)
    run('B=100000', 100000, 0.1, 3)
