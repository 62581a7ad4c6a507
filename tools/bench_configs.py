"""Measurements for the BASELINE.json configs other than the headline one (diagnostic;
bench.py is the contract).  Prints one JSON line per case:
  config2-sweep : training-step paths/s on Black-Scholes for several batch sizes
  config5       : PhysioNet-shaped masked model (B=50, 3000 Euler steps, d=41), forward and
                  training step, both hidden sizes (41 residual / 50 non-residual), with the
                  CPU oracle timed on the same batch
Usage: python tools/bench_configs.py [--skip-cpu]"""
import argparse
import copy
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from njode_amd import data_utils, models, synthetic_physionet  # noqa: E402
from oracle import njode_oracle  # noqa: E402

NN = ((50, 'tanh'), (50, 'tanh'))


def timed(fn, steps, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def sweep():
    cfg = dict(input_size=1, hidden_size=10, output_size=1, ode_nn=NN, readout_nn=NN, enc_nn=NN,
               use_rnn=False, bias=True, dropout_rate=0.1, options={'device_outputs': True})
    for B, steps in ((100, 30), (200, 30), (1000, 20), (4000, 20), (16000, 10), (20000, 10),
                     (100000, 5), (500000, 2)):
        hp = copy.deepcopy(data_utils.hyperparam_default)
        hp['nb_paths'] = B
        paths, obs, nb_obs, meta = data_utils.create_dataset('BlackScholes', hp, seed=0)
        b = data_utils.collate_arrays(paths, obs, nb_obs, meta['dt'])
        torch.manual_seed(0)
        m = models.NJODE(**cfg).cuda().train()
        opt = models.FusedAdam(m, lr=1e-3, weight_decay=0.0005)
        args = (b['times'], b['time_ptr'], b['X'].cuda(), b['obs_idx'].cuda().int(), meta['dt'],
                meta['maturity'], b['start_X'].cuda(), b['n_obs_ot'].cuda().int())

        def step():
            m.loss_and_grad(*args)
            opt.step()
        t = timed(step, steps)
        m.eval()

        def fwd():
            with torch.no_grad():
                m(*args)
        tf = timed(fwd, steps)
        print(json.dumps({'case': 'config2-sweep', 'B': B, 'train_ms': round(t * 1e3, 4),
                          'train_paths_per_s': round(B / t, 1), 'eval_fwd_ms': round(tf * 1e3, 4),
                          'eval_paths_per_s': round(B / tf, 1)}), flush=True)


def physionet(skip_cpu):
    for H, res in ((41, True), (50, False)):
        cfg = dict(input_size=41, hidden_size=H, output_size=41, ode_nn=NN, readout_nn=NN,
                   enc_nn=NN, use_rnn=False, bias=True, dropout_rate=0.1,
                   options={'masked': True, 'residual_enc_dec': res, 'device_outputs': True})
        b = synthetic_physionet.make_batch(batch_size=50, seed=0)
        torch.manual_seed(0)
        m = models.NJODE(**cfg).cuda().train()
        opt = models.FusedAdam(m, lr=1e-3, weight_decay=0.0005)
        args = (b['times'], b['time_ptr'], b['X'].cuda(), b['obs_idx'].cuda().int(), b['delta_t'],
                b['T'], b['start_X'].cuda(), b['n_obs_ot'].cuda().int())
        kw = {'M': b['M'].cuda()}

        def step():
            m.loss_and_grad(*args, **kw)
            opt.step()
        t = timed(step, 3, warmup=1)
        m.eval()

        def fwd():
            with torch.no_grad():
                m(*args, **kw)
        tf = timed(fwd, 3, warmup=1)
        rec = {'case': 'config5', 'hidden_size': H, 'residual': res, 'B': 50, 'd': 41,
               'euler_steps': 3000, 'n_times': int(len(b['times'])),
               'n_obs_rows': int(b['time_ptr'][-1]), 'params': int(m.flat_parameters().numel()),
               'train_ms': round(t * 1e3, 3), 'train_paths_per_s': round(50 / t, 2),
               'eval_fwd_ms': round(tf * 1e3, 3)}
        if not skip_cpu:
            torch.set_num_threads(8)
            o = njode_oracle.make_oracle(cfg)
            o.training = True
            params = {k: v.clone().requires_grad_(True) for k, v in o.init_params(0).items()}
            opt_c = torch.optim.Adam(list(params.values()), lr=1e-3, weight_decay=0.0005)
            t0 = time.perf_counter()
            opt_c.zero_grad()
            _, loss = o.forward(params, b['times'], b['time_ptr'], b['X'], b['obs_idx'],
                                b['delta_t'], b['T'], b['start_X'], b['n_obs_ot'], M=b['M'])
            loss.backward()
            opt_c.step()
            rec['cpu_oracle_train_ms_8threads'] = round((time.perf_counter() - t0) * 1e3, 1)
        print(json.dumps(rec), flush=True)


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--skip-cpu', action='store_true')
    a = ap.parse_args()
    sweep()
    physionet(a.skip_cpu)
