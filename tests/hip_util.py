"""Shared helpers of the GPU parity tests: build the HIP-backed model and the CPU
oracle from the same config / parameters and run both on the same batch."""
import copy

import numpy as np
import torch

from njode_amd import data_utils, models
from oracle import njode_oracle

NN50 = ((50, 'tanh'), (50, 'tanh'))
# fp32 tolerances of the HIP path vs the reference/oracle:
ATOL, RTOL = 1e-5, 1e-4        # hT, path_h, path_y at S = 100
RTOL_LONG = 1e-3               # ... at S = 3 000 (PhysioNet-shaped, masked)
LOSS_RTOL = 1e-4
GRAD_REL_L2 = 1e-3             # per-tensor relative L2 error of gradients


def demo_cfg(d=1, H=10, dropout=0.0, **options):
    return dict(input_size=d, hidden_size=H, output_size=d, ode_nn=NN50, readout_nn=NN50,
                enc_nn=NN50, use_rnn=False, bias=True, dropout_rate=dropout, options=options)


def hip_model(cfg, state_dict=None, device='cuda', device_outputs=True):
    cfg = copy.deepcopy(cfg)
    cfg.setdefault('options', {})
    cfg['options'] = dict(cfg['options'], device_outputs=device_outputs)
    m = models.NJODE(**cfg)
    if state_dict is not None:
        m.load_state_dict(state_dict)
    return m.to(device)


def to_dev(b, device='cuda'):
    out = dict(b)
    for k in ('X', 'start_X', 'n_obs_ot', 'M'):
        if k in out and out[k] is not None:
            out[k] = out[k].to(device)
    return out


def hip_forward(m, b, delta_t, T, **kw):
    d = to_dev(b)
    return m(d['times'], d['time_ptr'], d['X'], d['obs_idx'], delta_t, T, d['start_X'],
             d.get('n_obs_ot'), M=d.get('M'), **kw)


def oracle_forward(cfg, sd, b, delta_t, T, training=False, weight=None, grads=False, **kw):
    o = njode_oracle.make_oracle(cfg)
    o.training = training
    if weight is not None:
        o.weight = weight
    params = {k: v.clone().requires_grad_(grads) for k, v in sd.items()}
    out = o.forward(params, b['times'], b['time_ptr'], b['X'], b['obs_idx'], delta_t, T,
                    b['start_X'], b.get('n_obs_ot'), M=b.get('M'), **kw)
    return out, params


def rel_l2(got, ref):
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    return float(np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-12))


def bs_batch(n_paths, seed=0, name='BlackScholes', obs_perc=0.1, nb_steps=100):
    hp = copy.deepcopy(data_utils.hyperparam_default)
    hp.update(nb_paths=n_paths, obs_perc=obs_perc, nb_steps=nb_steps)
    paths, obs, nb_obs, meta = data_utils.create_dataset(name, hp, seed=seed)
    b = data_utils.collate_arrays(paths, obs, nb_obs, meta['dt'])
    return b, meta


def grads_by_name(m):
    return {k: p.grad.detach().cpu().numpy() for k, p in m.named_parameters()}
