"""Helpers to read the golden vectors in tests/golden/ (made by make_golden.py
from the reference itself)."""
import json
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _tuplify(nn_desc):
    if nn_desc is None:
        return None
    return tuple((int(w), str(a)) for w, a in nn_desc)


class Golden:
    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN_DIR, name + '.npz'), allow_pickle=False)
        self.cfg = json.loads(str(self.z['cfg_json']))
        for k in ('ode_nn', 'enc_nn', 'readout_nn'):
            if k in self.cfg:
                self.cfg[k] = _tuplify(self.cfg[k])

    def __getitem__(self, k):
        return self.z[k]

    def __contains__(self, k):
        return k in self.z.files

    def group(self, prefix):
        p = prefix + '/'
        return {k[len(p):]: self.z[k] for k in self.z.files if k.startswith(p)}

    def state_dict(self):
        return {k: torch.tensor(v) for k, v in self.group('sd').items()}

    def batch(self):
        b = {'times': self.z['times'], 'time_ptr': self.z['time_ptr'],
             'X': torch.tensor(self.z['X']),
             'obs_idx': torch.tensor(self.z['obs_idx'], dtype=torch.long),
             'start_X': torch.tensor(self.z['start_X']),
             'n_obs_ot': torch.tensor(self.z['n_obs_ot'])}
        if 'M' in self.z.files:
            b['M'] = torch.tensor(self.z['M'])
        return b

    @property
    def delta_t(self):
        return float(self.z['delta_t'])

    @property
    def T(self):
        return float(self.z['T'])


def all_model_cases():
    names = sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith('.npz'))
    return [n for n in names if n.startswith(('g1_', 'g2_', 'g5_', 'g6_'))]
