"""
Batch producer for the NJ-ODE hot path: synthetic datasets and the collate that
turns (paths, observation mask) into the CSR-by-time batch ``NJODE.forward``
consumes.

Mirrors the interface of the reference's ``NJODE/data_utils.py`` for the part
that feeds the hot path:

* ``hyperparam_default``            -- reference ``data_utils.py:25-31``
* ``create_dataset`` (in memory)    -- reference ``data_utils.py:59-108``
  (RNG order: paths first, then the observation mask)
* ``save_dataset`` / ``load_dataset_dir`` -- the reference's on-disk layout
  ``data.npy`` (three consecutive ``np.save``) + ``metadata.txt``
  (``data_utils.py:98-105, 231-249``)
* ``IrregularDataset``              -- reference ``data_utils.py:252-275``
* ``custom_collate_fn``             -- reference ``data_utils.py:278-316``
* ``CustomCollateFnGen``            -- reference ``data_utils.py:352-416``

The collate is vectorised (``np.nonzero`` on the transposed mask) instead of the
reference's O(S*B) Python loop; the output arrays are identical, including the
float64 accumulation of the observation times (``current_time += dt``).
"""
import copy
import json
import os

import numpy as np
import torch

from . import stock_model

hyperparam_default = {
    'drift': 2., 'volatility': 0.3, 'mean': 4,
    'speed': 2., 'correlation': 0.5, 'nb_paths': 10000, 'nb_steps': 100,
    'S0': 1, 'maturity': 1., 'dimension': 1,
    'obs_perc': 0.1,
    'scheme': 'euler', 'return_vol': False, 'v0': 1,
}

_STOCK_MODELS = stock_model.STOCK_MODELS


def create_dataset(stock_model_name="BlackScholes",
                   hyperparam_dict=hyperparam_default, seed=0):
    """Generate a synthetic dataset in memory.

    Same RNG protocol as the reference (``data_utils.py:73-81``): legacy
    ``np.random.seed(seed)``, then the model's ``generate_paths()``, then one
    uniform draw per (path, grid point) for the observation mask.

    :return: (stock_paths f64 [N, d, S+1], observed_dates int [N, S+1],
              nb_obs int [N], metadata dict incl. 'dt' and 'model_name')
    """
    hp = copy.deepcopy(hyperparam_dict)
    np.random.seed(seed=seed)
    hp['model_name'] = stock_model_name
    model = _STOCK_MODELS[stock_model_name](**hp)
    stock_paths, dt = model.generate_paths()
    n, _, s1 = stock_paths.shape
    observed_dates = (np.random.random(size=(n, s1)) < hp['obs_perc']) * 1
    nb_obs = np.sum(observed_dates[:, 1:], axis=1)
    hp['dt'] = dt
    return stock_paths, observed_dates, nb_obs, hp


def save_dataset(path, stock_paths, observed_dates, nb_obs, metadata):
    """Write the reference's on-disk dataset layout (``data_utils.py:98-105``)."""
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, 'data.npy'), 'wb') as f:
        np.save(f, stock_paths)
        np.save(f, observed_dates)
        np.save(f, nb_obs)
    with open(os.path.join(path, 'metadata.txt'), 'w') as f:
        json.dump(metadata, f, sort_keys=True)


def load_dataset_dir(path):
    """Read a dataset directory in the reference's layout
    (``data_utils.py:231-249``)."""
    with open(os.path.join(path, 'data.npy'), 'rb') as f:
        stock_paths = np.load(f)
        observed_dates = np.load(f)
        nb_obs = np.load(f)
    with open(os.path.join(path, 'metadata.txt'), 'r') as f:
        metadata = json.load(f)
    return stock_paths, observed_dates, nb_obs, metadata


class IrregularDataset(torch.utils.data.Dataset):
    """Index-able view of a dataset; items have the reference's keys
    (``data_utils.py:252-275``).  Built from in-memory arrays (or a dataset
    directory via ``from_dir``) instead of the reference's id registry."""

    def __init__(self, stock_paths, observed_dates, nb_obs, metadata, idx=None):
        if idx is None:
            idx = np.arange(len(nb_obs))
        self.metadata = metadata
        self.stock_paths = stock_paths[idx]
        self.observed_dates = observed_dates[idx]
        self.nb_obs = nb_obs[idx]

    @classmethod
    def from_dir(cls, path, idx=None):
        return cls(*load_dataset_dir(path), idx=idx)

    def __len__(self):
        return len(self.nb_obs)

    def __getitem__(self, idx):
        if isinstance(idx, (int, np.integer)):
            idx = [int(idx)]
        return {"idx": idx, "stock_path": self.stock_paths[idx],
                "observed_dates": self.observed_dates[idx],
                "nb_obs": self.nb_obs[idx], "dt": self.metadata['dt']}


def collate_arrays(stock_paths, observed_dates, nb_obs, dt, functions=()):
    """Vectorised CSR-by-time collate of a whole block of paths.

    :param stock_paths: f64 [B, d, S+1]
    :param observed_dates: {0,1} [B, S+1]
    :return: dict with the keys of the reference's collate
    """
    def lift(a, axis):
        out = a
        for f in functions:
            out = np.concatenate([out, f(a)], axis=axis)
        return out

    n_grid = observed_dates.shape[1]
    # the reference accumulates the clock in float64: t_k = dt + dt + ... (k x)
    clock = np.cumsum(np.full(n_grid - 1, dt, dtype=np.float64))
    mask_t = observed_dates[:, 1:].T == 1            # [S, B]
    per_time = mask_t.sum(axis=1)
    used = per_time > 0
    t_idx, b_idx = np.nonzero(mask_t)                # sorted by time, then path
    times = clock[used]
    time_ptr = np.concatenate([[0], np.cumsum(per_time[used])]).astype(np.int64)
    X = lift(stock_paths[b_idx, :, t_idx + 1], axis=1)
    X = X.reshape(len(b_idx), -1)
    start_X = lift(stock_paths[:, :, 0], axis=1)
    return {'times': times, 'time_ptr': time_ptr,
            'obs_idx': torch.tensor(b_idx, dtype=torch.long),
            'start_X': torch.tensor(start_X, dtype=torch.float32),
            'n_obs_ot': torch.tensor(nb_obs),
            'X': torch.tensor(X, dtype=torch.float32),
            'true_paths': stock_paths, 'observed_dates': observed_dates}


def _collate(batch, functions=()):
    dt = batch[0]['dt']
    stock_paths = np.concatenate([b['stock_path'] for b in batch], axis=0)
    observed_dates = np.concatenate([b['observed_dates'] for b in batch],
                                    axis=0)
    nb_obs = np.concatenate([b['nb_obs'] for b in batch], axis=0)
    return collate_arrays(stock_paths, observed_dates, nb_obs, dt, functions)


def custom_collate_fn(batch):
    """``torch.utils.data.DataLoader`` collate (reference
    ``data_utils.py:278-316``)."""
    return _collate(batch)


def _get_func(name):
    """'exp' / 'power-x' -> numpy function (reference ``data_utils.py:319-335``)."""
    if name in ['exp', 'exponential']:
        return np.exp
    if 'power-' in name:
        x = float(name.split('-')[1])
        return lambda a: np.power(a, x)
    return None


def CustomCollateFnGen(func_names=None):
    """Collate that appends f(X) for each named function as extra data
    dimensions (reference ``data_utils.py:352-416``).

    :return: (collate function, dimension multiplier)
    """
    functions = []
    for name in (func_names or []):
        f = _get_func(name)
        if f is not None:
            functions.append(f)

    def collate(batch):
        return _collate(batch, tuple(functions))

    return collate, len(functions) + 1


def recount_observations(obs_idx, batch_size):
    """``n_obs_ot`` as the training loop recomputes it per batch: the number of
    rows of ``obs_idx`` per path (reference ``train.py:501-507``)."""
    counts = torch.bincount(obs_idx.reshape(-1), minlength=batch_size)
    return counts.to(torch.int64)
