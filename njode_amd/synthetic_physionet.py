"""
PhysioNet-shaped synthetic batches for the masked NJ-ODE path (BASELINE config 5).

There is no network access, so the real PhysioNet-2012 records cannot be
downloaded; this generator reproduces the *layout* that the reference's
``latent_ODE/physionet_LODE.py:428-544`` (``variable_time_collate_fn1``) hands
to ``NJODE.forward`` through ``physionet_train.py:326-353``:

* the batch's time axis is the sorted union of all observation times (scaled to
  [0, 1]); one row of ``X`` / ``M`` per (time, patient) whose mask row has at
  least one observed feature; rows sorted by time then patient;
* ``X`` is zero where unobserved, ``M`` is the 0/1 feature mask;
* a time can be in the union while contributing **no rows** (``time_ptr`` has
  an empty slice) and the first time can be exactly ``t = 0.0`` (a jump with no
  preceding Euler step);
* ``start_X = 0 [B, d]``, ``T = 1 + 1e-12``, ``delta_t = 0.016 / 48``
  (3 000 Euler steps), ``options['masked'] = True``.

Recipe: per path ``n_t ~ U{lo..hi}`` distinct grid
times from a seeded ``RandomState``; per observation each feature is on with
probability ``p_feature`` (at least one on); values ``U[0,1) * mask``.
"""
import numpy as np
import torch

PHYSIONET_DELTA_T = 0.016 / 48
PHYSIONET_T = 1 + 1e-12
PHYSIONET_DIM = 41


def make_batch(batch_size=50, dim=PHYSIONET_DIM, n_grid=3000, n_obs_range=(30, 100),
               p_feature=0.15, seed=0, with_time_zero=True, n_empty_slices=2):
    """Build one masked batch.

    :param n_grid: number of Euler grid intervals on [0, 1]; observation times
            are multiples of ``1 / n_grid`` (use a small value for tests)
    :param with_time_zero: force path 0 to have an observation at t = 0.0
    :param n_empty_slices: number of extra union times that carry no rows
    :return: dict with times, time_ptr, X, M, obs_idx, start_X, n_obs_ot,
             delta_t, T
    """
    rng = np.random.RandomState(seed)
    lo, hi = n_obs_range
    hi = min(hi, n_grid)
    lo = min(lo, hi)
    observed = np.zeros((batch_size, n_grid + 1), dtype=bool)
    for b in range(batch_size):
        n_t = rng.randint(lo, hi + 1)
        first = 0 if (with_time_zero and b == 0) else 1
        ks = rng.choice(np.arange(first, n_grid + 1), size=n_t, replace=False)
        observed[b, ks] = True
        if with_time_zero and b == 0:
            observed[b, 0] = True
    union = observed.any(axis=0)
    # union times that no patient contributes a row to
    free = np.nonzero(~union)[0]
    free = free[free > 0]
    if n_empty_slices and len(free):
        extra = rng.choice(free, size=min(n_empty_slices, len(free)),
                           replace=False)
        union[extra] = True
    grid_idx = np.nonzero(union)[0]
    times = grid_idx.astype(np.float64) / n_grid

    rows_x, rows_m, obs_idx, time_ptr = [], [], [], [0]
    for k in grid_idx:
        who = np.nonzero(observed[:, k])[0]
        for b in who:
            m = rng.random_sample(dim) < p_feature
            if not m.any():
                m[rng.randint(dim)] = True
            rows_m.append(m.astype(np.float32))
            rows_x.append((rng.random_sample(dim) * m).astype(np.float32))
            obs_idx.append(b)
        time_ptr.append(len(obs_idx))
    obs_idx = np.asarray(obs_idx, dtype=np.int64)
    n_obs_ot = np.bincount(obs_idx, minlength=batch_size).astype(np.int64)
    return {
        'times': times,
        'time_ptr': np.asarray(time_ptr, dtype=np.int64),
        'X': torch.tensor(np.stack(rows_x)),
        'M': torch.tensor(np.stack(rows_m)),
        'obs_idx': torch.tensor(obs_idx, dtype=torch.long),
        'start_X': torch.zeros(batch_size, dim),
        'n_obs_ot': torch.tensor(n_obs_ot),
        'delta_t': 1.0 / n_grid if n_grid != 3000 else PHYSIONET_DELTA_T,
        'T': PHYSIONET_T,
    }
