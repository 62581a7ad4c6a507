"""
PhysioNet evaluation protocol of the reference (``NJODE/physionet_train.py:411-510``) for the
masked NJ-ODE path: observe the first half of a batch's time axis, predict the second half,
score the prediction at the held-out observation times.

The real PhysioNet-2012 records cannot be downloaded here (no network), so the batches come
from ``synthetic_physionet`` in the layout the reference's test-mode collate produces
(``latent_ODE/physionet_LODE.py:483-541``, ``data_type == "test"``):

* ``times`` / ``time_ptr`` / ``X`` / ``M`` / ``obs_idx``: rows of the FIRST ``len(times) // 2``
  union times only;
* ``times_val [T2]``, ``vals_val [B, T2, d]``, ``mask_val [B, T2, d]``: the held-out half,
  dense.

``evaluate_model`` returns the reference's triple ``(loss_val, mse_val, mse_val_2)``:
mean NJ-ODE loss per batch, masked MSE over all held-out observations, and the latent-ODE
style MSE (mean over patients of the mean over attributes of the per-attribute MSE,
``latent_ODE/likelihood_eval_LODE.py:171-193,229-236``).
"""
import numpy as np
import torch

from . import synthetic_physionet


def get_comparison_times_ind(path_t, times_val):
    """Indices into ``path_t`` whose entries are closest to the entries of ``times_val``
    (``physionet_train.py:473-505``): for t in [path_t[i], path_t[i+1]) the nearer of the two
    (ties to the left); past the last interval the last index.  Vectorised; path_t is
    non-decreasing (jumps repeat a time: the FIRST matching interval wins, like the reference's
    loop, which breaks at the first hit)."""
    path_t = np.asarray(path_t, dtype=np.float64)
    times_val = np.asarray(times_val, dtype=np.float64)
    assert np.min(path_t) < np.min(times_val) and np.max(path_t) + 1e-10 > np.max(times_val), \
        "mins: {}, {}, max: {}, {}".format(np.min(path_t), np.min(times_val), np.max(path_t),
                                           np.max(times_val))
    n = len(path_t)
    out = np.empty(len(times_val), dtype=np.int64)
    for j, t in enumerate(times_val):
        # first i in [0, n-2] with |path_t[i] - t| < 1e-10 or path_t[i] <= t < path_t[i+1]
        near = np.nonzero(np.abs(path_t[:n - 1] - t) < 1e-10)[0]
        inside = np.nonzero((path_t[:n - 1] <= t) & (t < path_t[1:]))[0]
        cands = [c[0] for c in (near, inside) if len(c)]
        if cands:
            i = min(cands)
            out[j] = i if abs(t - path_t[i]) <= abs(t - path_t[i + 1]) else i + 1
        else:
            out[j] = n - 1
    return out


def masked_mse_per_attribute(pred, vals, mask):
    """``torch.mean(compute_masked_likelihood(pred, vals, mask, mse))`` of the reference for one
    sample: per (patient, attribute) the mean squared error over that attribute's observed
    held-out times (0 if it has none), mean over attributes, mean over patients."""
    pred, vals, mask = (np.asarray(a, dtype=np.float32) for a in (pred, vals, mask))
    m = mask > 0
    cnt = m.sum(axis=1)                                        # [B, d]
    se = np.where(m, (pred - vals) ** 2, np.float32(0)).astype(np.float32)
    # nn.MSELoss on the selected entries = float32 mean
    per = np.where(cnt > 0, se.sum(axis=1, dtype=np.float32) / np.maximum(cnt, 1).astype(np.float32),
                   np.float32(0))
    return float(np.mean(np.mean(per, axis=-1, dtype=np.float32), dtype=np.float32))


def make_eval_batch(batch_size=50, dim=synthetic_physionet.PHYSIONET_DIM, n_grid=3000,
                    n_obs_range=(30, 100), p_feature=0.15, seed=0):
    """A synthetic batch in the reference's TEST layout: the full PhysioNet-shaped batch of
    ``synthetic_physionet.make_batch`` split at ``len(times) // 2`` into observed rows and the
    dense held-out half."""
    full = synthetic_physionet.make_batch(batch_size, dim, n_grid, n_obs_range, p_feature, seed,
                                          with_time_zero=False, n_empty_slices=0)
    times, time_ptr = full['times'], full['time_ptr']
    X, M, obs_idx = full['X'].numpy(), full['M'].numpy(), full['obs_idx'].numpy()
    n_t = len(times)
    vals = np.zeros((batch_size, n_t, dim), dtype=np.float32)
    mask = np.zeros((batch_size, n_t, dim), dtype=np.float32)
    for ti in range(n_t):
        rows = slice(time_ptr[ti], time_ptr[ti + 1])
        vals[obs_idx[rows], ti] = X[rows]
        mask[obs_idx[rows], ti] = M[rows]
    n_obs_t = n_t // 2
    n_rows = int(time_ptr[n_obs_t])
    return {
        'times': times[:n_obs_t], 'batch_size': batch_size,
        'time_ptr': np.asarray(time_ptr[:n_obs_t + 1]),
        'obs_idx': torch.tensor(obs_idx[:n_rows], dtype=torch.long),
        'X': torch.tensor(X[:n_rows]), 'M': torch.tensor(M[:n_rows]),
        'times_val': times[n_obs_t:], 'vals_val': vals[:, n_obs_t:], 'mask_val': mask[:, n_obs_t:],
        'delta_t': full['delta_t'], 'T': full['T'],
    }


def evaluate_model(model, batches, device, delta_t, T):
    """``physionet_train.evaluate_model`` (``:411-470``) for a list of test-layout batches."""
    with torch.no_grad():
        loss_val = 0.0
        num_obs = 0.0
        count = 0
        mse_val = 0.0
        mse_val_2 = 0.0
        model.eval()
        for b in batches:
            obs_idx = b['obs_idx']
            b_size = b['batch_size']
            X = b['X'].to(device)
            M = b['M'].to(device)
            n_obs_ot = torch.tensor(np.bincount(obs_idx.numpy(), minlength=b_size)).to(device)
            start_X = torch.zeros(b_size, X.shape[1], dtype=torch.float32, device=device)
            _, e_loss, path_t, _, path_y = model(
                b['times'], b['time_ptr'], X, obs_idx, delta_t, T, start_X, n_obs_ot,
                until_T=True, return_path=True, get_loss=True, M=M)
            ind = get_comparison_times_ind(path_t, b['times_val'])
            path_y = path_y.detach().cpu().numpy()[ind]
            path_y = np.transpose(path_y, (1, 0, 2))
            mse_val += float((((path_y - b['vals_val']) ** 2) * b['mask_val']).sum())
            loss_val += float(e_loss.detach().cpu())
            num_obs += float(b['mask_val'].sum())
            count += 1
            mse_val_2 += masked_mse_per_attribute(path_y, b['vals_val'], b['mask_val'])
        return loss_val / count, mse_val / num_obs, mse_val_2 / count
