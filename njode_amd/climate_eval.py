"""
Climate (USHCN) evaluation protocol of the reference (``NJODE/climate_train.py:508-566`` +
``GRU_ODE_Bayes/data_utils_gru_ode_bayes.py:379-408``) for the masked NJ-ODE path: observe a
station's measurements up to ``T_val``, run the model to ``T`` and score its prediction at up to
``max_val_samples`` later measurement times of that station.

The USHCN csv is not part of the reference checkout (``.MISSING_LARGE_BLOBS``) and there is no
network, so the batches come from a seeded synthetic stand-in in the layout the reference's
climate collate (``data_utils_gru_ode_bayes.custom_collate_fn``, ``:235-300``) produces:

* ``times`` / ``time_ptr`` / ``X`` / ``M`` / ``obs_idx``: rows of the observed part (time <=
  ``T_val``), sorted by time, values zero where unobserved, 5 measurement channels;
* ``X_val [L, d]``, ``M_val [L, d]``, ``times_val [L]``, ``index_val [L]``: the held-out rows
  (time > ``T_val``), sorted by (station, time), ``index_val`` = position of the station in
  the batch;
* ``start_X = 0``, ``T = 200``, ``delta_t = 0.1`` (``climate_train.py:244-247,526-527``),
  ``val_options = {"T_val": 150, "max_val_samples": 3}`` (``:215``).

``evaluate_model`` returns the reference's pair ``(loss_val, mse_val)``: mean NJ-ODE loss per
batch and the masked MSE over all held-out measurements.
"""
import numpy as np
import torch

CLIMATE_DIM = 5
CLIMATE_T = 200
CLIMATE_DELTA_T = 0.1
CLIMATE_T_VAL = 150


def extract_from_path(t_vec, p_vec, eval_times, path_idx_eval):
    """Prediction at ``eval_times`` for paths ``path_idx_eval`` from a prediction path
    (reference ``data_utils_gru_ode_bayes.py:379-400``).  A time occurs twice in ``t_vec`` when
    a jump happened there; the FIRST occurrence -- the prediction before the update -- is used.
    Evaluation times that are not on the path are mapped to the closest path time (ties to the
    earlier one, like the reference's ``argmin``)."""
    t_vec = np.asarray(t_vec)
    t_unique, first = np.unique(t_vec, return_index=True)
    p_vec = p_vec[first, :, :]
    eval_times = np.asarray(eval_times, dtype=np.float64)
    dist = np.abs(t_unique[None, :].astype(np.float64) - eval_times[:, None])
    time_idx = dist.argmin(axis=1)
    return p_vec[time_idx, np.asarray(path_idx_eval), :]


def n_decimals(delta_t):
    """``str(delta_t)[::-1].find('.')`` of the reference (``climate_train.py:550-551``)."""
    return str(delta_t)[::-1].find('.')


def make_climate_batch(batch_size=100, dim=CLIMATE_DIM, T=CLIMATE_T, T_val=CLIMATE_T_VAL,
                       delta_t=CLIMATE_DELTA_T, n_obs_range=(20, 60), max_val_samples=3,
                       p_feature=0.4, seed=0):
    """Synthetic stand-in for one batch of the reference's validation loader."""
    rng = np.random.RandomState(seed)
    n_grid_val = int(round(T_val / delta_t))
    n_grid = int(round(T / delta_t))
    rows = []          # (grid index, station, x, m)
    val = []           # (station, grid index, x, m)
    for b in range(batch_size):
        n_t = rng.randint(n_obs_range[0], n_obs_range[1] + 1)
        ks = np.sort(rng.choice(np.arange(1, n_grid_val + 1), size=n_t, replace=False))
        for k in ks:
            m = rng.random_sample(dim) < p_feature
            if not m.any():
                m[rng.randint(dim)] = True
            rows.append((int(k), b, (rng.standard_normal(dim) * m).astype(np.float32),
                         m.astype(np.float32)))
        n_v = rng.randint(1, max_val_samples + 1)
        kv = np.sort(rng.choice(np.arange(n_grid_val + 1, n_grid), size=n_v, replace=False))
        for k in kv:
            m = rng.random_sample(dim) < p_feature
            if not m.any():
                m[rng.randint(dim)] = True
            val.append((b, int(k), (rng.standard_normal(dim) * m).astype(np.float32),
                        m.astype(np.float32)))
    rows.sort(key=lambda r: (r[0], r[1]))
    grid = np.array([r[0] for r in rows])
    ks, counts = np.unique(grid, return_counts=True)
    # the reference's Time column holds decimal numbers with one digit
    times = np.round(ks * delta_t, n_decimals(delta_t))
    return {
        'times': times, 'time_ptr': np.concatenate([[0], np.cumsum(counts)]).astype(np.int64),
        'X': torch.tensor(np.stack([r[2] for r in rows])),
        'M': torch.tensor(np.stack([r[3] for r in rows])),
        'obs_idx': torch.tensor(np.array([r[1] for r in rows], dtype=np.int64)),
        'pat_idx': list(range(batch_size)),
        'X_val': torch.tensor(np.stack([v[2] for v in val])),
        'M_val': torch.tensor(np.stack([v[3] for v in val])),
        'times_val': np.round(np.array([v[1] for v in val]) * delta_t, n_decimals(delta_t)),
        'index_val': np.array([v[0] for v in val], dtype=np.int64),
        'delta_t': delta_t, 'T': T,
    }


def evaluate_model(model, batches, device, delta_t, T):
    """``climate_train.evaluate_model`` (``:508-566``) for a list of validation batches."""
    with torch.no_grad():
        loss_val = 0.0
        num_obs = 0.0
        mse_val = 0.0
        model.eval()
        for b in batches:
            obs_idx = b['obs_idx']
            b_size = len(b['pat_idx'])
            X = b['X'].to(device)
            M = b['M'].to(device)
            n_obs_ot = torch.tensor(np.bincount(obs_idx.numpy(), minlength=b_size)).to(device)
            start_X = torch.zeros(b_size, X.shape[1], dtype=torch.float32, device=device)
            _, e_loss, path_t, _, path_y = model(
                b['times'], b['time_ptr'], X, obs_idx, delta_t, T, start_X, n_obs_ot,
                until_T=True, return_path=True, get_loss=True, M=M)
            # round the floating point error out of the time vector (reference :549-551)
            t_vec = np.around(path_t, n_decimals(delta_t)).astype(np.float32)
            p_val = extract_from_path(t_vec, path_y.detach().cpu().numpy(), b['times_val'],
                                      b['index_val'])
            X_val, M_val = b['X_val'].numpy(), b['M_val'].numpy()
            mse_val += float((((X_val - p_val) ** 2) * M_val).sum())
            loss_val += float(e_loss.detach().cpu())
            num_obs += float(M_val.sum())
        return loss_val / len(batches), mse_val / num_obs
