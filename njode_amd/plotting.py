"""
Host-side plot of the reference harness (``NJODE/train.py:673-796``,
``plot_one_path_with_pred``): for chosen paths of a batch, the true path, its observed
points, the model's predicted conditional expectation (``model.get_pred``), the true
conditional expectation of the data-generating process and -- when the model was trained on
``func_appl_X=['power-2']`` -- a band of ``std_factor`` predicted conditional standard
deviations.  Same signature and return value (the optimal loss) as the reference; the batch
is what ``data_utils.custom_collate_fn`` returns plus ``true_paths`` / ``observed_dates``.
No GPU code here: matplotlib on the CPU (backend Agg unless one is already chosen).
"""
import os

import numpy as np


def _to_numpy(x):
    return x.detach().cpu().numpy() if hasattr(x, 'detach') else np.asarray(x)


def plot_one_path_with_pred(device, model, batch, stockmodel, delta_t, T, path_to_plot=(0,),
                            save_path='', filename='plot_{}.pdf', plot_variance=False,
                            functions=None, std_factor=1, model_name=None, ylabels=None,
                            save_extras={'bbox_inches': 'tight', 'pad_inches': 0.01}):
    import matplotlib
    if not os.environ.get('MPLBACKEND'):
        matplotlib.use('Agg', force=False)
    import matplotlib.colors
    import matplotlib.pyplot as plt

    if model_name is None or model_name == 'NJODE':
        model_name = 'our model'
    cycle = plt.rcParams['axes.prop_cycle'].by_key()['color']
    band_color = tuple(matplotlib.colors.to_rgb(cycle[1])) + (0.5,)
    if save_path:
        os.makedirs(save_path, exist_ok=True)

    times, time_ptr = batch['times'], batch['time_ptr']
    X, start_X = batch['X'].to(device), batch['start_X'].to(device)
    obs_idx, n_obs_ot = batch['obs_idx'], batch['n_obs_ot']
    true_X = np.asarray(batch['true_paths'])            # [B, d, S+1]
    observed = np.asarray(batch['observed_dates'])      # [B, S+1]
    grid = np.linspace(0., T, int(np.round(T / delta_t)) + 1)
    dim = true_X.shape[1]

    model.eval()
    res = model.get_pred(times, time_ptr, X, obs_idx, delta_t, T, start_X)
    pred, pred_t = _to_numpy(res['pred']), np.asarray(res['pred_t'])

    std = None
    if plot_variance and functions is not None and 'power-2' in functions:
        k = list(functions).index('power-2') + 1        # block of the squared coordinates
        var = pred[:, :, dim * k:dim * (k + 1)] - pred[:, :, :dim] ** 2
        if np.any(var < 0):
            print('WARNING: some predicted cond. variances below 0 -> clip')
        std = np.sqrt(np.maximum(var, 0.0))

    opt_loss, true_t, true_y = stockmodel.compute_cond_exp(
        times, time_ptr, _to_numpy(X), _to_numpy(obs_idx), delta_t, T, _to_numpy(start_X),
        _to_numpy(n_obs_ot), return_path=True, get_loss=True, weight=model.weight)

    for i in path_to_plot:
        seen = np.nonzero(observed[i] == 1)[0]
        seen = seen[seen > 0] if observed[i][0] == 1 else seen
        t_obs = np.concatenate([[0.], grid[seen]])
        x_obs = np.concatenate([true_X[i, :, :1], true_X[i][:, seen]], axis=1)   # [d, n]
        fig, axs = plt.subplots(dim)
        axs = [axs] if dim == 1 else list(axs)
        for j, ax in enumerate(axs):
            ax.plot(grid, true_X[i, j, :], label='true path', color=cycle[0])
            ax.scatter(t_obs, x_obs[j], label='observed', color=cycle[0])
            ax.plot(pred_t, pred[:, i, j], label=model_name, color=cycle[1])
            if std is not None:
                ax.fill_between(pred_t, pred[:, i, j] - std_factor * std[:, i, j],
                                pred[:, i, j] + std_factor * std[:, i, j], color=band_color)
            ax.plot(true_t, true_y[:, i, j], label='true conditional expectation', linestyle=':',
                    color=cycle[2])
            if ylabels:
                ax.set_ylabel(ylabels[j])
        plt.legend()
        plt.xlabel('$t$')
        plt.savefig(os.path.join(save_path, filename.format(i)), **save_extras)
        plt.close(fig)
    return opt_loss
