"""
Training / evaluation loop for the HIP-backed NJ-ODE model.

Reproduces the compute semantics of the reference harness ``NJODE/train.py`` for the
hot path -- nothing of its experiment management (model-id registry, plots, Telegram):

* data split ``train_test_split(arange(N), test_size, random_state=seed)``
  (``train.py:232-235``) -- implemented with numpy's RandomState-compatible
  permutation of sklearn's ShuffleSplit (same indices, asserted in the tests when
  sklearn is importable);
* per optimizer step (``train.py:491-523``): recount ``n_obs_ot`` from ``obs_idx``,
  ``model(...)`` in train mode, backward, ``Adam(lr, weight_decay=5e-4)``;
* per epoch (``train.py:527-574, 623-624``): eval loss on the whole validation set as
  one batch in eval mode, ``model.epoch += 1``, ``model.weight_decay_step()``;
  ``train_loss`` logged = loss of the last batch of the epoch;
* ``compute_optimal_eval_loss`` (``train.py:648-670``) from the analytic conditional
  expectation; optional ``evaluate`` mean-square distance (``train.py:562-566``);
* checkpoint / metric-file policy (``train.py:400-418, 585-621``) when ``model_path`` is
  given: ``metric_id-<id>.csv`` with the reference's columns, ``last_checkpoint/`` every
  ``save_every`` epochs and whenever the eval loss improves, ``best_checkpoint/`` on
  improvement, ``resume_training`` / ``load_best`` to continue from them.

The fused step (``NJODE.loss_and_grad`` + ``FusedAdam``) is used by default; pass
``fused=False`` to drive the model exactly like the reference does
(``loss.backward()`` + ``torch.optim.Adam``).  With torch.distributed initialised the
loop is data parallel (see ``parallel.py``).
"""
import csv
import os
import time

import numpy as np
import torch

from . import data_utils, device_data, models, parallel, stock_model

METR_COLUMNS = ['epoch', 'train_time', 'eval_time', 'train_loss', 'eval_loss',
                'optimal_eval_loss']


def split_indices(n, test_size=0.2, seed=398):
    """sklearn.model_selection.train_test_split(np.arange(n), test_size, random_state)
    without sklearn: ShuffleSplit draws one RandomState(seed).permutation(n); the first
    n_test entries are the test set, the next n_train the training set."""
    n_test = int(np.ceil(test_size * n))
    n_train = n - n_test
    perm = np.random.RandomState(seed).permutation(n)
    return perm[n_test:n_test + n_train], perm[:n_test]


def compute_optimal_eval_loss(batch, stockmodel, delta_t, T, weight=0.5):
    return stockmodel.get_optimal_loss(
        batch['times'], batch['time_ptr'], batch['X'].numpy(), batch['obs_idx'].numpy(),
        delta_t, T, batch['start_X'].numpy(), batch['n_obs_ot'].numpy(), weight=weight)


def _device_batch(b, device):
    return {'times': b['times'], 'time_ptr': b['time_ptr'], 'X': b['X'].to(device),
            'start_X': b['start_X'].to(device),
            'obs_idx': b['obs_idx'].to(device, torch.int32),
            'n_obs_ot': data_utils.recount_observations(
                b['obs_idx'], b['start_X'].shape[0]).to(device, torch.int32)}


def train(dataset_arrays, metadata, epochs=1, batch_size=100, learning_rate=1e-3,
          hidden_size=10, bias=True, dropout_rate=0.1,
          ode_nn=((50, 'tanh'), (50, 'tanh')), readout_nn=((50, 'tanh'), (50, 'tanh')),
          enc_nn=((50, 'tanh'), (50, 'tanh')), use_rnn=False, solver='euler', weight=0.5,
          weight_decay=1., test_size=0.2, seed=398, device='cuda', fused=True,
          evaluate=False, shuffle_seed=0, log=print, max_steps_per_epoch=None,
          device_collate=False, model_path=None, model_id=1, save_every=1,
          resume_training=False, load_best=False, plan_ahead=True, plan_ahead_min=0,
          init_state=None, **options):
    """Train on an in-memory dataset ``(stock_paths, observed_dates, nb_obs)`` with
    ``metadata`` as returned by ``data_utils.create_dataset``.  Returns
    ``(model, metrics)`` with one row of ``METR_COLUMNS`` per epoch.

    ``device_collate=True`` uploads the dataset once and builds every training batch with the
    GPU collate (``device_data.DeviceDataset``): same batches bit for bit, without the
    per-step host collate and host-to-device copies.  ``plan_ahead`` (fused loop): collate one
    batch ahead and, for local batches of 4 096 paths or more, build its execution plan beside
    the current step (same results bit for bit).  ``init_state``: a state_dict to start from
    instead of the seed-0 initialisation; ``dropout_seed`` (an option of the model) selects the
    dropout stream."""
    stock_paths, observed_dates, nb_obs = dataset_arrays
    delta_t, T = metadata['dt'], metadata['maturity']
    input_size = output_size = metadata['dimension']
    world, rank, local_rank = parallel.init_distributed() if torch.distributed.is_initialized() \
        else (1, 0, 0)
    functions = options.get('func_appl_X')
    funcs = tuple(f for f in (data_utils._get_func(n) for n in (functions or [])) if f)
    mult = len(funcs) + 1

    train_idx, val_idx = split_indices(len(nb_obs), test_size, seed)
    model_opts = dict(options)
    model_opts['device_outputs'] = True
    torch.manual_seed(0)
    model = models.NJODE(input_size * mult, hidden_size, output_size * mult, ode_nn,
                         readout_nn, enc_nn, use_rnn, bias=bias, dropout_rate=dropout_rate,
                         solver=solver, weight=weight, weight_decay=weight_decay,
                         options=model_opts).to(device)
    if init_state is not None:     # start from given weights (like-for-like runs against the reference)
        model.load_state_dict(init_state)
    parallel.broadcast_parameters_(model.flat_parameters())
    if fused:
        optimizer = models.FusedAdam(model, lr=learning_rate, weight_decay=0.0005,
                                     distributed=world > 1)
    else:
        optimizer = torch.optim.Adam(model.parameters(), lr=learning_rate, weight_decay=0.0005)

    val = data_utils.collate_arrays(stock_paths[val_idx], observed_dates[val_idx],
                                    nb_obs[val_idx], delta_t, funcs)
    stockmodel = stock_model.STOCK_MODELS[metadata['model_name']](**metadata)
    opt_eval_loss = compute_optimal_eval_loss(val, stockmodel, delta_t, T) if mult == 1 \
        else float('nan')
    val_d = _device_batch(val, device)
    val_d['n_obs_ot'] = val['n_obs_ot'].to(device, torch.int32)   # eval uses the dataset's count

    dev_ds = device_data.DeviceDataset.from_arrays(stock_paths, observed_dates, nb_obs, metadata,
                                                   device) if device_collate else None
    # checkpoints and the metric file (reference train.py:344-350, 400-418)
    columns = METR_COLUMNS + (['evaluation_mean_diff'] if evaluate else [])
    best_eval_loss = np.inf
    path_last = path_best = metric_file = None
    saved_rows = []
    if model_path is not None:
        model_dir = os.path.join(model_path, 'id-{}'.format(model_id))
        path_last = os.path.join(model_dir, 'last_checkpoint')
        path_best = os.path.join(model_dir, 'best_checkpoint')
        metric_file = os.path.join(model_dir, 'metric_id-{}.csv'.format(model_id))
        os.makedirs(model_dir, exist_ok=True)
        if resume_training and os.path.exists(os.path.join(
                path_best if load_best else path_last, 'checkpt.tar')):
            models.get_ckpt_model(path_best if load_best else path_last, model, optimizer, device)
            if os.path.exists(metric_file):
                with open(metric_file) as f:
                    saved_rows = [[float(x) for x in r[1:]] for r in list(csv.reader(f))[1:]]
                if saved_rows:
                    best_eval_loss = min(r[4] for r in saved_rows)
            model.epoch += 1
            model.weight_decay_step()

    def write_metrics():
        # pandas' DataFrame.to_csv layout: unnamed index column first
        with open(metric_file, 'w', newline='') as f:
            wr = csv.writer(f)
            wr.writerow([''] + columns)
            for i, r in enumerate(saved_rows):
                wr.writerow([i] + r)

    def plan_epoch(epoch):
        """Shuffled order of an epoch and (device collate) phase 1 of the collate for ALL of its
        batches: the count kernels enqueued back to back, ONE copy to the host, ONE wait
        (device_data.prepare_batches)."""
        order = train_idx[parallel.epoch_permutation(len(train_idx), epoch, shuffle_seed)]
        n_steps = (len(order) + batch_size - 1) // batch_size
        if max_steps_per_epoch:
            n_steps = min(n_steps, max_steps_per_epoch)
        prepared = None
        if dev_ds is not None and n_steps > 0:
            shards = []
            for s_ in range(n_steps):
                idx_ = order[s_ * batch_size:(s_ + 1) * batch_size]
                lo_, hi_ = parallel.shard_range(len(idx_), world, rank)
                shards.append(idx_[lo_:hi_])
            live = [i for i, m_ in enumerate(shards) if len(m_)]
            prepared = [None] * n_steps
            # (a rank whose shard is empty in EVERY step of the epoch -- batch_size < world --
            # prepares nothing and still joins every step's all-reduce, parallel.empty_shard_step)
            if live:
                for i, pr in zip(live, dev_ds.prepare_batches([shards[i] for i in live])):
                    prepared[i] = pr
        return order, n_steps, prepared

    metrics = []
    pending = []
    epoch_plan = {}
    while model.epoch <= epochs:
        t0 = time.time()
        model.train()
        order, n_steps, prepared = epoch_plan.pop(model.epoch, None) or plan_epoch(model.epoch)
        loss = None
        def prepare(s):
            idx = order[s * batch_size:(s + 1) * batch_size]
            lo, hi = parallel.shard_range(len(idx), world, rank)
            mine = idx[lo:hi]
            if len(mine) == 0:
                d = dict(times=None, time_ptr=None, X=None, obs_idx=None, start_X=None,
                         n_obs_ot=None)
            elif dev_ds is not None:
                # phase 2 of the device collate (two launches, no host wait); phase 1 ran for
                # the whole epoch at once (prepared)
                d = dev_ds.fill_batch(prepared[s], func_names=functions or ())
            else:
                b = data_utils.collate_arrays(stock_paths[mine], observed_dates[mine],
                                              nb_obs[mine], delta_t, funcs)
                d = _device_batch(b, device)
            return idx, lo, mine, d

        # (NJODE_PLAN_AHEAD=0: A/B switch for the look-ahead)
        plan_ahead = plan_ahead and os.environ.get('NJODE_PLAN_AHEAD', '1') != '0'
        nxt = prepare(0) if n_steps > 0 else None
        for s in range(n_steps):
            idx, lo, mine, d = nxt
            # one batch ahead: the next batch is collated now, and (fused loop) its execution
            # plan is built on a helper stream beside this step (NJODE.prefetch_plan)
            nxt = prepare(s + 1) if s + 1 < n_steps else None
            # Where it pays (profiles/r03_harness_epochs.jsonl, epoch paths/s with / without):
            # B = 100: 316 k / 297 k, B = 200: 572 k / 524 k -- the step is a chain of ~5 us
            # launches and the plan's six are off it; B = 1 000: 1.83 M / 2.04 M -- the plan
            # kernels queued ahead delay the step's own; B >= 4 096: the plan (~0.1 ms) hides
            # beside the ODE kernels (bench.py: 1.19 -> 1.14 ms at 20 000 paths).
            n_next = len(nxt[2]) if nxt is not None else 0
            # (round 5: up to ~260 000 rows the plan rides inside this step's ODE-forward launch --
            # no helper stream, it pays at every size: NJODE.plan_defer_ok)
            if plan_ahead and fused and n_next >= max(plan_ahead_min, 1) and \
                    (n_next <= 512 or n_next >= 4096 or
                     model.plan_defer_ok(int(nxt[3]['time_ptr'][-1]))):
                dn = nxt[3]
                model.prefetch_plan(dn['times'], dn['time_ptr'], dn['X'], dn['obs_idx'], delta_t, T,
                                    dn['start_X'], dn['n_obs_ot'], need_hT=False)
            parallel.configure_model(model, len(idx), lo)
            args = (d['times'], d['time_ptr'], d['X'], d['obs_idx'], delta_t, T, d['start_X'],
                    d['n_obs_ot'])
            optimizer.zero_grad()
            if len(mine) == 0:
                # more ranks than paths in this (last, partial) batch: contribute a zero
                # gradient and still join the all-reduce and the optimizer step
                loss = parallel.empty_shard_step(model, fused)
            elif fused:
                _, loss = model.loss_and_grad(*args)
            else:
                _, loss = model(*args, return_path=False, get_loss=True)
                loss.backward()
            if not fused and world > 1:
                for p in model.parameters():
                    parallel.allreduce_flat_(p.grad)
            optimizer.step()
        # the next epoch's order and collate counts, queued behind this epoch's steps: its one
        # host wait is the end-of-epoch wait the loop needs anyway
        if dev_ds is not None and model.epoch + 1 <= epochs:
            epoch_plan[model.epoch + 1] = plan_epoch(model.epoch + 1)
        torch.cuda.synchronize()
        train_time = time.time() - t0

        t0 = time.time()
        parallel.configure_model(model, val['start_X'].shape[0], 0)
        model.dp_global_batch = None
        with torch.no_grad():
            model.eval()
            _, c_loss = model(val_d['times'], val_d['time_ptr'], val_d['X'], val_d['obs_idx'],
                              delta_t, T, val_d['start_X'], val_d['n_obs_ot'],
                              return_path=False, get_loss=True)
            loss_val = float(c_loss)
            row_extra = []
            if evaluate:
                row_extra = [model.evaluate(
                    val_d['times'], val_d['time_ptr'], val_d['X'], val['obs_idx'], delta_t, T,
                    val_d['start_X'], val['n_obs_ot'], stockmodel)]
        eval_time = time.time() - t0
        # (fused data-parallel steps all-reduce the loss WITH the gradient: it is global already)
        if loss is None:
            train_loss = float('nan')
        elif fused and world > 1:
            train_loss = float(loss)
        else:
            train_loss = float(parallel.allreduce_flat_(loss.detach().reshape(1).clone()))
        if rank == 0:
            log("epoch {}, weight={:.5f}, train-loss={:.5f}, optimal-eval-loss={:.5f}, "
                "eval-loss={:.5f}, ".format(model.epoch, model.weight, train_loss,
                                            opt_eval_loss, loss_val))
        row = [model.epoch, train_time, eval_time, train_loss, loss_val, opt_eval_loss] + row_extra
        metrics.append(row)
        pending.append(row)
        if model_path is not None and rank == 0:
            improved = loss_val < best_eval_loss
            if model.epoch % save_every == 0 or improved:
                saved_rows.extend(pending)
                pending = []
                write_metrics()
                models.save_checkpoint(model, optimizer, path_last, model.epoch)
            if improved:
                log('save new best model: last-best-loss: {:.5f}, new-best-loss: {:.5f}, '
                    'epoch: {}'.format(best_eval_loss, loss_val, model.epoch))
                models.save_checkpoint(model, optimizer, path_best, model.epoch)
                best_eval_loss = loss_val
        model.epoch += 1
        model.weight_decay_step()
    return model, metrics
