"""
Generate the golden vectors under tests/golden/ by running the REFERENCE itself.

Runs only in the build container (needs /root/reference; the reference never
travels to the GPU box).  The committed ``*.npz`` files hold data only: inputs
(seeded recipes spelled out below), the reference's parameters, and the
reference's outputs.  Usage:  python tests/golden/make_golden.py

Cases:
  g1_bs_eval_B{7,64,200}  eval forward, demo.py config, until_T path
  g2_bs_grads_B64         dropout 0, train(): loss, 18 grads, Adam x1 / x5
  g3_ckpt_{BS,Heston,OU}  shipped checkpoints, N=200 seed-0 data, 3 known answers
  g4_data_{BS,OU,Heston}  N=50 seed-0 datasets (bit-exact f64) + cond. exp.
  g5_masked               PhysioNet-shaped masked batch, loss/path/grads
  g6_*                    variants: input_current_t, easy loss, no residual,
                          func_appl_X, use_rnn, sparse times (B=5), empty slice
  g7 (round 2; `python tests/golden/make_golden.py g7` makes only these):
  g5_full                 PhysioNet-shaped masked batch at FULL length: B = 8, d = H = 41,
                          3 000 Euler steps (physionet_train.py:93,326-353), loss / path / grads
  g6_w{10,40}             network widths of the convergence study (parallel_train.py:304-305)
  ds_ref_BS/              a dataset directory written by the reference's create_dataset
                          (data_utils.py:56-105): data.npy + metadata.txt, 12 paths
  g8_physionet_eval       physionet_train.evaluate_model protocol (:411-510) on a synthetic
                          stand-in: observe the first half, predict the second half
  g9_ref_training_curves  (round 3) the shipped metric_id-{1,2,3}.csv columns: what a training
                          run has to track
  g8_climate_eval (g11, round 3)  climate_train.evaluate_model protocol (:508-566) on a synthetic
                          stand-in: observe up to T_val, predict the held-out measurements
  g9b_ref_seeded_curves (g12, round 4)  the REFERENCE model trained like-for-like with the build's
                          harness: same seed-0 20 000-path datasets, split 398, batch 200, the
                          same shuffled order per epoch, the same initial weights, Adam lr 1e-3 /
                          wd 5e-4, dropout 0.1 drawn from three torch seeds; eval loss per epoch
  g10 (round 3): g6_w100, g6_none_h50, g6_mixed_nets, g5_w200, g5_climate -- shapes that run on
                          the shape-generic kernels (widths >= 64, nn_desc = None with H = 50,
                          per-network descriptions, the climate shape)
"""
import contextlib
import copy
import io
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, '/root/reference')

with contextlib.redirect_stdout(io.StringIO()):
    import NJODE.models as ref_models
    import NJODE.data_utils as ref_data
    import NJODE.stock_model as ref_stock

from njode_amd import synthetic_physionet  # noqa: E402

NN = ((50, 'tanh'), (50, 'tanh'))


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def ref_dataset(name, n_paths, seed=0):
    """reference create_dataset protocol without the file bookkeeping
    (data_utils.py:73-81)."""
    hp = copy.deepcopy(ref_data.hyperparam_default)
    hp['nb_paths'] = n_paths
    hp['model_name'] = name
    np.random.seed(seed)
    sm = ref_stock.STOCK_MODELS[name](**hp)
    paths, dt = sm.generate_paths()
    obs = (np.random.random(size=(paths.shape[0], paths.shape[2])) <
           hp['obs_perc']) * 1
    nb_obs = np.sum(obs[:, 1:], axis=1)
    hp['dt'] = dt
    return paths, obs, nb_obs, hp, sm


def ref_collate(paths, obs, nb_obs, dt, idx, func_names=None):
    items = [{'idx': [i], 'stock_path': paths[[i]], 'observed_dates': obs[[i]],
              'nb_obs': nb_obs[[i]], 'dt': dt} for i in idx]
    if func_names is None:
        return ref_data.custom_collate_fn(items)
    fn, _ = ref_data.CustomCollateFnGen(func_names)
    return fn(items)


def build(cfg, seed=0):
    torch.manual_seed(seed)
    return quiet(ref_models.NJODE, **cfg)


def sd_arrays(model):
    return {'sd/' + k: v.detach().numpy().copy()
            for k, v in model.state_dict().items()}


def batch_arrays(b, with_M=False):
    out = {'times': np.asarray(b['times'], dtype=np.float64),
           'time_ptr': np.asarray(b['time_ptr'], dtype=np.int64),
           'X': b['X'].numpy(), 'obs_idx': b['obs_idx'].numpy(),
           'start_X': b['start_X'].numpy(),
           'n_obs_ot': np.asarray(b['n_obs_ot'])}
    if with_M:
        out['M'] = b['M'].numpy()
    return out


def save(name, cfg, arrays):
    arrays = dict(arrays)
    arrays['cfg_json'] = np.array(json.dumps(cfg))
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **arrays)
    print('{:28s} {:8.1f} KB'.format(name, os.path.getsize(path) / 1024))


def demo_cfg(d=1, H=10, dropout=0.1, **options):
    return dict(input_size=d, hidden_size=H, output_size=d, ode_nn=NN,
                readout_nn=NN, enc_nn=NN, use_rnn=False, bias=True,
                dropout_rate=dropout, options=options)


def eval_outputs(model, b, delta_t, T, M=None, until_T=True):
    model.eval()
    with torch.no_grad():
        hT, loss, path_t, path_h, path_y = model(
            b['times'], b['time_ptr'], b['X'], b['obs_idx'], delta_t, T,
            b['start_X'], b['n_obs_ot'], return_path=True, get_loss=True,
            until_T=until_T, M=M)
    return {'hT': hT.numpy(), 'loss': np.float64(loss.item()),
            'path_t': np.asarray(path_t, dtype=np.float64),
            'path_h': path_h.numpy(), 'path_y': path_y.numpy()}


def grad_outputs(model, b, delta_t, T, M=None):
    model.train()
    model.zero_grad()
    hT, loss = model(b['times'], b['time_ptr'], b['X'], b['obs_idx'], delta_t,
                     T, b['start_X'], b['n_obs_ot'], return_path=False,
                     get_loss=True, M=M)
    loss.backward()
    out = {'train_loss': np.float64(loss.item()), 'train_hT': hT.detach().numpy()}
    for k, p in model.named_parameters():
        out['grad/' + k] = p.grad.numpy().copy()
    return out


def g1():
    paths, obs, nb_obs, hp, _ = ref_dataset('BlackScholes', 200)
    for B in (7, 64, 200):
        cfg = demo_cfg()
        model = build(cfg)
        b = ref_collate(paths, obs, nb_obs, hp['dt'], range(B))
        arrays = {**sd_arrays(model), **batch_arrays(b),
                  'delta_t': hp['dt'], 'T': hp['maturity']}
        out = eval_outputs(model, b, hp['dt'], hp['maturity'])
        if B == 200:
            out.pop('path_h')
        arrays.update(out)
        # training-style call: no until_T, no path
        model.eval()
        with torch.no_grad():
            hT2, loss2 = model(b['times'], b['time_ptr'], b['X'], b['obs_idx'],
                               hp['dt'], hp['maturity'], b['start_X'],
                               b['n_obs_ot'])
        arrays['hT_lastobs'] = hT2.numpy()
        arrays['loss_lastobs'] = np.float64(loss2.item())
        save('g1_bs_eval_B{}'.format(B), cfg, arrays)


def g2():
    paths, obs, nb_obs, hp, _ = ref_dataset('BlackScholes', 200)
    cfg = demo_cfg(dropout=0.0)
    model = build(cfg)
    b = ref_collate(paths, obs, nb_obs, hp['dt'], range(64))
    arrays = {**sd_arrays(model), **batch_arrays(b),
              'delta_t': hp['dt'], 'T': hp['maturity']}
    arrays.update(grad_outputs(model, b, hp['dt'], hp['maturity']))
    # Adam(lr=1e-3, weight_decay=5e-4) x1 and x5 on the same batch
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=0.0005)
    model.train()
    losses = []
    for step in range(1, 6):
        opt.zero_grad()
        _, loss = model(b['times'], b['time_ptr'], b['X'], b['obs_idx'],
                        hp['dt'], hp['maturity'], b['start_X'], b['n_obs_ot'])
        loss.backward()
        opt.step()
        losses.append(loss.item())
        if step in (1, 5):
            for k, v in model.state_dict().items():
                arrays['adam{}/{}'.format(step, k)] = v.numpy().copy()
    arrays['adam_losses'] = np.asarray(losses, dtype=np.float64)
    save('g2_bs_grads_B64', cfg, arrays)


def g3():
    for mid, name, tag in ((1, 'BlackScholes', 'BS'), (2, 'Heston', 'Heston'),
                           (3, 'OrnsteinUhlenbeck', 'OU')):
        ck = torch.load('/root/reference/data/saved_models/id-{}/'
                        'last_checkpoint/checkpt.tar'.format(mid),
                        weights_only=False)
        cfg = demo_cfg(which_loss='standard', residual_enc_dec=True)
        model = build(cfg)
        model.load_state_dict(ck['model_state_dict'])
        model.weight = ck['weight']
        paths, obs, nb_obs, hp, sm = ref_dataset(name, 200)
        b = ref_collate(paths, obs, nb_obs, hp['dt'], range(200))
        dt, T = hp['dt'], hp['maturity']
        model.eval()
        with torch.no_grad():
            _, loss = model(b['times'], b['time_ptr'], b['X'], b['obs_idx'], dt,
                            T, b['start_X'], b['n_obs_ot'])
            msd = model.evaluate(b['times'], b['time_ptr'], b['X'],
                                 b['obs_idx'], dt, T, b['start_X'],
                                 b['n_obs_ot'], sm)
        opt = sm.get_optimal_loss(b['times'], b['time_ptr'], b['X'].numpy(),
                                  b['obs_idx'].numpy(), dt, T,
                                  b['start_X'].numpy(), b['n_obs_ot'].numpy(),
                                  weight=model.weight)
        arrays = {**sd_arrays(model), 'dataset': np.array(name),
                  'ckpt_epoch': ck['epoch'], 'ckpt_weight': ck['weight'],
                  'eval_loss': np.float64(loss.item()),
                  'optimal_loss': np.float64(opt), 'msd_cond_exp': np.float64(msd)}
        print('   ', tag, loss.item(), opt, msd)
        save('g3_ckpt_{}'.format(tag), cfg, arrays)


def g4():
    for name, tag in (('BlackScholes', 'BS'), ('OrnsteinUhlenbeck', 'OU'),
                      ('Heston', 'Heston')):
        paths, obs, nb_obs, hp, sm = ref_dataset(name, 50)
        b = ref_collate(paths, obs, nb_obs, hp['dt'], range(50))
        loss, ct, cy = sm.compute_cond_exp(
            b['times'], b['time_ptr'], b['X'].numpy(), b['obs_idx'].numpy(),
            hp['dt'], hp['maturity'], b['start_X'].numpy(),
            b['n_obs_ot'].numpy(), return_path=True, get_loss=True)
        arrays = {'paths': paths, 'observed_dates': obs, 'nb_obs': nb_obs,
                  **batch_arrays(b), 'delta_t': hp['dt'], 'T': hp['maturity'],
                  'cond_loss': np.float64(loss), 'cond_t': ct, 'cond_y': cy}
        hp_json = {k: v for k, v in hp.items()}
        save('g4_data_{}'.format(tag), hp_json, arrays)


def g5():
    b = synthetic_physionet.make_batch(batch_size=8, n_grid=150,
                                       n_obs_range=(5, 20), seed=0)
    cfg = dict(input_size=41, hidden_size=41, output_size=41, ode_nn=NN,
               readout_nn=NN, enc_nn=NN, use_rnn=False, bias=True,
               dropout_rate=0.0, options={'masked': True})
    model = build(cfg)
    arrays = {**sd_arrays(model), **batch_arrays(b, with_M=True),
              'delta_t': b['delta_t'], 'T': b['T']}
    arrays.update(eval_outputs(model, b, b['delta_t'], b['T'], M=b['M']))
    arrays.update(grad_outputs(model, b, b['delta_t'], b['T'], M=b['M']))
    save('g5_masked', cfg, arrays)


def g6():
    paths, obs, nb_obs, hp, _ = ref_dataset('BlackScholes', 200)
    dt, T = hp['dt'], hp['maturity']
    variants = {
        'g6_current_t': (demo_cfg(dropout=0.0, input_current_t=True), 16, None),
        'g6_easy_loss': (demo_cfg(dropout=0.0, which_loss='easy'), 16, None),
        'g6_no_residual': (demo_cfg(dropout=0.0, residual_enc_dec=False), 16,
                           None),
        'g6_power2': (demo_cfg(d=2, dropout=0.0), 16, ['power-2']),
        'g6_sparse_B5': (demo_cfg(dropout=0.0), 5, None),
        'g6_weight075': (dict(demo_cfg(dropout=0.0), weight=0.75), 16, None),
        'g6_relu_w20': (dict(demo_cfg(dropout=0.0),
                             ode_nn=((20, 'relu'), (20, 'relu')),
                             enc_nn=((20, 'relu'), (20, 'relu')),
                             readout_nn=((20, 'relu'), (20, 'relu'))), 16, None),
        'g6_linear_nets': (dict(demo_cfg(dropout=0.0), ode_nn=None, enc_nn=None,
                                readout_nn=None), 16, None),
    }
    for name, (cfg, B, funcs) in variants.items():
        model = build(cfg)
        b = ref_collate(paths, obs, nb_obs, dt, range(B), funcs)
        arrays = {**sd_arrays(model), **batch_arrays(b), 'delta_t': dt, 'T': T}
        arrays.update(eval_outputs(model, b, dt, T))
        arrays.update(grad_outputs(model, b, dt, T))
        save(name, cfg, arrays)

    # GRU jump
    cfg = dict(demo_cfg(dropout=0.0), use_rnn=True)
    model = build(cfg)
    b = ref_collate(paths, obs, nb_obs, dt, range(16))
    arrays = {**sd_arrays(model), **batch_arrays(b), 'delta_t': dt, 'T': T}
    arrays.update(eval_outputs(model, b, dt, T))
    arrays.update(grad_outputs(model, b, dt, T))
    save('g6_use_rnn', cfg, arrays)

    # off-grid delta_t: Euler step 0.004 against observations on the 0.01 grid
    # => 2 full steps + 1 partial step per grid interval, and an until_T tail
    cfg = demo_cfg(dropout=0.0)
    model = build(cfg)
    b = ref_collate(paths, obs, nb_obs, dt, range(6))
    arrays = {**sd_arrays(model), **batch_arrays(b), 'delta_t': 0.004, 'T': 1.05}
    arrays.update(eval_outputs(model, b, 0.004, 1.05))
    arrays.update(grad_outputs(model, b, 0.004, 1.05))
    save('g6_offgrid_dt', cfg, arrays)


def g7():
    # ---- config 5 at its real length: 3 000 Euler steps, 30..100 observation times per path
    b = synthetic_physionet.make_batch(batch_size=8, n_grid=3000, n_obs_range=(30, 100), seed=1)
    cfg = dict(input_size=41, hidden_size=41, output_size=41, ode_nn=NN,
               readout_nn=NN, enc_nn=NN, use_rnn=False, bias=True,
               dropout_rate=0.0, options={'masked': True})
    model = build(cfg)
    arrays = {**sd_arrays(model), **batch_arrays(b, with_M=True),
              'delta_t': b['delta_t'], 'T': b['T']}
    out = eval_outputs(model, b, b['delta_t'], b['T'], M=b['M'])
    out.pop('path_h')
    n_rows = out['path_y'].shape[0]
    rows = np.unique(np.concatenate([np.arange(0, n_rows, 40), [n_rows - 1]]))
    out['path_rows'] = rows.astype(np.int64)          # the stored rows of the path
    out['path_y'] = out['path_y'][rows]
    arrays.update(out)
    arrays.update(grad_outputs(model, b, b['delta_t'], b['T'], M=b['M']))
    save('g5_full', cfg, arrays)

    # ---- widths of the convergence study that the matrix-core kernels cover
    paths, obs, nb_obs, hp, _ = ref_dataset('BlackScholes', 200)
    dt, T = hp['dt'], hp['maturity']
    for w in (10, 40):
        nn = ((w, 'tanh'), (w, 'tanh'))
        cfg = dict(demo_cfg(dropout=0.0), ode_nn=nn, enc_nn=nn, readout_nn=nn)
        model = build(cfg)
        b = ref_collate(paths, obs, nb_obs, dt, range(20))   # the study's batch size
        arrays = {**sd_arrays(model), **batch_arrays(b), 'delta_t': dt, 'T': T}
        arrays.update(eval_outputs(model, b, dt, T))
        arrays.update(grad_outputs(model, b, dt, T))
        save('g6_w{}'.format(w), cfg, arrays)

    # ---- a dataset directory exactly as the reference writes it (f2)
    import shutil
    import tempfile
    tmp = tempfile.mkdtemp()
    cwd = os.getcwd()
    try:
        os.makedirs(os.path.join(tmp, 'a', 'b'))
        os.chdir(os.path.join(tmp, 'a', 'b'))            # the reference writes to ../data/...
        old = ref_data.training_data_path
        ref_data.training_data_path = os.path.join(tmp, 'data', 'training_data') + '/'
        os.makedirs(ref_data.training_data_path)
        hp2 = copy.deepcopy(ref_data.hyperparam_default)
        hp2['nb_paths'] = 12
        path, time_id = quiet(ref_data.create_dataset, 'BlackScholes', hp2, seed=0)
        dst = os.path.join(HERE, 'ds_ref_BS')
        if os.path.exists(dst):
            shutil.rmtree(dst)
        shutil.copytree(path, dst)
        ref_data.training_data_path = old
        print('ds_ref_BS/ <-', path, os.listdir(dst))
    finally:
        os.chdir(cwd)
        shutil.rmtree(tmp)


def _reference_physionet_functions():
    """evaluate_model / get_comparison_times_ind of NJODE/physionet_train.py, executed from the
    reference's own source (the module itself cannot be imported here: its imports need
    torchvision / telegram)."""
    import ast
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        'likelihood_eval_LODE', '/root/reference/latent_ODE/likelihood_eval_LODE.py')
    sys.path.insert(0, '/root/reference/latent_ODE')
    sys.path.insert(0, '/root/reference')
    try:
        likelihood_eval = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(likelihood_eval)
    except Exception as e:      # its helper imports are not available: restate the two callees
        print('   (likelihood_eval_LODE not importable: {}; using its two functions only)'.format(e))
        src = open('/root/reference/latent_ODE/likelihood_eval_LODE.py').read()
        tree = ast.parse(src)
        keep = [n for n in tree.body if isinstance(n, ast.FunctionDef)
                and n.name in ('compute_masked_likelihood', 'mse')]
        ns = {'torch': torch, 'nn': torch.nn, 'np': np, 'get_device': lambda t: t.device}
        exec(compile(ast.Module(keep, []), 'likelihood_eval_LODE.py', 'exec'), ns)
        import types
        likelihood_eval = types.SimpleNamespace(**{k: ns[k] for k in ('compute_masked_likelihood', 'mse')})
    src = open('/root/reference/NJODE/physionet_train.py').read()
    tree = ast.parse(src)
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef)
            and n.name in ('evaluate_model', 'get_comparison_times_ind')]
    if not hasattr(np, 'int'):
        np.int = int            # the reference predates numpy 1.24
    ns = {'np': np, 'torch': torch, 'likelihood_eval': likelihood_eval}
    exec(compile(ast.Module(keep, []), 'physionet_train.py', 'exec'), ns)
    return ns['evaluate_model'], ns['get_comparison_times_ind']


def g8():
    from njode_amd import physionet_eval
    evaluate_model, get_ind = _reference_physionet_functions()
    cfg = dict(input_size=41, hidden_size=41, output_size=41, ode_nn=NN,
               readout_nn=NN, enc_nn=NN, use_rnn=False, bias=True,
               dropout_rate=0.0, options={'masked': True})
    model = build(cfg)
    batches = [physionet_eval.make_eval_batch(batch_size=6, n_grid=240, n_obs_range=(6, 16), seed=s)
               for s in (3, 4)]
    dt, T = batches[0]['delta_t'], batches[0]['T']
    loss_val, mse_val, mse_val_2 = quiet(evaluate_model, model, batches, 'cpu', {}, dt, T)
    arrays = {**sd_arrays(model), 'delta_t': dt, 'T': T, 'n_batches': len(batches),
              'loss_val': np.float64(loss_val), 'mse_val': np.float64(mse_val),
              'mse_val_2': np.float64(mse_val_2)}
    for i, b in enumerate(batches):
        for k in ('times', 'time_ptr', 'times_val', 'vals_val', 'mask_val'):
            arrays['b{}/{}'.format(i, k)] = np.asarray(b[k])
        for k in ('X', 'M', 'obs_idx'):
            arrays['b{}/{}'.format(i, k)] = b[k].numpy()
        arrays['b{}/batch_size'.format(i)] = b['batch_size']
        # comparison indices of the reference on this batch's prediction grid
        model.eval()
        with torch.no_grad():
            n_obs_ot = torch.tensor(np.bincount(b['obs_idx'].numpy(), minlength=b['batch_size']))
            _, _, path_t, _, _ = model(b['times'], b['time_ptr'], b['X'], b['obs_idx'], dt, T,
                                       torch.zeros(b['batch_size'], 41), n_obs_ot, until_T=True,
                                       return_path=True, get_loss=True, M=b['M'])
        arrays['b{}/path_t'.format(i)] = np.asarray(path_t, dtype=np.float64)
        arrays['b{}/cmp_ind'.format(i)] = np.asarray(get_ind(path_t, b['times_val']), dtype=np.int64)
    print('    physionet eval protocol:', loss_val, mse_val, mse_val_2)
    save('g8_physionet_eval', cfg, arrays)


def g9():
    """The training curves the reference ships with its three pre-trained models
    (data/saved_models/id-{1,2,3}/metric_id-N.csv: 200 epochs of train.py at batch 200, lr 1e-3,
    dropout 0.1 on the seed-0 20 000-path datasets, split seed 398): per epoch train_loss,
    eval_loss, optimal_eval_loss.  Data only -- the numbers a training run of the build has to
    track (tests/test_hip_convergence.py)."""
    import csv
    arrays = {}
    names = {1: 'BlackScholes', 2: 'Heston', 3: 'OrnsteinUhlenbeck'}
    for i, name in names.items():
        path = '/root/reference/data/saved_models/id-{0}/metric_id-{0}.csv'.format(i)
        with open(path) as f:
            rows = list(csv.reader(f))
        head = rows[0]
        cols = {c: np.array([float(r[head.index(c)]) for r in rows[1:]]) for c in
                ('epoch', 'train_loss', 'eval_loss', 'optimal_eval_loss')}
        for c, v in cols.items():
            arrays['{}/{}'.format(name, c)] = v
        print('    {:18s} epochs {:3d}  eval_loss[1,10,30,200] = {:.5f} {:.5f} {:.5f} {:.5f}  optimal {:.5f}'
              .format(name, len(cols['epoch']), cols['eval_loss'][0], cols['eval_loss'][9],
                      cols['eval_loss'][29], cols['eval_loss'][-1], cols['optimal_eval_loss'][0]))
    with open('/root/reference/data/saved_models/model_overview.csv') as f:
        desc = list(csv.reader(f))[1][2]
    save('g9_ref_training_curves', json.loads(desc), arrays)


def g12(names=('OrnsteinUhlenbeck', 'BlackScholes'), epochs=12, seeds=(0, 1, 2), batch_size=200):
    """Like-for-like training curves.  The shipped metric files (g9) come
    from another realisation of the data and another initialisation; here the reference's OWN
    model and loop semantics (train.py:488-574: zero_grad, n_obs_ot recounted from obs_idx,
    model(...) in train mode, loss.backward(), Adam(lr 1e-3, weight_decay 5e-4).step(); per epoch
    the eval loss of the whole validation set as one batch in eval mode) run on exactly what the
    build's harness feeds its model: dataset seed 0, train_test_split(seed 398), batch 200, the
    epoch orders of njode_amd.parallel.epoch_permutation, initial weights = torch.manual_seed(0)
    + NJODE(...) (stored, so the build starts from the same numbers).  Dropout masks come from
    torch's generator, seeded 1000 + s after the initialisation, s in `seeds`: the spread over s
    is the run-to-run noise a dropout-on comparison has to allow for.  Data only."""
    from njode_amd import data_utils as b_data, parallel as b_par, train as b_train
    arrays = {}
    for name in names:
        paths, obs, nb_obs, hp, sm = ref_dataset(name, 20000, seed=0)
        dt, T = hp['dt'], hp['maturity']
        train_idx, val_idx = b_train.split_indices(len(nb_obs), 0.2, 398)
        val = b_data.collate_arrays(paths[val_idx], obs[val_idx], nb_obs[val_idx], dt)
        # (the build's stock model: bit-identical conditional expectation, and without the
        # reference's TypeError at stock_model.py:139 when the last observation is before T)
        from njode_amd import stock_model as b_stock
        opt = b_stock.STOCK_MODELS[name](**hp).get_optimal_loss(val['times'], val['time_ptr'], val['X'].numpy(), val['obs_idx'].numpy(),
                                  dt, T, val['start_X'].numpy(), val['n_obs_ot'].numpy(), weight=0.5)
        arrays[name + '/optimal_eval_loss'] = np.float64(opt)
        ev = np.zeros((len(seeds), epochs))
        tr = np.zeros((len(seeds), epochs))
        for si, sd in enumerate(seeds):
            model = build(demo_cfg(dropout=0.1), seed=0)
            if si == 0:
                arrays.update({name + '/init/' + k[3:]: v for k, v in sd_arrays(model).items()})
            optimizer = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=0.0005)
            torch.manual_seed(1000 + sd)
            for ep in range(1, epochs + 1):
                model.train()
                order = train_idx[b_par.epoch_permutation(len(train_idx), ep, 0)]
                for s_ in range((len(order) + batch_size - 1) // batch_size):
                    mine = order[s_ * batch_size:(s_ + 1) * batch_size]
                    b = b_data.collate_arrays(paths[mine], obs[mine], nb_obs[mine], dt)
                    n_obs_ot = b_data.recount_observations(b['obs_idx'], len(mine))
                    optimizer.zero_grad()
                    _, loss = model(b['times'], b['time_ptr'], b['X'], b['obs_idx'], dt, T,
                                    b['start_X'], n_obs_ot, return_path=False, get_loss=True)
                    loss.backward()
                    optimizer.step()
                model.eval()
                with torch.no_grad():
                    _, c_loss = model(val['times'], val['time_ptr'], val['X'], val['obs_idx'], dt, T,
                                      val['start_X'], val['n_obs_ot'], return_path=False, get_loss=True)
                ev[si, ep - 1] = float(c_loss)
                tr[si, ep - 1] = float(loss)
                model.epoch += 1
                model.weight_decay_step()
                print('    {} seed {} epoch {:2d}: train {:.5f} eval {:.5f} (optimal {:.5f})'.format(
                    name, sd, ep, tr[si, ep - 1], ev[si, ep - 1], opt), flush=True)
        arrays[name + '/eval_loss'] = ev
        arrays[name + '/train_loss'] = tr
    arrays['seeds'] = np.array(seeds)
    save('g9b_ref_seeded_curves', dict(demo_cfg(dropout=0.1), epochs=epochs, batch_size=batch_size,
                                       names=list(names)), arrays)


def g10():
    """Shapes of the reference's grids that only the shape-generic kernels run (round 3):
    width 100 (parallel_train.py:609), PhysioNet width 200 (:650, masked d = H = 41),
    nn_desc = None with hidden_size 50 (:366-371), the climate shape d = 5, H = 10 masked
    (:433-448; residual cases 1 with mult 2 and 2 with mult 2), and one model whose three
    networks differ (3 hidden layers of different widths / activations, one hidden layer, none)."""
    paths, obs, nb_obs, hp, _ = ref_dataset('BlackScholes', 200)
    dt, T = hp['dt'], hp['maturity']
    nn100 = ((100, 'tanh'), (100, 'tanh'))
    cases = {
        'g6_w100': (dict(demo_cfg(dropout=0.0), ode_nn=nn100, enc_nn=nn100, readout_nn=nn100), 20, None),
        'g6_none_h50': (dict(demo_cfg(H=50, dropout=0.0), ode_nn=None, enc_nn=None, readout_nn=None), 20, None),
        'g6_mixed_nets': (dict(demo_cfg(d=2, dropout=0.0, input_current_t=True),
                               ode_nn=((64, 'tanh'), (32, 'relu'), (48, 'tanh')),
                               enc_nn=((30, 'relu'),), readout_nn=None), 18, ['power-2']),
    }
    for name, (cfg, B, funcs) in cases.items():
        model = build(cfg)
        b = ref_collate(paths, obs, nb_obs, dt, range(B), funcs)
        arrays = {**sd_arrays(model), **batch_arrays(b), 'delta_t': dt, 'T': T}
        arrays.update(eval_outputs(model, b, dt, T))
        arrays.update(grad_outputs(model, b, dt, T))
        save(name, cfg, arrays)
    nn200 = ((200, 'tanh'), (200, 'tanh'))
    masked = {
        'g5_w200': (dict(input_size=41, hidden_size=41, output_size=41, ode_nn=nn200, readout_nn=nn200,
                         enc_nn=nn200, use_rnn=False, bias=True, dropout_rate=0.0,
                         options={'masked': True}),
                    synthetic_physionet.make_batch(batch_size=5, n_grid=60, n_obs_range=(4, 10), seed=2)),
        'g5_climate': (dict(input_size=5, hidden_size=10, output_size=5, ode_nn=NN, readout_nn=NN,
                            enc_nn=NN, use_rnn=False, bias=True, dropout_rate=0.0,
                            options={'masked': True}),
                       synthetic_physionet.make_batch(batch_size=19, dim=5, n_grid=80, n_obs_range=(4, 12),
                                                      p_feature=0.4, seed=5)),
    }
    for name, (cfg, b) in masked.items():
        model = build(cfg)
        arrays = {**sd_arrays(model), **batch_arrays(b, with_M=True), 'delta_t': b['delta_t'], 'T': b['T']}
        out = eval_outputs(model, b, b['delta_t'], b['T'], M=b['M'])
        out.pop('path_h')
        arrays.update(out)
        arrays.update(grad_outputs(model, b, b['delta_t'], b['T'], M=b['M']))
        save(name, cfg, arrays)


def _reference_climate_functions():
    """evaluate_model of NJODE/climate_train.py executed from the reference's own source (the
    module cannot be imported: telegram / sklearn-side imports), with the reference's
    GRU_ODE_Bayes.data_utils_gru_ode_bayes (extract_from_path) imported as it is."""
    import ast
    if not hasattr(np, 'int'):
        np.int = int            # the reference predates numpy 1.24
    import GRU_ODE_Bayes.data_utils_gru_ode_bayes as data_utils_gru
    src = open('/root/reference/NJODE/climate_train.py').read()
    keep = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == 'evaluate_model']
    ns = {'np': np, 'torch': torch, 'data_utils_gru': data_utils_gru}
    exec(compile(ast.Module(keep, []), 'climate_train.py', 'exec'), ns)
    return ns['evaluate_model'], data_utils_gru.extract_from_path


def g11():
    """Climate evaluation protocol (climate_train.py:508-566) on the synthetic stand-in of
    njode_amd/climate_eval.py: the reference's (loss_val, mse_val) and, per batch, what its
    extract_from_path returns."""
    from njode_amd import climate_eval
    evaluate_model, extract = _reference_climate_functions()
    cfg = dict(input_size=5, hidden_size=10, output_size=5, ode_nn=NN, readout_nn=NN, enc_nn=NN,
               use_rnn=False, bias=True, dropout_rate=0.0, options={'masked': True})
    model = build(cfg)
    batches = [climate_eval.make_climate_batch(batch_size=7, T=20, T_val=15, n_obs_range=(5, 12), seed=s)
               for s in (1, 2)]
    dt, T = batches[0]['delta_t'], batches[0]['T']
    ref_batches = [dict(b, times_val=b['times_val'].copy()) for b in batches]   # (the reference edits times_val in place)
    loss_val, mse_val = quiet(evaluate_model, model, ref_batches, 'cpu', {}, dt, T)
    arrays = {**sd_arrays(model), 'delta_t': dt, 'T': T, 'n_batches': len(batches),
              'loss_val': np.float64(loss_val), 'mse_val': np.float64(mse_val)}
    for i, b in enumerate(batches):
        for k in ('times', 'time_ptr', 'times_val', 'index_val'):
            arrays['b{}/{}'.format(i, k)] = np.asarray(b[k])
        for k in ('X', 'M', 'obs_idx', 'X_val', 'M_val'):
            arrays['b{}/{}'.format(i, k)] = b[k].numpy()
        arrays['b{}/batch_size'.format(i)] = len(b['pat_idx'])
        model.eval()
        with torch.no_grad():
            n_obs_ot = torch.tensor(np.bincount(b['obs_idx'].numpy(), minlength=len(b['pat_idx'])))
            _, _, path_t, _, path_y = model(b['times'], b['time_ptr'], b['X'], b['obs_idx'], dt, T,
                                            torch.zeros(len(b['pat_idx']), 5), n_obs_ot, until_T=True,
                                            return_path=True, get_loss=True, M=b['M'])
        t_vec = np.around(path_t, 1).astype(np.float32)
        arrays['b{}/path_t'.format(i)] = np.asarray(path_t, dtype=np.float64)
        arrays['b{}/p_val'.format(i)] = extract(t_vec, path_y, b['times_val'].copy(), b['index_val']).numpy()
    print('    climate eval protocol:', loss_val, mse_val)
    save('g8_climate_eval', cfg, arrays)


def g13():
    """float64 TRUTH for the masked fixtures (round 5).  The reference's own model, the stored
    parameters and inputs of g5_masked / g5_full / g5_w200, evaluated in float64
    (`model.double()`, inputs `.double()`): the times the reference feeds its networks as float32
    tensors (models.py:419 `tau`, :187 `torch.tensor([t])`) stay float32 -- they are inputs that
    both fp32 implementations see -- and every product, tanh and sum behind them is float64.
    tests/test_hip_f64_truth.py compares err(HIP, truth) with err(reference fp32, truth)."""
    arrays = {}
    for name in ('g5_masked', 'g5_full', 'g5_w200'):
        z = np.load(os.path.join(HERE, name + '.npz'), allow_pickle=False)
        cfg = json.loads(str(z['cfg_json']))
        for k in ('ode_nn', 'readout_nn', 'enc_nn'):
            if cfg[k] is not None:
                cfg[k] = tuple(tuple(l) for l in cfg[k])
        model = build(cfg)
        sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('sd/')}
        model.load_state_dict(sd)
        model = model.double()
        b = {'times': z['times'], 'time_ptr': z['time_ptr'].tolist(),
             'X': torch.from_numpy(z['X']).double(), 'obs_idx': torch.from_numpy(z['obs_idx']),
             'start_X': torch.from_numpy(z['start_X']).double(),
             'n_obs_ot': torch.from_numpy(z['n_obs_ot'])}
        M = torch.from_numpy(z['M']).double()
        dt, T = float(z['delta_t']), float(z['T'])
        out = eval_outputs(model, b, dt, T, M=M)
        assert np.array_equal(out['path_t'], z['path_t'])
        if 'path_rows' in z.files:
            out['path_y'] = out['path_y'][z['path_rows']]
        arrays[name + '/path_y'] = out['path_y']
        arrays[name + '/hT'] = out['hT']
        arrays[name + '/loss'] = out['loss']
        g = grad_outputs(model, b, dt, T, M=M)
        arrays[name + '/train_loss'] = g['train_loss']
        arrays[name + '/train_hT'] = g['train_hT']
        for k, v in g.items():
            if k.startswith('grad/'):
                arrays[name + '/' + k] = v
        e32 = np.abs(z['path_y'].astype(np.float64) - out['path_y']).max()
        print('    {}: max |reference fp32 - float64| on path_y = {:.3e}, loss rel {:.2e}'.format(
            name, e32, abs(float(z['loss']) - float(out['loss'])) / abs(float(out['loss']))))
    path = os.path.join(HERE, 'g13_f64_truth.npz')
    np.savez_compressed(path, **arrays)
    print('{:28s} {:8.1f} KB'.format('g13_f64_truth', os.path.getsize(path) / 1024))


def g14():
    """output_size != input_size (round 5): the reference builds a readout to any output_size
    (models.py:350-352); the loss compares X with the readout, so such a model is usable with
    get_loss=False only.  Prediction path of the reference for two shapes: d = 1 -> 5 outputs
    (residual case 2: H = 10 folded by two) and d = 2 ('power-2') -> 20 outputs (case 1), relu nets."""
    paths, obs, nb_obs, hp, _ = ref_dataset('BlackScholes', 200)
    dt, T = hp['dt'], hp['maturity']
    nn32 = ((32, 'relu'), (32, 'tanh'))
    cases = {
        'g14_out5': (dict(demo_cfg(dropout=0.0), output_size=5), 24, None),
        'g14_out20': (dict(demo_cfg(d=2, dropout=0.0), output_size=20, ode_nn=nn32, enc_nn=nn32,
                           readout_nn=nn32), 17, ['power-2']),
    }
    for name, (cfg, B, funcs) in cases.items():
        model = build(cfg)
        b = ref_collate(paths, obs, nb_obs, dt, range(B), funcs)
        arrays = {**sd_arrays(model), **batch_arrays(b), 'delta_t': dt, 'T': T}
        model.eval()
        with torch.no_grad():
            hT, loss, path_t, path_h, path_y = model(
                b['times'], b['time_ptr'], b['X'], b['obs_idx'], dt, T, b['start_X'], b['n_obs_ot'],
                return_path=True, get_loss=False, until_T=True)
            hT2, loss2 = model(b['times'], b['time_ptr'], b['X'], b['obs_idx'], dt, T, b['start_X'],
                               b['n_obs_ot'], return_path=False, get_loss=False)
        assert loss == 0 and loss2 == 0
        arrays.update({'hT': hT.numpy(), 'path_t': np.asarray(path_t, dtype=np.float64),
                       'path_h': path_h.numpy(), 'path_y': path_y.numpy(), 'hT_lastobs': hT2.numpy()})
        save(name, cfg, arrays)


def g15():
    """use_rnn WITH masked data (round 5).  `models.py:353` carries a TODO, but the model runs: the
    GRU cell sees the zero-filled X_obs (no self-imputation at the jump, :460-461), the start state
    goes through the masked encoder (:411-414), the loss is masked, last_X <- Y (:483-484).  The
    reference's own prediction path, loss and gradients on a small PhysioNet-shaped batch."""
    cfg = dict(input_size=5, hidden_size=10, output_size=5, ode_nn=NN, readout_nn=NN, enc_nn=NN,
               use_rnn=True, bias=True, dropout_rate=0.0, options={'masked': True})
    b = synthetic_physionet.make_batch(batch_size=21, dim=5, n_grid=70, n_obs_range=(4, 11),
                                       p_feature=0.4, seed=9)
    model = build(cfg)
    arrays = {**sd_arrays(model), **batch_arrays(b, with_M=True), 'delta_t': b['delta_t'], 'T': b['T']}
    out = eval_outputs(model, b, b['delta_t'], b['T'], M=b['M'])
    arrays.update(out)
    arrays.update(grad_outputs(model, b, b['delta_t'], b['T'], M=b['M']))
    save('g15_rnn_masked', cfg, arrays)


def g16():
    """Gradient THROUGH hT (round 5): the reference returns hT inside its autograd graph
    (models.py:414-518).  For a fixed weight tensor W the reference's gradients of
    loss + <W, hT>  and of  <W, hT>  alone (train mode, dropout 0), on the demo shape (segment plan),
    a PhysioNet-shaped masked model, the GRU jump and a width-100 model (shape-generic kernels)."""
    paths, obs, nb_obs, hp, _ = ref_dataset('BlackScholes', 200)
    dt, T = hp['dt'], hp['maturity']
    nn100 = ((100, 'tanh'), (100, 'tanh'))
    cases = {
        'g16_hT_demo': (demo_cfg(dropout=0.0), ref_collate(paths, obs, nb_obs, dt, range(24)), dt, T, None),
        'g16_hT_rnn': (dict(demo_cfg(dropout=0.0), use_rnn=True), ref_collate(paths, obs, nb_obs, dt, range(30, 49)), dt, T, None),
        'g16_hT_w100': (dict(demo_cfg(dropout=0.0), ode_nn=nn100, enc_nn=nn100, readout_nn=nn100),
                        ref_collate(paths, obs, nb_obs, dt, range(60, 71)), dt, T, None),
    }
    bm = synthetic_physionet.make_batch(batch_size=9, n_grid=120, n_obs_range=(4, 14), seed=4)
    cases['g16_hT_masked'] = (dict(input_size=41, hidden_size=41, output_size=41, ode_nn=NN, readout_nn=NN,
                                   enc_nn=NN, use_rnn=False, bias=True, dropout_rate=0.0,
                                   options={'masked': True}), bm, bm['delta_t'], bm['T'], bm['M'])
    for name, (cfg, b, delta_t, T_, M) in cases.items():
        model = build(cfg)
        arrays = {**sd_arrays(model), **batch_arrays(b, with_M=M is not None), 'delta_t': delta_t, 'T': T_}
        B = b['start_X'].shape[0]
        W = torch.from_numpy(np.random.RandomState(7).standard_normal((B, cfg['hidden_size'])).astype(np.float32))
        arrays['W'] = W.numpy()
        for tag, with_loss in (('both', True), ('hT', False)):
            model.train()
            model.zero_grad()
            hT, loss = model(b['times'], b['time_ptr'], b['X'], b['obs_idx'], delta_t, T_, b['start_X'],
                             b['n_obs_ot'], return_path=False, get_loss=True, M=M)
            obj = (hT * W).sum() + (loss if with_loss else 0.0)
            obj.backward()
            arrays[tag + '/objective'] = np.float64(obj.item())
            for k, p in model.named_parameters():
                # (unmasked models: hT does not depend on the readout -- no gradient there)
                arrays[tag + '/grad/' + k] = (p.grad.numpy().copy() if p.grad is not None
                                              else np.zeros(tuple(p.shape), dtype=np.float32))
        arrays['train_loss'] = np.float64(loss.item())
        arrays['train_hT'] = hT.detach().numpy()
        save(name, cfg, arrays)


if __name__ == '__main__':
    torch.set_num_threads(4)
    which = sys.argv[1:] or ['g1', 'g2', 'g3', 'g4', 'g5', 'g6', 'g7', 'g8', 'g9', 'g10', 'g11', 'g13', 'g14', 'g15', 'g16']
    for name in which:
        globals()[name]()
