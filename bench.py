"""
bench.py -- paths/sec of one NJ-ODE training step (BASELINE.json's metric).

Workload at N = 1 (BASELINE.json configs[1]): 20 000 synthetic Black-Scholes paths
(stock_model.py hyper-parameters of demo.py: drift 2, vol 0.3, 100 grid steps,
obs_perc 0.1, seed = rank), model = demo.py constants (hidden 10, three 50-50 tanh
nets, dropout 0.1, train mode, residual, standard loss).  One "step" = one optimizer
step over all 20 000 paths of this rank as ONE batch:
    batch plan + forward + loss + exact backward (HIP kernels)
    + [N > 1: RCCL all-reduce of the flat gradient, P = 10 071 floats]
    + Adam(lr 1e-3, weight_decay 5e-4) on the flat parameter vector.
Inputs (start_X, X, obs_idx, n_obs_ot) are resident in HBM before the timed region.
N > 1: every rank holds its own 20 000 paths (weak scaling), loss normalised by the
global batch, one gradient all-reduce per step.

Launch:  python bench.py --gpus 1 --steps K --warmup W
         python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
Rank 0 prints ONE JSON line.
"""
import argparse
import contextlib
import copy
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NN = ((50, 'tanh'), (50, 'tanh'))
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP32_MFMA_PEAK_TF = 157.3    # MI355X_MICROARCH.md: peak FP32 matrix (= FP32 vector)


def model_cfg(dropout):
    return dict(input_size=1, hidden_size=10, output_size=1, ode_nn=NN, readout_nn=NN,
                enc_nn=NN, use_rnn=False, bias=True, dropout_rate=dropout,
                options={'device_outputs': True, 'which_loss': 'standard',
                         'residual_enc_dec': True})


def make_batch(n_paths, seed):
    from njode_amd import data_utils
    hp = copy.deepcopy(data_utils.hyperparam_default)
    hp['nb_paths'] = n_paths
    paths, obs, nb_obs, meta = data_utils.create_dataset('BlackScholes', hp, seed=seed)
    return data_utils.collate_arrays(paths, obs, nb_obs, meta['dt']), meta


def flops_per_batch(b, n_hidden_units=50, d=1, H=10):
    """Useful FLOPs of one training step: per Euler step the ODE net,
    per observation readout-before + encoder + readout-after; backward = 2x forward."""
    W = n_hidden_units
    ode = (d + H + 2) * W + W * W + W * H
    enc = d * W + W * W + W * H
    dec = H * W + W * W + W * d
    n_obs = int(b['time_ptr'][-1])
    B = b['start_X'].shape[0]
    # Euler steps actually needed: up to each path's last observation
    last = np.zeros(B, dtype=np.int64)
    obs = b['observed_dates'][:, 1:]
    has = obs.any(axis=1)
    last[has] = obs.shape[1] - np.argmax(obs[has, ::-1], axis=1)
    steps = int(last.sum())
    fwd = 2 * (steps * ode + n_obs * (2 * dec + enc) + B * enc)
    return 3 * fwd, steps, n_obs


def cpu_baseline(meta_dt, T, seconds_budget=28.0):
    """The oracle (CPU restatement of the reference, plain PyTorch) timed on this
    node's host cores on a bounded sample of the same workload: full training steps
    (forward + backward + Adam, dropout 0.1) at the reference's shipped batch size 200
    and on a 4 000-path batch, each at 8 intra-op threads (the survey container's
    setting; more threads only add dispatch overhead on these tiny ops) and at 32.
    The best paths/s is reported with the thread count that produced it."""
    from oracle import njode_oracle
    cfg = model_cfg(0.1)
    o = njode_oracle.make_oracle(cfg)
    params = {k: v.clone().requires_grad_(True) for k, v in o.init_params(0).items()}
    opt = torch.optim.Adam(list(params.values()), lr=1e-3, weight_decay=0.0005)
    default_threads = torch.get_num_threads()
    results = {}
    t_start = time.perf_counter()
    plan = ((200, 8, 10), (4000, 8, 2), (4000, 32, 2), (200, 1, 4))
    try:
        for bsz, threads, max_steps in plan:
            if time.perf_counter() - t_start > seconds_budget:
                break
            torch.set_num_threads(min(threads, os.cpu_count() or threads))
            b, _ = make_batch(bsz, seed=1234)
            njode_oracle.train_step(o, params, opt, b, meta_dt, T)      # warm-up
            n, t0 = 0, time.perf_counter()
            while n < max_steps and time.perf_counter() - t_start < seconds_budget:
                njode_oracle.train_step(o, params, opt, b, meta_dt, T)
                n += 1
            dt = time.perf_counter() - t0
            if n:
                results[(bsz, threads)] = (bsz * n / dt, n)
    finally:
        torch.set_num_threads(default_threads)
    best = max(results, key=lambda k: results[k][0])
    sample = '; '.join('{} steps B={} threads={} -> {:.0f} paths/s'.format(
        results[k][1], k[0], k[1], results[k][0]) for k in sorted(results))
    return {'value': round(results[best][0], 1), 'unit': 'paths/s', 'cores': best[1],
            'kind': 'port',
            'sample': 'oracle train step (fwd+bwd+Adam, dropout 0.1) on synthetic '
                      'Black-Scholes batches: ' + sample,
            'host_cpu_count': os.cpu_count()}


def measured_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC summary (rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE in separate passes, tools/summarize_pmc.py; KiB units).  The
    gfx950 x2 FETCH_SIZE correction applies to wide coalesced streams; this kernel's
    accesses are 4-byte, so the raw counters are reported (MI355X_MICROARCH.md, HBM)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*final_pmc_summary.json')))
    if not files:
        return None, None
    try:
        with open(files[-1]) as f:
            d = json.load(f).get(kernel, {})
        if d.get('FETCH_SIZE') is not None and d.get('WRITE_SIZE') is not None:
            return int((d['FETCH_SIZE'] + d['WRITE_SIZE']) * 1024), d.get('traffic_source')
    except (OSError, ValueError):
        pass
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--paths-per-gpu', type=int, default=20000)
    ap.add_argument('--dropout', type=float, default=0.1)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    distributed = world > 1
    if args.gpus != world and distributed:
        raise SystemExit('--gpus {} but WORLD_SIZE {}'.format(args.gpus, world))
    if args.gpus > 1 and not distributed:
        raise SystemExit('launch N > 1 with torch.distributed.run (one rank per GPU)')
    # NJODE_BENCH_SHARE_GPU=1 (self-test of the N > 1 code path on a one-GPU box): ranks share
    # device 0 and the collective runs over gloo; never set by the driver
    share = os.environ.get('NJODE_BENCH_SHARE_GPU') == '1'
    if share:
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if distributed:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if share:
            torch.distributed.init_process_group('gloo')
        else:
            torch.distributed.init_process_group('nccl', device_id=dev)

    from njode_amd import _lib, models

    B = args.paths_per_gpu
    b, meta = make_batch(B, seed=rank)
    dt, T = meta['dt'], meta['maturity']
    torch.manual_seed(0)                       # identical init on every rank
    with contextlib.redirect_stdout(sys.stderr):   # the ctor prints like the reference's does
        model = models.NJODE(**model_cfg(args.dropout)).to(dev).train()
    model.dp_global_batch = B * world
    model.dp_path_offset = B * rank
    opt = models.FusedAdam(model, lr=1e-3, weight_decay=0.0005, distributed=distributed)
    # inputs resident in HBM before the timed region
    X, start_X = b['X'].to(dev), b['start_X'].to(dev)
    obs_idx = b['obs_idx'].to(dev, torch.int32)
    n_obs_ot = b['n_obs_ot'].to(dev, torch.int32)
    step_args = (b['times'], b['time_ptr'], X, obs_idx, dt, T, start_X, n_obs_ot)

    def step():
        _, loss = model.loss_and_grad(*step_args)
        opt.step()
        return loss

    def sync():
        torch.cuda.synchronize()
        if distributed:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    timing = not args.no_kernel_timing
    sync()
    if timing:
        _lib.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    sync()
    elapsed = time.perf_counter() - t0
    kern = {}
    if timing:
        _lib.profile_enable(False)
        kern = _lib.profile_read()
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if distributed:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(t)
    final_loss = float(loss)

    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        value = B * world * args.steps / elapsed
        flops, euler_steps, n_obs = flops_per_batch(b)
        out = {
            'metric': 'paths/sec (training step) on 20k Black-Scholes, 100 steps',
            'value': round(value, 1), 'unit': 'paths/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(ms, 4), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'BlackScholes {} paths/GPU x 100 grid steps as one batch, '
                                   'hidden_size=10, 3x(50,50) tanh nets, dropout {}, train step '
                                   '= plan+fwd+bwd+{}Adam'.format(
                                       B, args.dropout, 'RCCL all-reduce+' if distributed else ''),
                       'global_batch': B * world, 'n_obs_rows': n_obs,
                       'euler_steps_per_batch': euler_steps, 'params': 10071,
                       'parallelism': 'dp{}'.format(world)},
            'final_loss': final_loss,
        }
        if kern:
            per = {k: round(v[1] / max(v[0], 1), 5) for k, v in kern.items()}
            out['kernel_ms'] = per
            dom = max(kern, key=lambda k: kern[k][1])
            dom_ms = kern[dom][1] / max(kern[dom][0], 1)
            H, W, IN0 = 10, 50, 13
            # ALGORITHMIC work of one launch of the ODE kernels, per
            # Euler step of one path, biases counted as one MAC per output:
            #   forward : (IN0+1) W + (W+1) W + (W+1) H                      = 3 760 MAC
            #   backward: recompute L1+L2 3 250 + transposed products 3 500
            #             + weight-gradient outer products 3 760             = 10 510 MAC
            macs = {'k_ode_bwd_mixed': 10510, 'k_ode_bwd_mfma': 10510, 'k_ode_bwd_items': 10510 + 510,
                    'k_ode_fwd_mfma': 3760, 'k_ode_fwd_items': 3760}.get(dom)
            bytes_ = euler_steps * H * 4 + n_obs * (2 * H * 4 + 32)
            if macs is not None:
                tf_k = 2.0 * macs * euler_steps / (dom_ms * 1e-3) / 1e12
                out['roofline'] = {
                    'bound': 'mfma', 'kernel': dom, 'achieved': round(tf_k, 3),
                    'peak': FP32_MFMA_PEAK_TF, 'unit': 'TFLOP/s',
                    'frac': round(tf_k / FP32_MFMA_PEAK_TF, 5),
                    'traffic': measured_traffic(dom)[0],
                    'traffic_source': measured_traffic(dom)[1],
                    'kernel_ms': round(dom_ms, 5),
                    'algorithmic_flops': int(2 * macs * euler_steps),
                    'note': 'f32 MFMA (v_mfma_f32_16x16x4_f32) dense peak = f32 vector peak; '
                            'useful FLOPs only (tile padding 50->64 not counted)'}
                gbs = bytes_ / (dom_ms * 1e-3) / 1e9
                out['hbm_roofline'] = {
                    'bound': 'hbm', 'kernel': dom, 'achieved': round(gbs, 3),
                    'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(gbs / HBM_PEAK_GBS, 6),
                    'algorithmic_bytes': int(bytes_),
                    'note': '~330 flop/B: the path is compute bound, the HBM fraction is '
                            'reported because BASELINE.json asks for it'}
            tf = flops / (ms * 1e-3) / 1e12
            out['step_flops'] = {'achieved': round(tf, 3), 'peak': FP32_MFMA_PEAK_TF,
                                 'unit': 'TFLOP/s', 'frac': round(tf / FP32_MFMA_PEAK_TF, 5),
                                 'useful_flops_per_step': int(flops),
                                 'note': 'whole step incl. plan, reductions, Adam, launches'}
        if world == 1 and not args.no_cpu_baseline:
            with contextlib.redirect_stdout(sys.stderr):
                out['cpu_baseline'] = cpu_baseline(dt, T)
        print(json.dumps(out), flush=True)   # the ONE line on stdout
    if distributed:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
