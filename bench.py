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
N > 1: every rank holds its own 20 000 paths (weak scaling; --paths-per-gpu), or, with
--global-paths G, the ranks share G paths (strong scaling, G / N each; BASELINE config 4 =
--gpus 8 --global-paths 1000000); the loss is normalised by the global batch, one gradient
all-reduce per step.  The global dataset is a sequence of 20 000-path chunks, chunk c drawn
with seed c, and rank r owns a contiguous slice of it -- so an N-rank run and a 1-rank run
over the same number of paths see the same data (and, dropout being keyed by the global
path id, the same masks).

Launch:  python bench.py --gpus N --steps K --warmup W
  * N = 1, or under a launcher (WORLD_SIZE set: `python -m torch.distributed.run
    --nproc-per-node N ... bench.py --gpus N ...`): this process is a rank.
  * N > 1 without a launcher: this process only SPAWNS the N ranks (torch.distributed.run as a
    child process, started before anything here touches the GPU), relays rank 0's line and
    exits with the children's code.
Rank 0 prints ONE JSON line.
"""
import argparse
import contextlib
import copy
import json
import os
import socket
import subprocess
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


CHUNK = 20000   # paths per chunk of the global dataset (chunk c is drawn with seed c)


def make_global_slice(lo, hi):
    """Paths [lo, hi) of the global synthetic dataset, collated as ONE batch.  The dataset is a
    sequence of CHUNK-path Black-Scholes blocks, block c = create_dataset(seed=c) (so the default
    N = 1 workload is exactly the seed-0 20 000-path dataset), which makes the data a function of
    the global path index only: however the paths are sharded over ranks, their union is the
    same dataset."""
    from njode_amd import data_utils
    hp = copy.deepcopy(data_utils.hyperparam_default)
    hp['nb_paths'] = CHUNK
    parts, meta = [], None
    for c in range(lo // CHUNK, (max(hi, lo + 1) - 1) // CHUNK + 1):
        paths, obs, nb_obs, meta = data_utils.create_dataset('BlackScholes', hp, seed=c)
        a, b = max(lo, c * CHUNK) - c * CHUNK, min(hi, (c + 1) * CHUNK) - c * CHUNK
        parts.append((paths[a:b], obs[a:b], nb_obs[a:b]))
    paths = np.concatenate([p[0] for p in parts])
    obs = np.concatenate([p[1] for p in parts])
    nb_obs = np.concatenate([p[2] for p in parts])
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


def _cpu_model_name():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.lower().startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


CPU_THREADS = 8   # intra-op threads of the CPU baseline (the survey container's core count)


def cpu_baseline(meta_dt, T):
    """The oracle (CPU restatement of the reference, plain PyTorch: kind "port") timed on
    this node's host cores on a bounded sample of the same workload:
    full training steps (forward + backward + Adam(lr 1e-3, wd 5e-4), dropout 0.1, train
    mode) on seeded synthetic Black-Scholes batches, after 2 warm-up steps each:
      * 20 timed steps at B = 100 (demo.py:81),
      * 20 timed steps at B = 200 (the shipped models' batch size)  <- `value`,
      * 2 timed steps at B = 4 000 (large-batch reference point),
    all with a FIXED thread count (CPU_THREADS intra-op threads, or the host's core count
    if smaller); no best-of."""
    from oracle import njode_oracle
    cfg = model_cfg(0.1)
    o = njode_oracle.make_oracle(cfg)
    params = {k: v.clone().requires_grad_(True) for k, v in o.init_params(0).items()}
    opt = torch.optim.Adam(list(params.values()), lr=1e-3, weight_decay=0.0005)
    default_threads = torch.get_num_threads()
    threads = max(1, min(CPU_THREADS, os.cpu_count() or CPU_THREADS))
    results = {}
    try:
        torch.set_num_threads(threads)
        for bsz, n_steps in ((100, 20), (200, 20), (4000, 2)):
            b, _ = make_batch(bsz, seed=1234)
            for _ in range(2):
                njode_oracle.train_step(o, params, opt, b, meta_dt, T)      # warm-up
            t0 = time.perf_counter()
            for _ in range(n_steps):
                njode_oracle.train_step(o, params, opt, b, meta_dt, T)
            dt = time.perf_counter() - t0
            results[bsz] = (bsz * n_steps / dt, n_steps, 1e3 * dt / n_steps)
    finally:
        torch.set_num_threads(default_threads)
    sample = '; '.join('{} steps at B={}: {:.0f} paths/s ({:.1f} ms/step)'.format(
        results[k][1], k, results[k][0], results[k][2]) for k in sorted(results))
    return {'value': round(results[200][0], 1), 'unit': 'paths/s', 'cores': threads,
            'kind': 'port',
            'sample': 'oracle train step (fwd+bwd+Adam, dropout 0.1, train mode) on seeded '
                      'synthetic Black-Scholes batches, {} intra-op threads, 2 warm-up steps '
                      'each: {}; value = the B=200 line'.format(threads, sample),
            'by_batch': {str(k): round(v[0], 1) for k, v in results.items()},
            'cpu_model': _cpu_model_name(), 'host_cpu_count': os.cpu_count()}


PMC_SUMMARY_GLOB = os.path.join(ROOT, 'profiles', 'r*_final_pmc_summary.json')   # newest round first


def _lib_sha256():
    """sha256 of the library the process actually LOADED (njode_amd._lib.LIB_PATH: $NJODE_LIB when
    an ablation / stamp build is selected), so that counters of another build are never reported."""
    import hashlib
    from njode_amd import _lib
    h = hashlib.sha256()
    try:
        with open(_lib.LIB_PATH, 'rb') as f:
            for chunk in iter(lambda: f.read(1 << 20), b''):
                h.update(chunk)
    except OSError:
        return None
    return h.hexdigest()


def measured_traffic(kernel, n_paths, dropout):
    """HBM bytes per launch of `kernel` as MEASURED on this build: profiles/
    r05_final_pmc_summary.json is written by tools/summarize_pmc.py from `rocprofv3 --pmc
    FETCH_SIZE` / `--pmc WRITE_SIZE` passes of `python bench.py` (recipe: profiles/README.md),
    (any round's profiles/r*_final_pmc_summary.json)
    and records the workload it was taken on AND the sha256 of the library it was taken with.
    Returned only when that workload is the one being benchmarked on that very library, else
    None (a counter cannot be collected inside this process; after any kernel change the
    recipe has to be re-run).
    Units: rocprofv3 reports KiB.  MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports half
    the bytes of a coalesced streaming read, WRITE_SIZE is exact -- calibrated on this path's own
    pattern (one dword per lane, 256 B per wave access): the forward's WRITE_SIZE equals the
    bytes it stores (activations + checkpoints, 844 MB), and the backward, which reads those
    same bytes back, shows FETCH_SIZE = 0.56x of them.  Reported: 2 x FETCH_SIZE + WRITE_SIZE."""
    import glob
    d, sha = None, _lib_sha256()
    for path in sorted(glob.glob(PMC_SUMMARY_GLOB), reverse=True):
        try:
            with open(path) as f:
                cand = json.load(f)
        except (OSError, ValueError):
            continue
        if cand.get('_workload', {}).get('lib_sha256') == sha:   # the counters of THIS library
            d = cand
            break
    if d is None:
        return None, None
    wl = d.get('_workload', {})
    if wl.get('paths_per_gpu') != n_paths or abs(wl.get('dropout', -1) - dropout) > 1e-12:
        return None, None
    if wl.get('lib_sha256') != _lib_sha256():     # counters of another build: stale
        return None, None
    k = d.get(kernel, {})
    if k.get('FETCH_SIZE') is None or k.get('WRITE_SIZE') is None:
        return None, None
    return int((2 * k['FETCH_SIZE'] + k['WRITE_SIZE']) * 1024), wl.get('command')


def small_batch_ms(model, opt, dev, dt, T, sizes=(100, 200), steps=30, prefetch=True):
    """ms per training step at the reference's own batch sizes (demo.py:81 trains at B = 100,
    the shipped models at 200), same model / optimizer, inputs resident, the plan built one
    step ahead like the headline loop: an extra, outside the timed region of `value`."""
    out = {}
    for bsz in sizes:
        b, _ = make_batch(bsz, seed=4321)
        args = (b['times'], b['time_ptr'], b['X'].to(dev), b['obs_idx'].to(dev, torch.int32), dt, T,
                b['start_X'].to(dev), b['n_obs_ot'].to(dev, torch.int32))
        model.dp_global_batch, model.dp_path_offset = bsz, 0

        def one():
            if prefetch:
                model.prefetch_plan(*args, need_hT=False)
            model.loss_and_grad(*args)
            opt.step()

        if prefetch:
            model.prefetch_plan(*args, need_hT=False)
        for _ in range(5):
            one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            one()
        torch.cuda.synchronize()
        out[bsz] = 1e3 * (time.perf_counter() - t0) / steps
        model._plans.clear()
    return out


def config5_ms(dev, dropout, sizes=(50, 800, 1000), steps=5):
    """BASELINE config 5 (PhysioNet-shaped, physionet_train.py:93,326-353: d = H = 41, masked,
    3 000 Euler steps): ms per training step (forward + exact backward + fused Adam, inputs resident)
    at the reference's batch size (50), at 800 and at the per-GPU shard of the 8 000-patient data set
    on 8 GPUs (1 000), each with its STRICT f32 fraction: SURVEY 8d's 8 750 MAC per Euler step and
    path x 3 (forward + exact backward) against the 157.3 TF peak.  An extra outside `value`."""
    from njode_amd import models, synthetic_physionet
    out = {}
    cfg = dict(input_size=41, hidden_size=41, output_size=41, ode_nn=NN, readout_nn=NN, enc_nn=NN,
               use_rnn=False, bias=True, dropout_rate=dropout,
               options={'masked': True, 'device_outputs': True})
    for bsz in sizes:
        b = synthetic_physionet.make_batch(batch_size=bsz, seed=0)
        torch.manual_seed(0)
        with contextlib.redirect_stdout(sys.stderr):
            m = models.NJODE(**cfg).to(dev).train()
        opt = models.FusedAdam(m, lr=1e-3, weight_decay=0.0005)
        args = (b['times'], b['time_ptr'], b['X'].to(dev), b['obs_idx'].to(dev, torch.int32), b['delta_t'],
                b['T'], b['start_X'].to(dev), b['n_obs_ot'].to(dev, torch.int32))
        M = b['M'].to(dev)

        def one():
            m.loss_and_grad(*args, M=M)
            opt.step()

        for _ in range(2):
            one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            one()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / steps
        n_steps = int(round(float(b['T']) / float(b['delta_t'])))
        flops = 2.0 * 8750 * 3 * n_steps * bsz
        out[bsz] = {'ms': round(ms, 3), 'paths_per_s': round(bsz / (ms * 1e-3), 1), 'euler_steps': n_steps,
                    'strict_tflops': round(flops / (ms * 1e-3) / 1e12, 3),
                    'strict_frac': round(flops / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TF, 5)}
        del m, opt
    return out


def autograd_route_ms(dev, dt, T, step_args, dropout, steps):
    """The literal call sequence of the reference's training loop (train.py:492-523) on the same
    resident batch: optimizer.zero_grad(); hT, loss = model(...); loss.backward();
    torch.optim.Adam(lr 1e-3, weight_decay 5e-4).step() -- the autograd bridge, hT included (the
    fused step skips the per-path tail evolve nobody reads), no plan prefetch, no fused Adam.
    A fresh model; an extra outside the timed region of `value`."""
    from njode_amd import models
    torch.manual_seed(0)
    with contextlib.redirect_stdout(sys.stderr):
        m = models.NJODE(**model_cfg(dropout)).to(dev).train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=0.0005)

    def one():
        opt.zero_grad()
        hT, loss = m(*step_args, return_path=False, get_loss=True)
        loss.backward()
        opt.step()

    for _ in range(3):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


def comm_probe(model, dev, world, step_args, reps=50):
    """What sets an N-GPU step beside the kernels, each measured ALONE (N > 1 only; extras outside
    `value`): the step's one collective -- all-reduce (SUM) of the [P + 1] gradient bucket -- with
    the ranks aligned by a barrier (`allreduce_alone_ms`: pure collective, no waiting for a slower
    rank), the same collective on ONE float (`allreduce_4B_ms`: the latency floor of a one-shot
    exchange; SURVEY section 5: 40 KB over xGMI is latency, not bandwidth), and the ranks' local
    step without any collective (`local_compute_ms_by_rank`: the shard imbalance).  The in-loop
    `allreduce_ms` of the headline contains both the collective and the wait for the slowest rank;
    these three say which of the two sets the scaling curve."""
    dist = torch.distributed
    bucket = model.grad_bucket().detach().clone()
    tiny = torch.zeros(1, device=dev)

    def timed(t):
        for _ in range(5):
            dist.all_reduce(t)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            dist.all_reduce(t)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    ar_bucket, ar_tiny = timed(bucket), timed(tiny)
    model._plans.clear()
    for _ in range(3):
        model.loss_and_grad(*step_args)
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(10):
        model.loss_and_grad(*step_args)
    torch.cuda.synchronize()
    local = torch.tensor([1e3 * (time.perf_counter() - t0) / 10], device=dev, dtype=torch.float64)
    allr = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(allr, local)
    mx = torch.tensor([ar_bucket, ar_tiny], device=dev, dtype=torch.float64)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    return {'bucket_bytes': int(bucket.numel() * 4), 'allreduce_alone_ms': round(float(mx[0]), 5),
            'allreduce_4B_ms': round(float(mx[1]), 5),
            'local_compute_ms_by_rank': [round(float(x), 4) for x in allr],
            'note': 'local compute = plan in line + forward + backward of the rank\'s shard, no collective, '
                    'no optimizer; allreduce_* = max over ranks of the mean of {} back-to-back '
                    'collectives behind a barrier'.format(reps)}


def strong_scaling_20k(model, opt, dev, world, rank, steps, warmup, prefetch):
    """BASELINE's own 20 000-path batch SHARED by the N ranks (20 000 / N paths each, loss over the
    global batch, the same one all-reduce + fused Adam per step), timed like the headline (barrier
    + synchronize on both sides, max over ranks): the strong-scaling reading of `paths/sec on 20k
    Black-Scholes at 1/2/4/8 GPUs`, next to the weak-scaling headline, so that a SCALE record
    cannot be read two ways.  An extra outside the timed region of `value`."""
    from njode_amd import parallel
    G = 20000
    lo, hi = parallel.shard_range(G, world, rank)
    b, meta = make_global_slice(lo, hi)
    model.dp_global_batch, model.dp_path_offset = G, lo
    model._plans.clear()
    args = (b['times'], b['time_ptr'], b['X'].to(dev), b['obs_idx'].to(dev, torch.int32), meta['dt'],
            meta['maturity'], b['start_X'].to(dev), b['n_obs_ot'].to(dev, torch.int32))

    def one():
        if prefetch:
            model.prefetch_plan(*args, need_hT=False)
        model.loss_and_grad(*args)
        opt.step()

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    if prefetch:
        model.prefetch_plan(*args, need_hT=False)
    for _ in range(warmup):
        one()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    sync()
    t = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    model._plans.clear()
    return 1e3 * float(t) / steps, hi - lo


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def count_gpus_without_runtime():
    """Number of AMD GPUs of this node WITHOUT touching the HIP / ROCr runtime (the parent of the
    ranks must never initialise the GPU: a later exec / fork of an initialised process is what the
    pool forbids, and `torch.cuda.device_count()` is `hipGetDeviceCount` on a ROCm torch without
    amdsmi).  KFD's topology lists every node; GPUs are the nodes with `simd_count > 0`.  The
    visibility variables the runtime would apply are applied here too.  Returns None when the
    topology is not readable (no KFD: not a ROCm box)."""
    root = '/sys/class/kfd/kfd/topology/nodes'
    try:
        nodes = sorted(os.listdir(root), key=lambda x: int(x) if x.isdigit() else 1 << 30)
    except OSError:
        return None
    n = 0
    for node in nodes:
        try:
            with open(os.path.join(root, node, 'properties')) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
        except OSError:
            continue
        if int(props.get('simd_count', '0')) > 0:
            n += 1
    for var in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        v = os.environ.get(var)
        if v is not None:
            listed = [x for x in v.split(',') if x.strip() != '']
            n = min(n, len(listed))
    return n


def spawn_ranks(n, argv, rank_timeout=900.0):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes
    (python -m torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1) in a process
    group of their own, relay rank 0's JSON line and return their exit code.  This parent never
    touches the GPU (devices are counted from KFD's topology, not through the runtime) and never
    replaces itself.  `rank_timeout` seconds without the children finishing (a hung rendezvous,
    a rank that died before the first collective) kills the whole group and returns 124; when
    the ranks fail, the tail of their stderr (torchrun names the first failing rank) is shown."""
    share = os.environ.get('NJODE_BENCH_SHARE_GPU') == '1'
    n_dev = count_gpus_without_runtime()
    if n_dev is not None and n_dev < n and not share:
        print('bench.py: --gpus {} but this node has {} GPU(s)'.format(n, n_dev), file=sys.stderr)
        return 2
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
           os.path.abspath(__file__)] + list(argv)
    import signal
    import tempfile

    def kill_group(child):
        """SIGTERM, then SIGKILL, to the launcher AND its ranks (they are a session of their own)."""
        for sig in (signal.SIGTERM, signal.SIGKILL):
            try:
                os.killpg(child.pid, sig)
            except (ProcessLookupError, PermissionError):
                return
            try:
                child.wait(timeout=10)
                return
            except subprocess.TimeoutExpired:
                continue

    with tempfile.TemporaryFile(mode='w+') as err:
        child = subprocess.Popen(cmd, env=env, stderr=err, start_new_session=True)
        # The ranks do not share this process's group (start_new_session), so a terminal Ctrl-C or
        # a SIGTERM to bench.py does not reach them by itself: whatever ends the wait below --
        # the timeout, KeyboardInterrupt, a SIGTERM (turned into SystemExit here), any other
        # exception -- takes the whole group down before it propagates.  Nothing is re-executed.
        def on_term(signum, frame):
            raise SystemExit(128 + signum)
        old_term = signal.signal(signal.SIGTERM, on_term)
        try:
            rc = child.wait(timeout=rank_timeout if rank_timeout and rank_timeout > 0 else None)
        except subprocess.TimeoutExpired:
            rc = 124
            kill_group(child)
            print('bench.py: the {} ranks did not finish within {:.0f} s (--rank-timeout): process '
                  'group killed'.format(n, rank_timeout), file=sys.stderr)
        except BaseException:
            kill_group(child)
            raise
        finally:
            signal.signal(signal.SIGTERM, old_term)
        err.seek(0)
        text = err.read()
    if rc != 0:
        tail = text.strip().splitlines()[-40:]
        print('bench.py: ranks exited with code {}; last lines of their stderr:'.format(rc),
              file=sys.stderr)
        for line in tail:
            print('  | ' + line, file=sys.stderr)
    elif text:
        sys.stderr.write(text)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    # defaults: 20 untimed + 100 timed steps (~0.1 s).  The clocks of an idle MI355X take tens of
    # steps of this size to settle: from a cold start a 3 + 20 window reads ~4 % slow
    # (0.921 - 0.933 ms against 0.889 ms per step on one box, profiles/r04_bench_window.txt)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--paths-per-gpu', type=int, default=20000,
                    help='weak scaling: every rank holds this many paths (default)')
    ap.add_argument('--global-paths', type=int, default=0,
                    help='strong scaling: the ranks share this many paths (e.g. 1000000 on 8 GPUs)')
    ap.add_argument('--dropout', type=float, default=0.1)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--no-small-batch', action='store_true')
    ap.add_argument('--no-autograd-route', action='store_true')
    ap.add_argument('--no-config5', action='store_true',
                    help='skip the PhysioNet-shaped extras c5_b50_ms / c5_b800_ms / c5_b1000_ms')
    ap.add_argument('--no-strong-20k', action='store_true',
                    help='N > 1, weak scaling: skip the extra strong-scaling pass over the 20 000-path batch')
    ap.add_argument('--dump-params', default='',
                    help='rank 0 saves the flat parameter vector after the timed steps (.npy)')
    ap.add_argument('--no-plan-prefetch', action='store_true',
                    help='build every step\'s plan in line instead of one step ahead')
    ap.add_argument('--rank-timeout', type=float, default=900.0,
                    help='self-spawned ranks (--gpus N without a launcher): kill the ranks\' process '
                         'group and exit 124 after this many seconds (0: wait for ever)')
    args = ap.parse_args()

    launched = 'WORLD_SIZE' in os.environ and 'RANK' in os.environ
    if args.gpus > 1 and not launched:
        # no launcher: spawn the ranks.  Nothing in this process has touched the GPU
        # (importing torch does not initialise it; devices are counted from KFD's topology).
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:], args.rank_timeout))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    distributed = world > 1
    if args.gpus != world:
        raise SystemExit('--gpus {} but WORLD_SIZE {}'.format(args.gpus, world))
    # NJODE_BENCH_SHARE_GPU=1 (self-test of the N > 1 code path on a one-GPU box): ranks share
    # device 0 and the collective runs over gloo; never set by the driver
    share = os.environ.get('NJODE_BENCH_SHARE_GPU') == '1'
    n_dev = torch.cuda.device_count()
    if share:
        local_rank = local_rank % max(n_dev, 1)
    elif local_rank >= n_dev:
        raise SystemExit('rank {} needs cuda:{} but this node has {} GPU(s) (one rank per GPU; '
                         'NJODE_BENCH_SHARE_GPU=1 runs the ranks on one device over gloo as a '
                         'self-test)'.format(rank, local_rank, n_dev))
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    backend = None
    if distributed:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        backend = 'gloo' if share else 'nccl'
        if share:
            torch.distributed.init_process_group('gloo')
        else:
            torch.distributed.init_process_group('nccl', device_id=dev)

    from njode_amd import _lib, models, parallel

    strong = args.global_paths > 0
    global_batch = args.global_paths if strong else args.paths_per_gpu * world
    lo, hi = parallel.shard_range(global_batch, world, rank)
    B, path_offset = hi - lo, lo
    b, meta = make_global_slice(lo, hi)
    dt, T = meta['dt'], meta['maturity']
    torch.manual_seed(0)                       # identical init on every rank
    with contextlib.redirect_stdout(sys.stderr):   # the ctor prints like the reference's does
        model = models.NJODE(**model_cfg(args.dropout)).to(dev).train()
    model.dp_global_batch = global_batch
    model.dp_path_offset = path_offset
    opt = models.FusedAdam(model, lr=1e-3, weight_decay=0.0005, distributed=distributed)
    # inputs resident in HBM before the timed region
    X, start_X = b['X'].to(dev), b['start_X'].to(dev)
    obs_idx = b['obs_idx'].to(dev, torch.int32)
    n_obs_ot = b['n_obs_ot'].to(dev, torch.int32)
    step_args = (b['times'], b['time_ptr'], X, obs_idx, dt, T, start_X, n_obs_ot)

    # The plan of a step (schedule copy, rows linked per path, segments sorted by length,
    # trajectory layout: ~0.12 ms of small latency-bound kernels) depends on the batch only, so a
    # training loop that holds the next batch builds it on a helper stream beside the current
    # step (NJODE.prefetch_plan -> njode_plan_f32).  Every step still builds exactly one plan
    # inside the timed region -- the next step's; --no-plan-prefetch builds each in line.
    prefetch = not args.no_plan_prefetch

    def step():
        if prefetch:
            model.prefetch_plan(*step_args, need_hT=False)
        _, loss = model.loss_and_grad(*step_args)
        opt.step()
        return loss

    if prefetch:
        model.prefetch_plan(*step_args, need_hT=False)   # the first step's own plan

    def sync():
        torch.cuda.synchronize()
        if distributed:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    timing = not args.no_kernel_timing and world == 1   # per-kernel times / roofline: N = 1 only
    sync()
    # HIP events inside the timed region: around the DOMINANT kernel only (the ODE backward) --
    # bracketing all six hot kernels costs ~3.5 % of the step.  The other kernels' times come
    # from a second, untimed pass of the same K steps below (`kernel_ms`).
    if timing:
        _lib.profile_enable(2)
    opt.time_allreduce = distributed        # HIP events around the collective (device time)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    sync()
    elapsed_local = time.perf_counter() - t0
    opt.time_allreduce = False
    allreduce_ms = opt.allreduce_ms()
    flat_after = model.flat_parameters().detach().clone()   # parameters after warmup + K steps
    kern, kern_timed = {}, {}
    if timing:
        _lib.profile_enable(False)
        kern_timed = _lib.profile_read()          # the dominant kernel, live in the timed region
        _lib.profile_enable(1)
        for _ in range(args.steps):
            step()
        sync()
        _lib.profile_enable(False)
        kern = _lib.profile_read()                # all kernels, second pass (not timed)
        kern.update(kern_timed)
    t = torch.tensor([elapsed_local], device=dev, dtype=torch.float64)
    per_rank = [elapsed_local]
    if distributed:
        gathered = [torch.zeros_like(t) for _ in range(world)]
        torch.distributed.all_gather(gathered, t)
        per_rank = [float(x) for x in gathered]
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(t)
    # the ranks' losses are partial sums over their shards (global denominator): add them up;
    # and every rank must hold the same parameters after the same all-reduced steps
    # (N > 1: the loss was all-reduced WITH the gradient -- the last slot of the bucket -- by the
    # last optimizer step: `loss` is a view of that slot and holds the global loss already)
    lsum = loss.detach().reshape(1).to(torch.float64).clone()
    flat = flat_after
    identical = True
    if distributed:
        pmax, pmin = flat.clone(), flat.clone()
        torch.distributed.all_reduce(pmax, op=torch.distributed.ReduceOp.MAX)
        torch.distributed.all_reduce(pmin, op=torch.distributed.ReduceOp.MIN)
        identical = bool(torch.equal(pmax, pmin))
    final_loss = float(lsum)
    if args.dump_params and rank == 0:
        np.save(args.dump_params, flat.cpu().numpy())

    comm = comm_probe(model, dev, world, step_args) if distributed else None
    strong20 = None
    if distributed and not strong and not args.no_strong_20k:
        strong20 = strong_scaling_20k(model, opt, dev, world, rank, args.steps, min(args.warmup, 10), prefetch)

    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        value = global_batch * args.steps / elapsed
        flops, euler_steps, n_obs = flops_per_batch(b)
        out = {
            'metric': 'paths/sec (training step) on 20k Black-Scholes, 100 steps',
            'value': round(value, 1), 'unit': 'paths/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(ms, 4), 'higher_is_better': True,
            'scaling': 'strong' if strong else 'weak', 'vs_baseline': None, 'dtype': 'f32',
            'data': 'synthetic',
            'config': {'workload': 'BlackScholes {} x 100 grid steps, every rank\'s paths as one '
                                   'batch, hidden_size=10, 3x(50,50) tanh nets, dropout {}, train '
                                   'step = plan+fwd+bwd+{}Adam'.format(
                                       '{} paths shared by {} GPU(s)'.format(global_batch, world)
                                       if strong else '{} paths/GPU'.format(B),
                                       args.dropout, 'RCCL all-reduce+' if distributed else ''),
                       'global_batch': global_batch, 'paths_rank0': B, 'n_obs_rows': n_obs,
                       'euler_steps_per_batch': euler_steps, 'params': 10071,
                       'parallelism': 'dp{}'.format(world)},
            'final_loss': final_loss,
            'plan_prefetch': bool(prefetch),
        }
        if distributed:
            out['params_identical_across_ranks'] = identical
            out['rccl_world'] = torch.distributed.get_world_size()
            out['collective_backend'] = backend + (' (RCCL)' if backend == 'nccl' else '')
            out['allreduce_ms'] = None if allreduce_ms is None else round(allreduce_ms, 5)
            out['comm'] = comm
            out['allreduce_floats'] = int(model.grad_bucket().numel())   # gradient + the loss slot
            out['ms_per_step_by_rank'] = [round(1e3 * x / args.steps, 4) for x in per_rank]
            if strong20 is not None:
                out['strong_20k_ms'] = round(strong20[0], 4)
                out['strong_20k_paths_per_s'] = round(20000 / (strong20[0] * 1e-3), 1)
                out['strong_20k_note'] = ('extra, outside `value`: BASELINE\'s 20 000-path batch shared by the {} '
                                          'ranks ({} paths on rank 0), same all-reduce + Adam per step; `value` '
                                          'is WEAK scaling ({} paths per rank)'.format(world, strong20[1], B))
        if kern:
            per = {k: round(v[1] / max(v[0], 1), 5) for k, v in kern.items()}
            out['kernel_ms'] = per
            dom = max(kern, key=lambda k: kern[k][1])
            dom_ms = kern[dom][1] / max(kern[dom][0], 1)
            H, W, IN0 = 10, 50, 13
            # ALGORITHMIC work of one launch of the ODE kernels, per
            # Euler step of one path.  STRICT: the network is 13.50 + 50.50 +
            # 50.10 = 3 650 MAC forward, its exact backward 2x that = 7 300 MAC = 14 600 flop.
            # EXECUTED (useful work the kernel really issues, biases as one MAC per output):
            # forward 3 760; the backward reads the forward's stored hidden activations, so it
            # issues the transposed products 3 500 + the weight-gradient outer products 3 760 =
            # 7 260 MAC and recomputes nothing (round 1 recomputed L1+L2: 10 510).  Tile padding
            # (50->64, 10->16) is never counted.
            strict = {'k_ode_bwd_mixed': 7300, 'k_ode_bwd_mfma': 7300, 'k_ode_bwd_items': 7300,
                      'k_ode_fwd_mixed': 3650, 'k_ode_fwd_mfma': 3650, 'k_ode_fwd_items': 3650}.get(dom)
            executed = {'k_ode_bwd_mixed': 7260, 'k_ode_bwd_mfma': 7260,
                        'k_ode_bwd_items': 10510 + 510, 'k_ode_fwd_mixed': 3760,
                        'k_ode_fwd_mfma': 3760, 'k_ode_fwd_items': 3760}.get(dom)
            # HBM bytes one launch must move: per Euler step of one path the state checkpoint
            # (H floats) and the stored activation record (8 x ceil(W/4) floats, unmasked
            # train steps only); per observation row the jump state in and out + its item record.
            act_rec = 8 * ((W + 3) // 4) * 4
            bytes_ = euler_steps * (H * 4 + act_rec) + n_obs * (2 * H * 4 + 32)
            if strict is not None:
                tf_s = 2.0 * strict * euler_steps / (dom_ms * 1e-3) / 1e12
                tf_e = 2.0 * executed * euler_steps / (dom_ms * 1e-3) / 1e12
                traffic, traffic_src = measured_traffic(dom, B, args.dropout)
                out['roofline'] = {
                    'bound': 'mfma', 'kernel': dom, 'achieved': round(tf_s, 3),
                    'peak': FP32_MFMA_PEAK_TF, 'unit': 'TFLOP/s',
                    'frac': round(tf_s / FP32_MFMA_PEAK_TF, 5),
                    'frac_strict': round(tf_s / FP32_MFMA_PEAK_TF, 5),
                    'frac_executed': round(tf_e / FP32_MFMA_PEAK_TF, 5),
                    'achieved_executed': round(tf_e, 3),
                    'traffic': traffic, 'traffic_source': traffic_src,
                    'kernel_ms': round(dom_ms, 5),
                    'kernel_ms_source': ('HIP events on the launch stream inside the timed region'
                                         if dom in kern_timed else
                                         'HIP events, second pass of the same steps'),
                    'algorithmic_flops': int(2 * strict * euler_steps),
                    'executed_useful_flops': int(2 * executed * euler_steps),
                    'note': 'achieved / frac = STRICT algorithmic work (SURVEY 8d: 14 600 flop per '
                            'Euler step of the backward, 7 300 forward); frac_executed counts the '
                            'MACs the kernel really issues (7 260 backward on stored activations, '
                            '3 760 forward with biases).  Peak = dense f32 MFMA = f32 vector peak; on gfx950 '
                            'v_mfma_f32_16x16x4_f32 and the VALU share ONE pipe (measured: '
                            'profiles/r02_pipe_ubench.jsonl), so the tanh / dropout / delta VALU '
                            'work of the kernel spends the same peak (DESIGN.md section 5)'}
                gbs = bytes_ / (dom_ms * 1e-3) / 1e9
                out['hbm_roofline'] = {
                    'bound': 'hbm', 'kernel': dom, 'achieved': round(gbs, 3),
                    'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(gbs / HBM_PEAK_GBS, 6),
                    'algorithmic_bytes': int(bytes_),
                    'note': '~32 flop/B with the stored activations (machine balance '
                            '~20 flop/B): still compute bound; the HBM fraction is reported '
                            'because BASELINE.json asks for it'}
            tf = flops / (ms * 1e-3) / 1e12
            out['step_flops'] = {'achieved': round(tf, 3), 'peak': FP32_MFMA_PEAK_TF,
                                 'unit': 'TFLOP/s', 'frac': round(tf / FP32_MFMA_PEAK_TF, 5),
                                 'useful_flops_per_step': int(flops),
                                 'note': 'whole step incl. plan, reductions, Adam, launches'}
        if world == 1 and not args.no_small_batch:
            model._plans.clear()
            sb = small_batch_ms(model, opt, dev, dt, T, prefetch=prefetch)
            out['b100_ms'] = round(sb[100], 4)
            out['b200_ms'] = round(sb[200], 4)
            out['b100_paths_per_s'] = round(100 / (sb[100] * 1e-3), 1)
            out['b200_paths_per_s'] = round(200 / (sb[200] * 1e-3), 1)
        if world == 1 and not args.no_autograd_route:
            model._plans.clear()
            out['autograd_route_ms'] = round(autograd_route_ms(dev, dt, T, step_args, args.dropout,
                                                               args.steps), 4)
            out['autograd_route_paths_per_s'] = round(B / (out['autograd_route_ms'] * 1e-3), 1)
            # ... and at the reference's own batch size (demo.py:81): the same literal sequence, B = 100
            b100, _ = make_batch(100, seed=4321)
            args100 = (b100['times'], b100['time_ptr'], b100['X'].to(dev), b100['obs_idx'].to(dev, torch.int32),
                       dt, T, b100['start_X'].to(dev), b100['n_obs_ot'].to(dev, torch.int32))
            out['autograd_route_b100_ms'] = round(autograd_route_ms(dev, dt, T, args100, args.dropout, 50), 4)
        if world == 1 and not args.no_config5:
            c5 = config5_ms(dev, args.dropout)
            for bsz, v in c5.items():
                out['c5_b{}_ms'.format(bsz)] = v['ms']
            out['config5'] = {'workload': 'PhysioNet-shaped masked batch (synthetic_physionet.py: d = H = 41, '
                                          '3 000 Euler steps, 30-100 observation times per path), training step '
                                          '= forward + exact backward + fused Adam, dropout as the headline',
                              'strict_flops_note': '2 x 8 750 MAC x 3 (fwd + bwd) x Euler steps x paths (SURVEY 8d)',
                              'peak_tflops': FP32_MFMA_PEAK_TF,
                              **{'b{}'.format(k): v for k, v in c5.items()}}
        if world == 1 and not args.no_cpu_baseline:
            with contextlib.redirect_stdout(sys.stderr):
                out['cpu_baseline'] = cpu_baseline(dt, T)
        print(json.dumps(out), flush=True)   # the ONE line on stdout
    if distributed:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
