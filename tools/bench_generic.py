"""Training-step times of the shape-generic kernels (njode_gen.h) on the reference's grid shapes
(fused step: forward + sweep + weight-gradient GEMMs + Adam, dropout 0.1, resident batch), with
the per-kernel split.  One JSON line per (shape, batch).  NJODE_GENERIC=1 in the environment
also sends the demo shape (width 50) to them, for an A/B against the specialised kernels."""
import contextlib, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from njode_amd import _lib, models, synthetic_physionet  # noqa: E402


def w(n):
    return ((n, 'tanh'), (n, 'tanh'))


def run(tag, cfg, args, B, steps=10):
    dev = torch.device('cuda', 0)
    with contextlib.redirect_stdout(sys.stderr):
        torch.manual_seed(0)
        m = models.NJODE(**cfg).to(dev).train()
    opt = models.FusedAdam(m, lr=1e-3, weight_decay=0.0005)

    def one():
        m.loss_and_grad(*args)
        opt.step()
    for _ in range(3):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    _lib.profile_enable(1)
    for _ in range(3):
        one()
    _lib.profile_enable(False)
    k = {n: round(v[1] / max(v[0], 1), 4) for n, v in _lib.profile_read().items()}
    print(json.dumps({'shape': tag, 'B': B, 'ms_per_step': round(ms, 3),
                      'paths_per_s': round(B / ms * 1e3, 1), 'kernel_ms': k,
                      'params': int(m.flat_parameters().numel())}), flush=True)


def main():
    dev = torch.device('cuda', 0)
    for width in (50, 100, 200, 400):
        for B in (100, 2000, 20000):
            if width >= 200 and B > 2000:
                continue
            b, meta = bench.make_batch(B, seed=1)
            cfg = dict(bench.model_cfg(0.1), ode_nn=w(width), enc_nn=w(width), readout_nn=w(width))
            args = (b['times'], b['time_ptr'], b['X'].to(dev), b['obs_idx'].to(dev, torch.int32),
                    meta['dt'], meta['maturity'], b['start_X'].to(dev), b['n_obs_ot'].to(dev, torch.int32))
            run('BS d1 H10 w{}'.format(width), cfg, args, B)
    # PhysioNet grid (parallel_train.py:650): d = H = 41, width 200, B = 50, 3 000 Euler steps
    for width in (50, 200):
        b = synthetic_physionet.make_batch(batch_size=50, seed=0)
        cfg = dict(input_size=41, hidden_size=41, output_size=41, ode_nn=w(width), readout_nn=w(width),
                   enc_nn=w(width), use_rnn=False, bias=True, dropout_rate=0.1,
                   options={'masked': True, 'device_outputs': True})
        args = (b['times'], b['time_ptr'], b['X'].to(dev), b['obs_idx'].to(dev, torch.int32), b['delta_t'],
                b['T'], b['start_X'].to(dev), b['n_obs_ot'].to(dev, torch.int32), b['M'].to(dev))
        run('PhysioNet d41 H41 w{} masked'.format(width), cfg, args, 50, steps=3)


if __name__ == '__main__':
    main()
