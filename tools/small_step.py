"""The fused training step at a small batch size, alone (maintainer aid: run it under
`rocprofv3 --kernel-trace --stats` to get the per-kernel times at the reference's batch sizes;
bench.py's b100_ms / b200_ms are the same loop).   python3 tools/small_step.py [B] [steps]"""
import contextlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    from njode_amd import models
    dev = torch.device('cuda', 0)
    b, meta = bench.make_batch(B, seed=4321)
    dt, T = meta['dt'], meta['maturity']
    torch.manual_seed(0)
    with contextlib.redirect_stdout(sys.stderr):
        model = models.NJODE(**bench.model_cfg(0.1)).to(dev).train()
    opt = models.FusedAdam(model, lr=1e-3, weight_decay=0.0005)
    args = (b['times'], b['time_ptr'], b['X'].to(dev), b['obs_idx'].to(dev, torch.int32), dt, T,
            b['start_X'].to(dev), b['n_obs_ot'].to(dev, torch.int32))
    model.dp_global_batch, model.dp_path_offset = B, 0

    def one():
        model.prefetch_plan(*args, need_hT=False)
        model.loss_and_grad(*args)
        opt.step()

    model.prefetch_plan(*args, need_hT=False)
    for _ in range(10):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    torch.cuda.synchronize()
    print('B = {}: {:.4f} ms / step'.format(B, 1e3 * (time.perf_counter() - t0) / steps))


if __name__ == '__main__':
    main()
