"""What the next step's plan costs the step it runs beside (maintainer measurement, round 5).
bench.py's loop builds the plan of step i+1 on a second queue beside the forward and the row pass
of step i.  Here the same loop runs three ways on the same resident batch:
  prefetch   as bench.py (one plan built per step, beside the step)
  reuse      ONE plan built before the loop and handed to every step (a plan is read-only for the
             forward and the backward): no plan work at all inside the loop -- the floor
  inline     every step builds its plan in line (no second queue)
prints ms per step and the kernels' times (HIP events) for each."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench                                    # noqa: E402
from njode_amd import _lib, models              # noqa: E402


def main():
    steps, warm = 100, 20
    n_paths = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    dev = torch.device('cuda', 0)
    b, meta = bench.make_global_slice(0, n_paths)
    torch.manual_seed(0)
    model = models.NJODE(**bench.model_cfg(0.1)).to(dev).train()
    model.dp_global_batch, model.dp_path_offset = n_paths, 0
    opt = models.FusedAdam(model, lr=1e-3, weight_decay=0.0005)
    args = (b['times'], b['time_ptr'], b['X'].to(dev), b['obs_idx'].to(dev, torch.int32), meta['dt'],
            meta['maturity'], b['start_X'].to(dev), b['n_obs_ot'].to(dev, torch.int32))

    def run(mode):
        model._plans.clear()
        shared = None
        if mode == 'reuse':
            shared = model.prefetch_plan(*args, need_hT=False)
            model._plans.clear()
        elif mode == 'prefetch':
            model.prefetch_plan(*args, need_hT=False)

        def one():
            if mode == 'prefetch':
                model.prefetch_plan(*args, need_hT=False)
                model.loss_and_grad(*args)
            elif mode == 'reuse':
                shared.taken = False
                model.loss_and_grad(*args, plan=shared)
            else:
                model.loss_and_grad(*args)
            opt.step()

        for _ in range(warm):
            one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            one()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / steps
        _lib.profile_enable(1)
        for _ in range(steps):
            one()
        torch.cuda.synchronize()
        _lib.profile_enable(False)
        k = _lib.profile_read()
        per = {n: round(v[1] / max(v[0], 1), 4) for n, v in k.items() if n.startswith('k_')}
        print(mode, round(ms, 4), per, flush=True)

    for _ in range(2):
        for mode in ('prefetch', 'reuse', 'inline'):
            run(mode)


if __name__ == '__main__':
    main()
