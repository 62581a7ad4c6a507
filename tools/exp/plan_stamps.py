"""Stage times of the one-launch plan (njode_amd/csrc/njode_plan.h; maintainer measurement, round 5):
the plan of the next step rides in front of the ODE forward of bench.py's loop; NJODE_PLAN_STAMPS=1
makes every plan block write the wall clock at its stage ends.  Prints, per stage, when the first and the
last block passed it (us after the first block's entry)."""
import ctypes
import os
import sys

os.environ['NJODE_PLAN_STAMPS'] = '1'
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench                                    # noqa: E402
from njode_amd import _lib, models              # noqa: E402


def main():
    n_paths = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    mode = sys.argv[2] if len(sys.argv) > 2 else 'hosted'     # hosted | alone
    dev = torch.device('cuda', 0)
    b, meta = bench.make_global_slice(0, n_paths)
    torch.manual_seed(0)
    model = models.NJODE(**bench.model_cfg(0.1)).to(dev).train()
    model.dp_global_batch, model.dp_path_offset = n_paths, 0
    opt = models.FusedAdam(model, lr=1e-3, weight_decay=0.0005)
    args = (b['times'], b['time_ptr'], b['X'].to(dev), b['obs_idx'].to(dev, torch.int32), meta['dt'],
            meta['maturity'], b['start_X'].to(dev), b['n_obs_ot'].to(dev, torch.int32))
    L = _lib.lib()
    model.prefetch_plan(*args, need_hT=False)
    for _ in range(10):
        model.prefetch_plan(*args, need_hT=False)
        if mode == 'alone':
            torch.cuda.synchronize()
            L.njode_plan_flush()                # the plan as a kernel of its own on an idle chip
            torch.cuda.synchronize()
        model.loss_and_grad(*args)
        opt.step()
    torch.cuda.synchronize()
    out = np.zeros(256 * 8, dtype=np.uint64)
    nb = L.njode_debug_plan_stamps(out.ctypes.data_as(ctypes.c_void_p), 256)
    st = out.reshape(256, 8).astype(np.int64)
    used = st[:, 6] > 0
    st = st[used]
    t0 = st[:, 6].min()
    names = ['clear + schedule', 'row times + scatter', 'links + histogram', 'count | layout', 'scan', 'scatter']
    print('paths', n_paths, 'mode', mode, 'plan blocks', int(used.sum()))
    print('  entry          first %7.1f  last %7.1f us' % (0.0, (st[:, 6].max() - t0) / 100.0))
    for s, nm in enumerate(names):
        print('  %-22s first %7.1f  last %7.1f us' % (nm, (st[:, s].min() - t0) / 100.0, (st[:, s].max() - t0) / 100.0))


if __name__ == '__main__':
    main()
