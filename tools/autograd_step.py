"""The literal reference call sequence (train.py:492-523) on the resident 20 000-path batch, alone
(maintainer aid: run under `rocprofv3 --kernel-trace` to see the route's kernels; bench.py's
autograd_route_ms is the same loop).   python3 tools/autograd_step.py [steps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device('cuda', 0)
b, meta = bench.make_global_slice(0, 20000)
args = (b['times'], b['time_ptr'], b['X'].to(dev), b['obs_idx'].to(dev, torch.int32), meta['dt'],
        meta['maturity'], b['start_X'].to(dev), b['n_obs_ot'].to(dev, torch.int32))
print('autograd route: {:.4f} ms / step'.format(bench.autograd_route_ms(dev, meta['dt'], meta['maturity'], args, 0.1, steps)))
