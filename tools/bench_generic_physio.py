"""The PhysioNet grid shape of the shape-generic kernels alone (parallel_train.py:650: d = H = 41,
width 200, masked, B = 50, 3 000 Euler steps; lockstep plan) -- the last block of
tools/bench_generic.py, for A/B runs of the paths-per-tile choice (NJODE_GEN_PT)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
sys.path.insert(0, ROOT)
import bench_generic as bg  # noqa: E402
from njode_amd import synthetic_physionet  # noqa: E402

dev = torch.device('cuda', 0)
for width in (200,):
    for B in (50,):
        b = synthetic_physionet.make_batch(batch_size=B, seed=0)
        cfg = dict(input_size=41, hidden_size=41, output_size=41, ode_nn=bg.w(width), readout_nn=bg.w(width),
                   enc_nn=bg.w(width), use_rnn=False, bias=True, dropout_rate=0.1,
                   options={'masked': True, 'device_outputs': True})
        args = (b['times'], b['time_ptr'], b['X'].to(dev), b['obs_idx'].to(dev, torch.int32), b['delta_t'],
                b['T'], b['start_X'].to(dev), b['n_obs_ot'].to(dev, torch.int32), b['M'].to(dev))
        bg.run('PhysioNet d41 H41 w{} masked'.format(width), cfg, args, B, steps=3)
