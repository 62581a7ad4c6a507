import sys, os, time, contextlib
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np, torch
import bench
from njode_amd import models
dev = torch.device('cuda', 0)
w = ((100, 'tanh'), (100, 'tanh'))
cfg = dict(bench.model_cfg(0.1), ode_nn=w, enc_nn=w, readout_nn=w)
with contextlib.redirect_stdout(sys.stderr):
    torch.manual_seed(0)
    m = models.NJODE(**cfg).to(dev).train()
def args_of(b, meta):
    return (b['times'], b['time_ptr'], b['X'].to(dev), b['obs_idx'].to(dev, torch.int32), meta['dt'], meta['maturity'],
            b['start_X'].to(dev), b['n_obs_ot'].to(dev, torch.int32))
N = 125000
b, meta = bench.make_global_slice(0, N)
m._step_counter = 1
m.dp_global_batch, m.dp_path_offset = N, 0
t0 = time.perf_counter()
_, loss = m.loss_and_grad(*args_of(b, meta)); torch.cuda.synchronize()
print('full', float(loss), 'first call s', round(time.perf_counter() - t0, 2), 'ws GB', round(torch.cuda.max_memory_allocated() / 1e9, 1), flush=True)
g_full = m.flat_grad().clone()
t0 = time.perf_counter()
for _ in range(3):
    m._step_counter = 1
    m.loss_and_grad(*args_of(b, meta))
torch.cuda.synchronize()
print('ms/step', round((time.perf_counter() - t0) / 3 * 1e3, 2), flush=True)
tot, gs = 0.0, torch.zeros_like(g_full)
for lo, hi in ((0, 60000), (60000, N)):
    bs, ms = bench.make_global_slice(lo, hi)
    m._step_counter = 1
    m.dp_global_batch, m.dp_path_offset = N, lo
    _, l = m.loss_and_grad(*args_of(bs, ms))
    tot += float(l); gs += m.flat_grad()
print('shards', tot, 'rel loss', abs(tot - float(loss)) / abs(float(loss)), 'rel grad', float((gs - g_full).norm() / g_full.norm()))
