import sys, os, faulthandler
faulthandler.dump_traceback_later(25, exit=True)
sys.path.insert(0, 'tests'); sys.path.insert(0, os.getcwd())
import numpy as np, torch
from hip_util import bs_batch, demo_cfg, hip_model
def dev_args(b, meta):
    return (b['times'], b['time_ptr'], b['X'].cuda(), b['obs_idx'].cuda().int(), meta['dt'], meta['maturity'],
            b['start_X'].cuda(), b['n_obs_ot'].cuda().int())
for B, seed, perc in ((7, 3, 0.1), (100, 5, 0.1), (150, 6, 0.02), (1000, 7, 0.1), (1600, 8, 0.1), (6000, 9, 0.1)):
  print('=== B', B, flush=True)
  faulthandler.cancel_dump_traceback_later(); faulthandler.dump_traceback_later(25, exit=True)
  batches = [dev_args(*bs_batch(B, seed=seed + 100 * i, obs_perc=perc)) for i in range(2)]
  torch.manual_seed(0)
  m = hip_model(demo_cfg(dropout=0.1)).train()
  print('prefetch 0', flush=True)
  m.prefetch_plan(*batches[0], need_hT=False)
  torch.cuda.synchronize(); print('synced', flush=True)
  for step in range(4):
      m.prefetch_plan(*batches[(step + 1) % 2], need_hT=False)
      print('prefetched', step, flush=True)
      _, loss = m.loss_and_grad(*batches[step % 2])
      print('step', step, float(loss), flush=True)
  m._plans.clear()
  _, loss = m.loss_and_grad(*batches[1])
  print('inline', float(loss), flush=True)
  m.eval()
  with torch.no_grad():
      m.prefetch_plan(*batches[0])
      hT1, loss1 = m(*batches[1])
      print('eval 1', float(loss1), flush=True)
      hT0, loss0 = m(*batches[0])
      print('eval 0', float(loss0), flush=True)
