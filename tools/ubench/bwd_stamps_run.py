"""Per-wave time stamps of k_ode_bwd_mixed (diagnostic build: tools/ubench/bwd_variant.sh stamps
-DNJ_BWD_STAMPS).  Runs a few training steps of bench.py's workload, reads the stamps of the LAST
launch and prints where the launch's time goes: prologue (fragment staging, image zeroing), the
sweep over the worker's tiles, the flush, and the idle tail (launch end - the worker's own end).

  NJODE_LIB=$PWD/tools/ubench/libnjode_stamps.so python tools/ubench/bwd_stamps_run.py [--paths N]
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench                     # noqa: E402
from njode_amd import _lib, models   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--paths', type=int, default=20000)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--dropout', type=float, default=0.1)
    ap.add_argument('--json', default='')
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    b, meta = bench.make_global_slice(0, args.paths)
    torch.manual_seed(0)
    model = models.NJODE(**bench.model_cfg(args.dropout)).to(dev).train()
    model.dp_global_batch, model.dp_path_offset = args.paths, 0
    opt = models.FusedAdam(model, lr=1e-3, weight_decay=0.0005)
    step_args = (b['times'], b['time_ptr'], b['X'].to(dev), b['obs_idx'].to(dev, torch.int32),
                 meta['dt'], meta['maturity'], b['start_X'].to(dev), b['n_obs_ot'].to(dev, torch.int32))
    model.prefetch_plan(*step_args, need_hT=False)
    for _ in range(args.steps):
        model.prefetch_plan(*step_args, need_hT=False)
        model.loss_and_grad(*step_args)
        opt.step()
    torch.cuda.synchronize()
    L = _lib.lib()
    n_words = 8192 * 16
    buf = np.zeros(n_words, dtype=np.uint64)
    L.njode_debug_bwd_stamps.argtypes = [C.c_void_p, C.c_uint64]
    rc = L.njode_debug_bwd_stamps(buf.ctypes.data, n_words)
    assert rc == 0, rc
    s = buf.reshape(-1, 16)
    grid = int(s[0, 7] >> np.uint64(32))
    T = int((s[0, 7] >> np.uint64(8)) & np.uint64(0xffffff))
    s = s[:grid * 4]
    role = (s[:, 7] & np.uint64(1)).astype(int)
    t0 = s[:, 0].astype(np.int64)
    base = t0.min()
    tick = 0.01   # us per tick of the 100 MHz wall clock
    start = (t0 - base) * tick
    pro = (s[:, 1].astype(np.int64) - t0) * tick
    sweep = (s[:, 2].astype(np.int64) - s[:, 1].astype(np.int64)) * tick
    end_sweep = (s[:, 2].astype(np.int64) - base) * tick
    # the flush stamp exists for the storing wave only (bulk: wave 0 of a block; four-wave: all)
    has3 = s[:, 3] > s[:, 2]
    flush = np.where(has3, (s[:, 3].astype(np.int64) - s[:, 2].astype(np.int64)) * tick, 0.0)
    end = np.where(has3, (s[:, 3].astype(np.int64) - base) * tick, end_sweep)
    tiles, steps = s[:, 4].astype(int), s[:, 5].astype(int)
    total = end.max()
    out = {'paths': args.paths, 'grid_blocks': grid, 'split_T': T, 'launch_us': round(float(total), 1)}
    print('k_ode_bwd_mixed, {} paths, last of {} launches: grid {} blocks, T = {}, launch {:.1f} us '
          '(first wave start -> last wave end)'.format(args.paths, args.steps, grid, T, total))
    for r, name in ((0, 'bulk (one wave per tile)'), (1, 'four-wave role')):
        m = role == r
        if not m.any():
            continue
        blocks_start = start[m]
        d = {
            'waves': int(m.sum()),
            'start_us': [round(float(np.percentile(blocks_start, p)), 1) for p in (0, 50, 90, 100)],
            'prologue_us': [round(float(np.percentile(pro[m], p)), 2) for p in (50, 90, 100)],
            'sweep_us': [round(float(np.percentile(sweep[m], p)), 1) for p in (0, 50, 90, 100)],
            'flush_us': [round(float(np.percentile(flush[m][has3[m]], p)), 2) for p in (50, 90, 100)] if has3[m].any() else None,
            'end_us': [round(float(np.percentile(end[m], p)), 1) for p in (0, 10, 50, 90, 100)],
            'idle_tail_us_mean': round(float((total - end_sweep[m]).mean()), 1),
            'tiles_per_wave': [int(tiles[m].min()), float(np.round(tiles[m].mean(), 2)), int(tiles[m].max())],
            'steps_per_wave': [int(steps[m].min()), float(np.round(steps[m].mean(), 1)), int(steps[m].max())],
            'us_per_tile_step': round(float(sweep[m].sum() / max(steps[m].sum(), 1)), 3),
            'tile_prologue_us_per_tile': round(float(s[m, 8].sum() * tick / max(tiles[m].sum(), 1)), 3),
            'step_loop_us_per_step': round(float(s[m, 9].sum() * tick / max(steps[m].sum(), 1)), 3),
            'queue_pop_us_per_tile': round(float(s[m, 10].sum() * tick / max(tiles[m].sum(), 1)), 3),
        }
        out[name] = d
        print('  ' + name)
        for k, v in d.items():
            print('    {:22s} {}'.format(k, v))
    # wave-time budget: sum over waves of each phase / (waves x launch time)
    wt = len(s) * total
    budget = {'prologue': float(pro.sum() / wt), 'sweep': float(sweep.sum() / wt),
              'flush': float(flush.sum() / wt), 'not_started': float(start.sum() / wt),
              'idle_after_end': float((total - end).sum() / wt)}
    out['wave_time_share'] = {k: round(v, 4) for k, v in budget.items()}
    print('  share of (waves x launch time):', out['wave_time_share'])
    if args.json:
        with open(args.json, 'a') as f:
            f.write(json.dumps(out) + '\n')


if __name__ == '__main__':
    main()
