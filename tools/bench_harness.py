"""Epoch throughput of njode_amd.train.train (the build's harness, reference train.py:488-524
semantics) at the reference's batch sizes and at large batches: 16 000 training paths of the
seed-0 20 000-path Black-Scholes dataset, dropout 0.1, fused step, host collate vs device
collate; beside it the kernel-only rate of the same batch size (resident batch, plan
prefetched: bench.small_batch_ms).  One JSON line per batch size."""
import contextlib, json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from njode_amd import data_utils, models, train  # noqa: E402


def main():
    sizes = [int(x) for x in (sys.argv[1].split(',') if len(sys.argv) > 1 else (100, 200, 1000, 16000))]
    hp = dict(data_utils.hyperparam_default, nb_paths=20000)
    paths, obs, nb_obs, meta = data_utils.create_dataset('BlackScholes', hp, seed=0)
    dev = torch.device('cuda', 0)
    with contextlib.redirect_stdout(sys.stderr):
        model = models.NJODE(**bench.model_cfg(0.1)).to(dev).train()
    opt = models.FusedAdam(model, lr=1e-3, weight_decay=0.0005)
    for B in sizes:
        row = {'case': 'epoch', 'train_paths': 16000, 'batch': B}
        for dc, pa in ((False, 0), (True, 0), (True, 1 << 30)):
            # four epochs of one run; the first one pays the one-off costs (dataset upload,
            # workspace allocation, its own collate counts), the steady-state epochs are 2-4
            with contextlib.redirect_stdout(sys.stderr):
                _, met = train.train((paths, obs, nb_obs), meta, epochs=4, batch_size=B,
                                     log=lambda s: None, device_collate=dc, plan_ahead_min=pa)
            best = min(m[1] for m in met[1:])
            key = ('device_collate' if dc else 'host_collate') + ('_inline_plan' if pa else '')
            n_steps = (16000 + B - 1) // B
            row[key + '_ms_per_step'] = round(best * 1e3 / n_steps, 4)
            row[key + '_paths_per_s'] = round(16000 / best, 1)
        sb = bench.small_batch_ms(model, opt, dev, meta['dt'], meta['maturity'], sizes=(B,), steps=30)
        row['kernel_only_ms_per_step'] = round(sb[B], 4)
        row['kernel_only_paths_per_s'] = round(B / (sb[B] * 1e-3), 1)
        row['harness_over_kernel'] = round(row['device_collate_paths_per_s'] / row['kernel_only_paths_per_s'], 3)
        print(json.dumps(row), flush=True)


if __name__ == '__main__':
    main()
