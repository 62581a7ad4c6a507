"""Worker of tests/test_hip_two_rank.py, launched by torch.distributed.run with two ranks that
SHARE the one GPU of the test box (RCCL refuses two ranks on one device, so the collective
runs over gloo on the CUDA tensors; the data-parallel code path -- shard by path, global
loss denominator, dropout keyed by global path id, one all-reduce of the flat gradient,
identical fused Adam step -- is the one bench.py / train.py run on N GPUs).

Each rank trains the demo model for three steps on its shard of the same global batches;
afterwards the parameters must be BIT-identical on the two ranks, and equal (1e-5) to a
single-process run over the whole batches."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main(out_dir):
    from hip_util import demo_cfg
    from njode_amd import data_utils, models, parallel

    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    torch.cuda.set_device(0)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.distributed.init_process_group('gloo')
    hp = dict(data_utils.hyperparam_default, nb_paths=96)
    paths, obs, nb_obs, meta = data_utils.create_dataset('BlackScholes', hp, seed=0)
    dt, T = meta['dt'], meta['maturity']
    cfg = demo_cfg(dropout=0.1)           # dropout ON: masks must not depend on the sharding
    batches = [parallel.epoch_permutation(96, epoch=e)[:37 + 8 * e] for e in range(3)]   # ragged

    def run(world_, rank_, distributed):
        torch.manual_seed(0)
        m = models.NJODE(**dict(cfg, options=dict(cfg['options'], device_outputs=True))).cuda().train()
        flat = m.flat_parameters()
        if distributed:
            if rank_ == 1:
                flat.add_(1.0)            # rank 1 starts from garbage; the broadcast repairs it
            parallel.broadcast_parameters_(flat, src=0)
        opt = models.FusedAdam(m, lr=1e-3, weight_decay=0.0005, distributed=distributed)
        losses = []
        for idx in batches:
            lo, hi = parallel.shard_range(len(idx), world_, rank_)
            parallel.configure_model(m, len(idx), lo)
            mine = idx[lo:hi]
            b = data_utils.collate_arrays(paths[mine], obs[mine], nb_obs[mine], dt)
            n_obs_ot = data_utils.recount_observations(b['obs_idx'], len(mine))
            _, loss = m.loss_and_grad(b['times'], b['time_ptr'], b['X'].cuda(),
                                      b['obs_idx'].cuda().int(), dt, T, b['start_X'].cuda(),
                                      n_obs_ot.cuda().int())
            opt.step()
            losses.append(loss.detach().reshape(1).clone())
        torch.cuda.synchronize()
        return m.flat_parameters().detach().clone(), torch.cat(losses)

    # (the loss is all-reduced WITH the gradient by FusedAdam.step: the values read after the
    # step are the global losses already)
    p_dp, l_dp = run(world, rank, True)
    gathered = [torch.zeros_like(p_dp) for _ in range(world)]
    torch.distributed.all_gather(gathered, p_dp)
    identical = bool(torch.equal(gathered[0], gathered[1]))
    if rank == 0:
        p_1, l_1 = run(1, 0, False)
        rel = float((p_dp - p_1).norm() / p_1.norm())
        max_abs = float((p_dp - p_1).abs().max())
        with open(os.path.join(out_dir, 'result.json'), 'w') as f:
            json.dump({'identical_across_ranks': identical, 'rel_vs_single': rel,
                       'max_abs_vs_single': max_abs, 'losses_dp': l_dp.cpu().tolist(),
                       'losses_single': l_1.cpu().tolist(), 'world': world}, f)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main(sys.argv[1])
