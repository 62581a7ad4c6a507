"""
Data parallelism for the NJ-ODE training step: one process per GPU, the batch sharded
by path, ONE all-reduce of the flat gradient per optimizer step.

The reference has no distributed code; the path
shards embarrassingly because paths interact only through the scalar loss
``sum / batch_size`` and the weight gradient.  Contract (tested on CPU with gloo in
``tests/test_parallel_gloo.py`` and on the GPU by the shard-additivity test):

* rank r owns the contiguous slice ``shard_range(B_global, world, r)`` of every
  global batch, re-collated locally (local path indices, only the observation
  times the shard actually has);
* the loss denominator is the GLOBAL batch size (``model.dp_global_batch``), so
  per-rank losses and gradients are partial sums;
* dropout streams are keyed by the global path id (``model.dp_path_offset``), so
  results do not depend on the number of ranks;
* ONE ``all_reduce(SUM)`` of the bucket [flat gradient (P = 10 071 fp32 = 40 KB for the demo
  model), the step's scalar loss] (latency-bound, no overlap machinery needed because backward
  is three kernels), then the identical Adam step on every rank.  The loss ``loss_and_grad``
  returns is a view of the bucket's last slot: this rank's partial sum until
  ``FusedAdam.step()``, the global loss afterwards.
"""
import os

import numpy as np
import torch
import torch.distributed as dist


def shard_range(n, world, rank):
    """Contiguous, balanced slice [lo, hi) of n items for `rank` of `world`."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_dataset_arrays(stock_paths, observed_dates, nb_obs, world, rank):
    lo, hi = shard_range(len(nb_obs), world, rank)
    return stock_paths[lo:hi], observed_dates[lo:hi], nb_obs[lo:hi], lo


def init_distributed(backend=None):
    """Initialise torch.distributed from the torchrun environment (RANK, WORLD_SIZE,
    LOCAL_RANK, MASTER_ADDR/PORT).  backend 'nccl' is RCCL on ROCm."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend)
    return world, rank, local_rank


def configure_model(model, global_batch, path_offset):
    """Tell the model it holds a shard: loss normalised by the global batch, dropout
    keyed by global path ids."""
    model.dp_global_batch = int(global_batch)
    model.dp_path_offset = int(path_offset)


def empty_shard_step(model, fused=True):
    """A rank whose shard of the global batch is empty (fewer paths than ranks in the last,
    partial batch of an epoch) runs no kernels: its gradient contribution is zero, but it must
    still take part in the gradient all-reduce and the optimizer step, or the other ranks
    hang in the collective.  Returns the rank's (zero) loss contribution."""
    if fused:
        g = model.flat_grad()
        g.zero_()
        if getattr(model, 'dp_loss_in_bucket', False):
            slot = model.loss_slot()      # the loss travels in the gradient's bucket
            slot.zero_()
            return slot.reshape(())
        return torch.zeros((), dtype=torch.float32, device=g.device)
    dev = None
    for p in model.parameters():
        dev = p.device
        if p.grad is None:
            p.grad = torch.zeros_like(p)
        else:
            p.grad.zero_()
    return torch.zeros((), dtype=torch.float32, device=dev)


def allreduce_flat_(flat, group=None):
    """In-place SUM all-reduce of a flat tensor (gradient, or [loss] for logging)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


def broadcast_parameters_(flat, src=0, group=None):
    """Make every rank start from rank `src`'s parameters."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat, src=src, group=group)
    return flat


def epoch_permutation(n, epoch, seed=0):
    """The same shuffled order of the training set on every rank (DataLoader
    shuffle=True equivalent, train.py:250-252), drawn from (seed, epoch)."""
    return np.random.RandomState((seed * 1000003 + epoch) % (2 ** 32)).permutation(n)
