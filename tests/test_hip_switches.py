"""Round-5 switches of the segment plan that are NOT the default but ship in the library
(DESIGN.md section 4a): the backward's tile queue (`NJODE_BWD_QUEUE=1`) and the encoder evaluated
at the head of every item inside the ODE forward (`NJODE_ENC_FUSED=1`).  Both are performance
decisions only; a training step must give the same loss and gradient with and without them --
the fused encoder bit for bit (same instructions, same dropout words), the queue to fp32 summation
order (which tiles meet in one accumulator depends on timing).  The switches are read once per
process, so every variant runs in a child process, at a batch size that takes the mixed kernels
(more than 768 tiles of 16 items)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from hip_util import rel_l2

pytestmark = pytest.mark.gpu

TESTS = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(TESTS)

_SNIPPET = r'''
import sys
sys.path.insert(0, {tests!r}); sys.path.insert(0, {repo!r})
import numpy as np, torch
from hip_util import bs_batch, demo_cfg, hip_model
b, meta = bs_batch(1600, seed=5)
assert int(b['time_ptr'][-1]) > 768 * 16          # the mixed ODE kernels, both roles
torch.manual_seed(0)
m = hip_model(demo_cfg(dropout=0.1)).train()
args = (b['times'], b['time_ptr'], b['X'].cuda(), b['obs_idx'].cuda().int(), meta['dt'], meta['maturity'],
        b['start_X'].cuda(), b['n_obs_ot'].cuda().int())
out = []
for step in range(3):                              # (the queue's counters must be clean again each time)
    m._step_counter = 11 + step
    _, loss = m.loss_and_grad(*args)
    out.append(np.concatenate([[float(loss)], m.flat_grad().cpu().numpy().astype(np.float64)]))
np.save({out!r}, np.stack(out))
'''


def _run(tmp_path, tag, env_extra):
    out = str(tmp_path / (tag + '.npy'))
    # (NJODE_SEG_CHAIN_MAX=0: the mixed matrix-core kernels these switches belong to, not the wave-per-item
    # kernels that a batch of this size runs by default since round 6)
    env = dict(os.environ, NJODE_SEG_CHAIN_MAX='0', **env_extra)
    p = subprocess.run([sys.executable, '-c', _SNIPPET.format(tests=TESTS, repo=REPO, out=out)], env=env,
                       cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:]
    return np.load(out)


def test_tile_queue_and_fused_encoder_do_not_change_the_step(tmp_path):
    ref = _run(tmp_path, 'default', {'NJODE_BWD_QUEUE': '0', 'NJODE_ENC_FUSED': '0'})
    assert np.isfinite(ref).all() and (np.abs(ref[:, 0]) > 0).all()
    # the static rounds are bitwise reproducible
    again = _run(tmp_path, 'default2', {'NJODE_BWD_QUEUE': '0', 'NJODE_ENC_FUSED': '0'})
    assert np.array_equal(ref, again)
    fused = _run(tmp_path, 'fused', {'NJODE_BWD_QUEUE': '0', 'NJODE_ENC_FUSED': '1'})
    assert np.array_equal(ref, fused)              # same instructions, same dropout words
    queue = _run(tmp_path, 'queue', {'NJODE_BWD_QUEUE': '1', 'NJODE_ENC_FUSED': '0'})
    assert np.array_equal(queue[:, 0], ref[:, 0])  # the loss is the forward's: untouched
    for s in range(ref.shape[0]):
        assert rel_l2(queue[s, 1:], ref[s, 1:]) < 1e-5, (s, rel_l2(queue[s, 1:], ref[s, 1:]))


# ---- the one-launch plan (njode_plan.h) ---------------------------------------------------------
# Same arrays as the multi-launch plan (row times, links, the STABLE order by length, the trajectory
# layout, the split points): the tiles are the same rows in the same lanes, so with the static rounds
# a training step and a prediction call give the same bits whether the plan is
#   defer   built by the first blocks of the previous step's ODE-forward launch (the default of
#           prefetch_plan: NJODE_C_PLAN_DEFER), or by k_plan_grid in line when nothing was prefetched,
#   side    built on the helper stream (NJODE_PLAN_DEFER=0), in line by k_plan_grid for small plans,
#   legacy  built by the multi-launch kernels everywhere (NJODE_PLAN_GRID=0).
_SNIPPET_PLAN = r'''
import sys
sys.path.insert(0, {tests!r}); sys.path.insert(0, {repo!r})
import numpy as np, torch
from hip_util import bs_batch, demo_cfg, hip_model
out = []
def dev_args(b, meta):
    return (b['times'], b['time_ptr'], b['X'].cuda(), b['obs_idx'].cuda().int(), meta['dt'], meta['maturity'],
            b['start_X'].cuda(), b['n_obs_ot'].cuda().int())
for B, seed, perc in ((7, 3, 0.1), (100, 5, 0.1), (150, 6, 0.02), (1000, 7, 0.1), (1600, 8, 0.1), (6000, 9, 0.1)):
    batches = [dev_args(*bs_batch(B, seed=seed + 100 * i, obs_perc=perc)) for i in range(2)]   # (0.02: paths without a row)
    torch.manual_seed(0)
    m = hip_model(demo_cfg(dropout=0.1)).train()
    m.prefetch_plan(*batches[0], need_hT=False)
    for step in range(4):
        m._step_counter = 11 + step
        m.prefetch_plan(*batches[(step + 1) % 2], need_hT=False)      # the next step's plan, ahead
        _, loss = m.loss_and_grad(*batches[step % 2])
        out.append(np.concatenate([[float(loss)], m.flat_grad().cpu().numpy().astype(np.float64)]))
    m._plans.clear()
    m._step_counter = 21
    _, loss = m.loss_and_grad(*batches[1])                            # nothing prefetched: plan in line
    out.append(np.concatenate([[float(loss)], m.flat_grad().cpu().numpy().astype(np.float64)]))
    m.eval()
    with torch.no_grad():
        m.prefetch_plan(*batches[0])                                  # with the tail order (hT)
        hT1, loss1 = m(*batches[1])                                   # ... hosted by another batch's call
        hT0, loss0 = m(*batches[0])
    for hT, loss in ((hT1, loss1), (hT0, loss0)):
        out.append(np.concatenate([[float(loss)], hT.cpu().numpy().astype(np.float64).ravel()]))
np.save({out!r}, np.concatenate(out))
'''


def test_one_launch_plan_is_the_same_plan(tmp_path):
    res = {}
    big = {'NJODE_PLAN_DEFER': '1', 'NJODE_PLAN_DEFER_MAX': '1000000'}   # (also the 60 000-row batch)
    for tag, env in (('defer', dict(big, NJODE_PLAN_GRID='1')),
                     ('default', {}),
                     ('side', {'NJODE_PLAN_DEFER': '0', 'NJODE_PLAN_GRID': '1'}),
                     ('legacy', {'NJODE_PLAN_DEFER': '0', 'NJODE_PLAN_GRID': '0'}),
                     ('defer_p3', dict(big, NJODE_PLAN_BLOCKS='3')),
                     ('defer_p200', dict(big, NJODE_PLAN_BLOCKS='200'))):
        out = str(tmp_path / (tag + '.npy'))
        p = subprocess.run([sys.executable, '-c', _SNIPPET_PLAN.format(tests=TESTS, repo=REPO, out=out)],
                           env=dict(os.environ, **env), cwd=REPO, stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True, timeout=900)
        assert p.returncode == 0, p.stdout[-3000:]
        res[tag] = np.load(out)
    assert np.isfinite(res['legacy']).all()
    for tag in res:
        assert np.array_equal(res[tag], res['legacy']), tag
