"""Lifetime of what a forward call leaves behind for its backward:
* the plan decision travels with the call (NJODE_C_SCHED_KNOWN): many forwards before the
  first backward -- the pinned schedule ring (16 slots) has long been reused -- still give
  the right gradients;
* a second backward through the same forward (released workspace) raises instead of reading
  a recycled workspace;
* NJODE_VALIDATE=1 (child process: the variable is read once) rejects an out-of-range
  obs_idx, a path twice in one time slice and n_obs_ot == 0 for an observed path."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from golden_util import Golden
from hip_util import GRAD_REL_L2, grads_by_name, hip_forward, hip_model, rel_l2

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def test_backward_after_many_later_forwards():
    g = Golden('g2_bs_grads_B64')
    m = hip_model(g.cfg, g.state_dict()).train()
    b = g.batch()
    _, loss0 = hip_forward(m, b, g.delta_t, g.T)            # segment plan, saved for later
    # 24 other forwards (lockstep plan: until_T with a tail; other schedules) recycle every
    # slot of the pinned ring before loss0 is back-propagated
    keep = []
    for i in range(24):
        _, li = hip_forward(m, b, g.delta_t, g.T + 0.01 * (i + 1), until_T=True)
        keep.append(li)
    loss0.backward()
    got = grads_by_name(m)
    for k, ref in g.group('grad').items():
        assert rel_l2(got[k], ref) < GRAD_REL_L2, (k, rel_l2(got[k], ref))
    # ... and the lockstep forwards are still differentiable afterwards
    m.zero_grad()
    keep[-1].backward()
    assert float(keep[-1]) == pytest.approx(float(g['train_loss']), rel=1e-4)


def test_second_backward_raises():
    g = Golden('g2_bs_grads_B64')
    m = hip_model(g.cfg, g.state_dict()).train()
    _, loss = hip_forward(m, g.batch(), g.delta_t, g.T)
    loss.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match='second backward'):
        loss.backward()


_CHILD = r'''
import sys
import numpy as np, torch
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
from golden_util import Golden
from hip_util import hip_forward, hip_model
from njode_amd import _lib
g = Golden('g1_bs_eval_B7')
m = hip_model(g.cfg, g.state_dict()).eval()
def run(b):
    with torch.no_grad():
        return hip_forward(m, b, g.delta_t, g.T)
run(g.batch())                                          # a valid batch passes
out = []
b = g.batch(); b['obs_idx'] = b['obs_idx'].clone(); b['obs_idx'][3] = 7      # B = 7: out of range
try: run(b); out.append('no error')
except _lib.NjodeError as e: out.append(str(e))
b = g.batch(); tp = b['time_ptr']; i = int(np.argmax(np.diff(tp) >= 2)); lo = int(tp[i])
b['obs_idx'] = b['obs_idx'].clone(); b['obs_idx'][lo + 1] = b['obs_idx'][lo]   # duplicate in a slice
try: run(b); out.append('no error')
except _lib.NjodeError as e: out.append(str(e))
b = g.batch(); b['n_obs_ot'] = b['n_obs_ot'].clone(); b['n_obs_ot'][int(b['obs_idx'][0])] = 0
try: run(b); out.append('no error')
except _lib.NjodeError as e: out.append(str(e))
print('|'.join(out))
'''


def test_validate_mode_rejects_bad_batches():
    env = dict(os.environ, NJODE_VALIDATE='1')
    code = _CHILD.format(root=os.path.dirname(HERE), tests=HERE)
    p = subprocess.run([sys.executable, '-c', code], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    msgs = p.stdout.strip().splitlines()[-1].split('|')
    assert len(msgs) == 3
    assert 'outside [0, batch_size)' in msgs[0]
    assert 'two rows in one time slice' in msgs[1]
    assert 'n_obs_ot <= 0' in msgs[2]
