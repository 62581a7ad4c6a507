"""Debug aid: per-tensor gradient error of the masked fixture g5_masked (HIP vs the reference's)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np, torch
from golden_util import Golden
from hip_util import hip_model, hip_forward, grads_by_name, rel_l2
name = sys.argv[1] if len(sys.argv) > 1 else 'g5_masked'
g = Golden(name)
m = hip_model(g.cfg, g.state_dict()).train()
hT, loss = hip_forward(m, g.batch(), g.delta_t, g.T)
loss.backward()
torch.cuda.synchronize()
print('loss', float(loss), float(g['train_loss']))
got = grads_by_name(m)
for k, ref in g.group('grad').items():
    print(k, 'rel_l2 %.3e' % rel_l2(got[k], ref), 'norm got %.3e ref %.3e' % (np.linalg.norm(got[k]), np.linalg.norm(ref)))
