"""Debug aid: the gradient of the wave-per-path kernels against the matrix-core tiles' and float64's."""
import os, sys, subprocess
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import numpy as np
name = sys.argv[1] if len(sys.argv) > 1 else 'g5_masked'
if len(sys.argv) > 2:   # child: compute and save
    import torch
    from golden_util import Golden
    from hip_util import hip_model, hip_forward, grads_by_name
    g = Golden(name)
    m = hip_model(g.cfg, g.state_dict()).train()
    hT, loss = hip_forward(m, g.batch(), g.delta_t, g.T)
    loss.backward()
    got = grads_by_name(m)
    np.savez(sys.argv[2], loss=float(loss), hT=hT.detach().cpu().numpy(), **{'grad/' + k: v for k, v in got.items()})
    sys.exit(0)
from golden_util import GOLDEN_DIR, Golden
t = np.load(os.path.join(GOLDEN_DIR, 'g13_f64_truth.npz'))
g = Golden(name)
out = {}
for tag, env in (('chain', {}), ('tiles', {'NJODE_CHAIN_MAX': '0'})):
    f = '/tmp/cvt_%s.npz' % tag
    subprocess.run([sys.executable, __file__, name, f], env=dict(os.environ, **env), check=True, stderr=subprocess.DEVNULL, stdout=subprocess.DEVNULL)
    out[tag] = np.load(f)
for k, ref32 in g.group('grad').items():
    tr = t[name + '/grad/' + k].astype(np.float64)
    c, q = out['chain']['grad/' + k].astype(np.float64), out['tiles']['grad/' + k].astype(np.float64)
    n = np.linalg.norm(tr)
    sc = float((c * tr).sum() / (tr * tr).sum())   # least-squares scale of chain against truth
    sq = float((q * tr).sum() / (tr * tr).sum())
    print('%-30s |chain-f64| %.2e |tiles-f64| %.2e |ref32-f64| %.2e |chain-tiles| %.2e   scale chain %+.2e tiles %+.2e ; after rescale %.2e'
          % (k, np.linalg.norm(c - tr) / n, np.linalg.norm(q - tr) / n, np.linalg.norm(ref32 - tr) / n, np.linalg.norm(c - q) / n,
             sc - 1, sq - 1, np.linalg.norm(c / sc - tr) / n))
print('hT chain-f64 %.2e tiles-f64 %.2e' % (np.abs(out['chain']['hT'] - t[name + '/hT']).max(), np.abs(out['tiles']['hT'] - t[name + '/hT']).max()))
print('loss', out['chain']['loss'], out['tiles']['loss'], float(t[name + '/train_loss']))
