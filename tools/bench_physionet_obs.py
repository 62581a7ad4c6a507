import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from njode_amd import _lib, models, synthetic_physionet
NN = ((50, 'tanh'), (50, 'tanh'))
for name, kw_b in (('few_obs', dict(n_obs_range=(1, 2))), ('default', {}), ('many_obs', dict(n_obs_range=(150, 200)))):
    cfg = dict(input_size=41, hidden_size=41, output_size=41, ode_nn=NN, readout_nn=NN, enc_nn=NN,
               use_rnn=False, bias=True, dropout_rate=0.0, options={'masked': True, 'device_outputs': True})
    b = synthetic_physionet.make_batch(batch_size=50, seed=0, **kw_b)
    torch.manual_seed(0)
    m = models.NJODE(**cfg).cuda().train()
    args = (b['times'], b['time_ptr'], b['X'].cuda(), b['obs_idx'].cuda().int(), b['delta_t'],
            b['T'], b['start_X'].cuda(), b['n_obs_ot'].cuda().int())
    kw = {'M': b['M'].cuda()}
    m.loss_and_grad(*args, **kw)
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    m.loss_and_grad(*args, **kw)
    torch.cuda.synchronize()
    _lib.profile_enable(False)
    k = {n: round(v[1] / v[0], 3) for n, v in _lib.profile_read().items()}
    print(json.dumps({'case': name, 'n_times': len(b['times']), 'n_obs': int(b['time_ptr'][-1]), 'kernel_ms': k}), flush=True)
