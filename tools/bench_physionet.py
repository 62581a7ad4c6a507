"""Per-kernel timing of the PhysioNet-shaped masked training step (BASELINE config 5)."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from njode_amd import _lib, models, synthetic_physionet  # noqa: E402
NN = ((50, 'tanh'), (50, 'tanh'))
for B in [int(x) for x in os.environ.get('BATCHES', '50,800').split(',')]:
    cfg = dict(input_size=41, hidden_size=41, output_size=41, ode_nn=NN, readout_nn=NN, enc_nn=NN,
               use_rnn=False, bias=True, dropout_rate=float(os.environ.get('DROPOUT', '0.0')),
               options={'masked': True, 'device_outputs': True})
    b = synthetic_physionet.make_batch(batch_size=B, seed=0)
    torch.manual_seed(0)
    m = models.NJODE(**cfg).cuda().train()
    args = (b['times'], b['time_ptr'], b['X'].cuda(), b['obs_idx'].cuda().int(), b['delta_t'],
            b['T'], b['start_X'].cuda(), b['n_obs_ot'].cuda().int())
    kw = {'M': b['M'].cuda()}
    m.loss_and_grad(*args, **kw)
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    t0 = time.perf_counter()
    m.loss_and_grad(*args, **kw)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    _lib.profile_enable(False)
    k = {n: round(v[1] / v[0], 3) for n, v in _lib.profile_read().items()}
    print(json.dumps({'case': 'config5-kernels', 'dropout': float(os.environ.get('DROPOUT', '0.0')), 'B': B, 'n_times': len(b['times']),
                      'n_obs': int(b['time_ptr'][-1]), 'step_ms': round(el * 1e3, 2),
                      'kernel_ms': k}), flush=True)
