"""Evaluation-path timings (lockstep plan): get_pred / evaluate on the 4 000-path
validation batch of the demo config, as train.py:527-574 / :724 uses them."""
import copy, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from njode_amd import _lib, data_utils, models, stock_model  # noqa: E402

NN = ((50, 'tanh'), (50, 'tanh'))
cfg = dict(input_size=1, hidden_size=10, output_size=1, ode_nn=NN, readout_nn=NN, enc_nn=NN,
           use_rnn=False, bias=True, dropout_rate=0.1, options={'device_outputs': True})
for B in (200, 4000, 20000):
    hp = copy.deepcopy(data_utils.hyperparam_default); hp['nb_paths'] = B
    paths, obs, nb_obs, meta = data_utils.create_dataset('BlackScholes', hp, seed=0)
    b = data_utils.collate_arrays(paths, obs, nb_obs, meta['dt'])
    torch.manual_seed(0)
    m = models.NJODE(**cfg).cuda().eval()
    args = (b['times'], b['time_ptr'], b['X'].cuda(), b['obs_idx'].cuda().int(), meta['dt'],
            meta['maturity'], b['start_X'].cuda())
    def pred():
        with torch.no_grad():
            return m.get_pred(*args)
    for _ in range(2): pred()
    torch.cuda.synchronize(); _lib.profile_enable(True); t0 = time.perf_counter()
    for _ in range(5): pred()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 5
    _lib.profile_enable(False)
    k = {n: round(v[1] / v[0], 4) for n, v in _lib.profile_read().items()}
    print(json.dumps({'case': 'get_pred', 'B': B, 'ms': round(t * 1e3, 3),
                      'paths_per_s': round(B / t, 1), 'kernel_ms': k}), flush=True)
