"""Where the generic forward's time goes: loss-only forward (eval mode, nothing stored) of a
width-100 model at B = 100 on Black-Scholes batches with almost no / the usual / all grid times
observed.  Prints ms per call."""
import contextlib, json, os, sys, time
import torch
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
from hip_util import bs_batch
from njode_amd import models
nn = ((100, 'tanh'), (100, 'tanh'))
cfg = dict(input_size=1, hidden_size=10, output_size=1, ode_nn=nn, readout_nn=nn, enc_nn=nn, use_rnn=False,
           bias=True, dropout_rate=0.0, options={'device_outputs': True})
with contextlib.redirect_stdout(sys.stderr):
    m = models.NJODE(**cfg).cuda().eval()
for p in (0.0005, 0.1):
    b, meta = bs_batch(100, seed=1, obs_perc=p)
    args = (b['times'], b['time_ptr'], b['X'].cuda(), b['obs_idx'].cuda().int(), meta['dt'], meta['maturity'],
            b['start_X'].cuda(), b['n_obs_ot'].clamp(min=1).cuda().int())
    with torch.no_grad():
        for _ in range(3):
            m(*args)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            m(*args)
        torch.cuda.synchronize()
    rec = {'obs_perc': p, 'n_times': len(b['times']), 'n_obs': int(b['time_ptr'][-1]),
           'fwd_ms': round((time.perf_counter() - t0) * 100, 3)}
    from njode_amd import _lib
    L = _lib.lib()
    if hasattr(L, 'njode_gen_debug_stamps'):      # library built with -DNJ_GEN_STAMPS
        import ctypes
        buf = (ctypes.c_uint64 * 16)()
        L.njode_gen_debug_stamps(buf)
        n_ev = 13 * 100                            # 13 calls x 100 Euler steps (ODE-step stamps)
        if buf[15]:                                # segment plan: steps of block 0 (the longest tile), counted
            n_ev = int(buf[15])
            rec['steps_of_block0_per_call'] = n_ev / 13
        names = {0: 'ode_input+sync', 1: 'net_forward', 2: 'update+sync', 4: 'layer: bias/rec/sync',
                 5: 'layer0 product', 6: 'layer1 product', 7: 'layer2 product', 8: 'layer: end sync'}
        rec['cycles_per_euler_step'] = {names[i]: round(buf[i] / n_ev) for i in names}
    print(json.dumps(rec), flush=True)
