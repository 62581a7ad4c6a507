import sys, os
sys.path.insert(0, os.getcwd())
import torch
from njode_amd import models, synthetic_physionet
NN = ((50, 'tanh'), (50, 'tanh'))
cfg = dict(input_size=41, hidden_size=41, output_size=41, ode_nn=NN, readout_nn=NN, enc_nn=NN,
           use_rnn=False, bias=True, dropout_rate=0.0, options={'masked': True, 'device_outputs': True})
b = synthetic_physionet.make_batch(batch_size=50, seed=0, n_obs_range=(30, 100))
torch.manual_seed(0)
m = models.NJODE(**cfg).cuda().train()
args = (b['times'], b['time_ptr'], b['X'].cuda(), b['obs_idx'].cuda().int(), b['delta_t'],
        b['T'], b['start_X'].cuda(), b['n_obs_ot'].cuda().int())
m.loss_and_grad(*args, M=b['M'].cuda())
torch.cuda.synchronize()
