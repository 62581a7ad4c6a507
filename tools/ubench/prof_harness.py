import cProfile, pstats, sys, os, io
sys.path.insert(0, os.getcwd())
import torch
from njode_amd import data_utils, train
hp = dict(data_utils.hyperparam_default, nb_paths=20000)
paths, obs, nb_obs, meta = data_utils.create_dataset('BlackScholes', hp, seed=0)
kw = dict(epochs=1, batch_size=100, log=lambda s: None, device_collate=True)
train.train((paths, obs, nb_obs), meta, **kw)   # warm
pr = cProfile.Profile()
pr.enable()
train.train((paths, obs, nb_obs), meta, **kw)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(28)
print(s.getvalue()[:6000])
