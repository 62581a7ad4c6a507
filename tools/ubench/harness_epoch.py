"""One warm-up epoch + one measured epoch of njode_amd.train.train at batch size argv[1] (device
collate): the process rocprofv3 --kernel-trace --stats is pointed at to see what the GPU does per
harness step (tools/trace_harness.sh)."""
import os, sys
sys.path.insert(0, os.getcwd())
from njode_amd import data_utils, train
B = int(sys.argv[1]) if len(sys.argv) > 1 else 100
hp = dict(data_utils.hyperparam_default, nb_paths=20000)
paths, obs, nb_obs, meta = data_utils.create_dataset('BlackScholes', hp, seed=0)
kw = dict(epochs=2, batch_size=B, log=lambda s: None, device_collate=True)
_, met = train.train((paths, obs, nb_obs), meta, **kw)
print('epoch train_time', [m[1] for m in met])
