"""Batch-producer measurements: dataset generation on the GPU vs the
host generator, and one training epoch (16 000 paths, batch 100 / 200 / 1000) with the host
collate + H2D copies vs the device collate.  Prints one JSON line per measurement."""
import copy, json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from njode_amd import _lib, data_utils, device_data, train  # noqa: E402


def sync():
    torch.cuda.synchronize()


def main():
    hp = copy.deepcopy(data_utils.hyperparam_default)
    # --- generation
    for n in (20000, 1000000):
        hp['nb_paths'] = n
        for name in ('BlackScholes', 'Heston'):
            device_data.DeviceDataset.generate(name, hp, seed=1)
            sync()
            t0 = time.perf_counter()
            ds = device_data.DeviceDataset.generate(name, hp, seed=2)
            sync()
            el = time.perf_counter() - t0
            bytes_ = ds.paths_tm.numel() * 8 + ds.observed_tm.numel()
            print(json.dumps({'case': 'generate', 'model': name, 'paths': n, 'ms': round(el * 1e3, 3),
                              'paths_per_s': round(n / el), 'write_GBps': round(bytes_ / el / 1e9, 1)}),
                  flush=True)
    hp['nb_paths'] = 20000
    t0 = time.perf_counter()
    paths, obs, nb_obs, meta = data_utils.create_dataset('BlackScholes', hp, seed=0)
    print(json.dumps({'case': 'generate-host-numpy', 'paths': 20000,
                      'ms': round((time.perf_counter() - t0) * 1e3, 1)}), flush=True)
    # --- collate only
    ds = device_data.DeviceDataset.from_arrays(paths, obs, nb_obs, meta)
    for B in (100, 1000, 20000):
        idx = np.random.RandomState(0).permutation(20000)[:B]
        didx = torch.as_tensor(idx, dtype=torch.int32, device='cuda')
        ds.collate(didx)
        sync()
        t0 = time.perf_counter()
        for _ in range(20):
            ds.collate(didx)
        sync()
        dev_ms = (time.perf_counter() - t0) / 20 * 1e3
        t0 = time.perf_counter()
        for _ in range(5):
            b = data_utils.collate_arrays(paths[idx], obs[idx], nb_obs[idx], meta['dt'])
            d = train._device_batch(b, 'cuda')
        sync()
        host_ms = (time.perf_counter() - t0) / 5 * 1e3
        print(json.dumps({'case': 'collate', 'B': B, 'device_ms': round(dev_ms, 3),
                          'host_plus_h2d_ms': round(host_ms, 3)}), flush=True)
    # --- one epoch of the training harness
    for B in (100, 200, 1000):
        row = {'case': 'epoch', 'train_paths': 16000, 'batch': B}
        for dc in (False, True):
            train.train((paths, obs, nb_obs), meta, epochs=1, batch_size=B, log=lambda s: None,
                        device_collate=dc, max_steps_per_epoch=3)          # warm-up
            _, met = train.train((paths, obs, nb_obs), meta, epochs=1, batch_size=B,
                                 log=lambda s: None, device_collate=dc)
            key = 'device_collate' if dc else 'host_collate'
            row[key + '_epoch_s'] = round(met[0][1], 4)
            row[key + '_paths_per_s'] = round(16000 / met[0][1])
        print(json.dumps(row), flush=True)


if __name__ == '__main__':
    main()
