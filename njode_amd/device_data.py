"""
GPU-resident datasets and the device-side collate (C ABI: include/njode_producer.h).

Host-side mirror of the reference's batch producer -- ``data_utils.create_dataset``
(``data_utils.py:56-105``: SDE paths + observation mask) and ``custom_collate_fn`` /
``CustomCollateFnGen`` (``data_utils.py:278-316, 352-416``) -- for training loops that run at
GPU speed: the dataset lives in HBM in time-major layout (paths f64 ``[S+1, d, N]``, observed
u8 ``[S+1, N]``), a batch is a device index list, and the only thing that crosses PCIe per
batch is the ``S`` per-time observation counts the host needs to lay out the Euler schedule
(``times`` / ``time_ptr`` stay numpy arrays exactly as ``NJODE.forward`` expects them).

No CPU fallback: everything here calls into ``libnjode_hip.so``.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

_HP_KEYS = ('drift', 'volatility', 'mean', 'speed', 'correlation', 'S0', 'maturity')


def _stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def parse_powers(func_names):
    """``func_appl_X`` names -> lift codes of the C ABI (reference ``data_utils.py:319-335``:
    ``power-k`` -> k, ``exp`` -> 0)."""
    out = []
    for name in func_names or ():
        if name in ('exp', 'exponential'):
            out.append(0)
        elif name.startswith('power-') and float(name.split('-')[1]) == int(float(name.split('-')[1])) \
                and int(float(name.split('-')[1])) >= 1:
            out.append(int(float(name.split('-')[1])))
        else:
            raise ValueError('unsupported func_appl_X entry: {}'.format(name))
    if len(out) > 4:
        raise ValueError('at most 4 func_appl_X entries')
    return out


def times_from_counts(counts, dt):
    """``times`` / ``time_ptr`` of ``custom_collate_fn`` from the per-grid-time observation
    counts (index 0 = grid time 1).  The reference accumulates the clock in float64,
    ``current_time += dt`` (``data_utils.py:296``), so t_k is a running sum, not k * dt."""
    counts = np.asarray(counts, dtype=np.int64)
    clock = np.cumsum(np.full(len(counts), dt, dtype=np.float64))
    used = counts > 0
    time_ptr = np.concatenate([[0], np.cumsum(counts[used])]).astype(np.int64)
    return clock[used], time_ptr


class DeviceDataset:
    """Synthetic dataset resident on one GPU."""

    def __init__(self, paths_tm, observed_tm, nb_obs, metadata):
        self.paths_tm = paths_tm          # f64 [S+1, d, N]
        self.observed_tm = observed_tm    # u8  [S+1, N]
        self.nb_obs = nb_obs              # i32 [N]
        self.metadata = dict(metadata)
        self.n_steps = paths_tm.shape[0] - 1
        self.dim = paths_tm.shape[1]
        self.n_paths = paths_tm.shape[2]
        self.device = paths_tm.device
        self._counts_host = torch.empty(self.n_steps, dtype=torch.int32).pin_memory()

    def __len__(self):
        return self.n_paths

    # -- construction --------------------------------------------------------------------
    @classmethod
    def generate(cls, stock_model_name, hyperparam_dict, seed=0, device='cuda',
                 normals=None, uniforms=None):
        """``create_dataset`` on the GPU.  ``normals`` / ``uniforms`` (numpy f64, reference
        draw order) replace the Philox streams -- used to pin the recurrences to the
        reference's numbers."""
        L = _lib.lib()
        hp = dict(hyperparam_dict)
        dev = torch.device(device)
        dim = int(np.size(hp.get('S0', 1)))
        if dim != 1 and np.ptp(np.asarray(hp['S0'], dtype=np.float64)) != 0:
            raise ValueError('S0 must be the same in every dimension')
        sde = _lib.NjodeSde()
        sde.model = _lib.SDE_MODELS[stock_model_name]
        sde.n_paths, sde.dim, sde.n_steps = int(hp['nb_paths']), dim, int(hp['nb_steps'])
        sc = hp.get('sine_coeff')
        sde.has_sine, sde.sine_coeff = (0, 0.0) if sc is None else (1, float(sc))
        for k in _HP_KEYS:
            v = hp.get(k)
            setattr(sde, k, float(np.ravel(v)[0]) if v is not None else 0.0)
        N, S = sde.n_paths, sde.n_steps
        paths = torch.empty((S + 1, dim, N), dtype=torch.float64, device=dev)
        observed = torch.empty((S + 1, N), dtype=torch.uint8, device=dev)
        nb_obs = torch.empty(N, dtype=torch.int32, device=dev)
        z = u = None
        if normals is not None:
            z = torch.as_tensor(np.ascontiguousarray(normals, dtype=np.float64)).to(dev)
            per = 2 if stock_model_name == 'Heston' else 1
            if z.numel() != N * S * per * dim:
                raise ValueError('normals must have N * S * {} * dim entries'.format(per))
        if uniforms is not None:
            u = torch.as_tensor(np.ascontiguousarray(uniforms, dtype=np.float64)).to(dev)
            if u.numel() != N * (S + 1):
                raise ValueError('uniforms must have N * (S + 1) entries')
        with torch.cuda.device(dev):
            st = _stream(dev)
            _lib.check(L.njode_generate_paths(C.byref(sde), C.c_uint64(seed), _ptr(z), _ptr(paths),
                                              st))
            _lib.check(L.njode_sample_observations(N, S, float(hp['obs_perc']),
                                                   C.c_uint64(seed), _ptr(u), _ptr(observed),
                                                   _ptr(nb_obs), st))
        hp['dt'] = hp['maturity'] / hp['nb_steps']
        hp['model_name'] = stock_model_name
        return cls(paths, observed, nb_obs, hp)

    @classmethod
    def from_arrays(cls, stock_paths, observed_dates, nb_obs, metadata, device='cuda'):
        """Upload a host dataset (``create_dataset`` / ``load_dataset_dir`` arrays:
        paths f64 ``[N, d, S+1]``, observed ``[N, S+1]``) in time-major layout."""
        dev = torch.device(device)
        p = torch.as_tensor(np.ascontiguousarray(np.transpose(stock_paths, (2, 1, 0)),
                                                 dtype=np.float64)).to(dev)
        o = torch.as_tensor(np.ascontiguousarray(observed_dates.T != 0).astype(np.uint8)).to(dev)
        n = torch.as_tensor(np.asarray(nb_obs, dtype=np.int32)).to(dev)
        return cls(p, o, n, metadata)

    def to_arrays(self):
        """Host copy in the reference's layout (paths ``[N, d, S+1]``, observed ``[N, S+1]``)."""
        return (self.paths_tm.permute(2, 1, 0).contiguous().cpu().numpy(),
                self.observed_tm.t().contiguous().cpu().numpy().astype(np.int64),
                self.nb_obs.cpu().numpy().astype(np.int64))

    # -- batches -------------------------------------------------------------------------
    def collate(self, idx=None, func_names=()):
        """The batch ``custom_collate_fn`` builds for dataset rows ``idx`` (device int32
        tensor / array-like in batch order; None = the whole dataset), with ``X``,
        ``start_X``, ``obs_idx`` (int32) and ``n_obs_ot`` (int32) on the device and
        ``times`` / ``time_ptr`` as numpy arrays.

        The per-time counts must reach the host before ``X`` can be sized, so the call waits for
        its own first kernel (``prepare_batches`` + ``fill_batch`` is the form without a wait per
        batch: one host round trip per epoch)."""
        L = _lib.lib()
        dev = self.device
        if idx is not None:
            idx = torch.as_tensor(idx, device=dev).to(torch.int32).contiguous()
            B = idx.numel()
        else:
            B = self.n_paths
        if B == 0:
            raise ValueError('empty batch')
        powers = parse_powers(func_names)
        width = self.dim * (1 + len(powers))
        pw = (C.c_int32 * max(len(powers), 1))(*powers)
        counts = torch.empty(self.n_steps, dtype=torch.int32, device=dev)
        n_obs_ot = torch.empty(B, dtype=torch.int32, device=dev)
        start_X = torch.empty((B, width), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            st = _stream(dev)
            _lib.check(L.njode_collate_count(_ptr(self.observed_tm), _ptr(self.nb_obs),
                                             self.n_paths, self.n_steps, _ptr(idx), B,
                                             _ptr(counts), _ptr(n_obs_ot), st))
            self._counts_host.copy_(counts, non_blocking=True)
            torch.cuda.current_stream(dev).synchronize()
            times, time_ptr = times_from_counts(self._counts_host.numpy(), self.metadata['dt'])
            n_obs = int(time_ptr[-1])
            X = torch.empty((n_obs, width), dtype=torch.float32, device=dev)
            obs_idx = torch.empty(n_obs, dtype=torch.int32, device=dev)
            _lib.check(L.njode_collate_fill(_ptr(self.paths_tm), _ptr(self.observed_tm),
                                            self.n_paths, self.dim, self.n_steps, _ptr(idx), B,
                                            _ptr(counts), pw, len(powers), _ptr(start_X),
                                            _ptr(X) if n_obs else C.c_void_p(0),
                                            _ptr(obs_idx) if n_obs else C.c_void_p(0), st))
        return {'times': times, 'time_ptr': time_ptr, 'obs_idx': obs_idx, 'start_X': start_X,
                'n_obs_ot': n_obs_ot, 'X': X}

    # -- a whole epoch's batches: one host round trip instead of one per batch -----------------
    def prepare_batches(self, idx_list):
        """Phase 1 of the collate (per-time observation counts) for MANY batches at once: the
        count kernels of all batches are enqueued back to back, their results cross PCIe in ONE
        copy and the host waits ONCE -- a training loop at the reference's batch sizes
        (B = 100 / 200) is host-bound, and the per-batch wait for its own count kernel was the
        largest item of its step.  ``idx_list``: dataset rows of each batch (array-likes on the
        host, in batch order).  Returns one descriptor per batch for ``fill_batch``."""
        L = _lib.lib()
        dev = self.device
        sizes = [len(ix) for ix in idx_list]
        if not sizes or min(sizes) <= 0:
            raise ValueError('empty batch')
        flat = np.concatenate([np.asarray(ix, dtype=np.int32) for ix in idx_list])
        idx_dev = torch.as_tensor(flat).to(dev)                       # one upload per epoch
        n = len(sizes)
        counts = torch.empty((n, self.n_steps), dtype=torch.int32, device=dev)
        n_obs_ot = torch.empty(len(flat), dtype=torch.int32, device=dev)
        offs = np.concatenate([[0], np.cumsum(sizes)])
        with torch.cuda.device(dev):
            st = _stream(dev)
            for i in range(n):
                lo, hi = int(offs[i]), int(offs[i + 1])
                _lib.check(L.njode_collate_count(
                    _ptr(self.observed_tm), _ptr(self.nb_obs), self.n_paths, self.n_steps,
                    C.c_void_p(idx_dev.data_ptr() + 4 * lo), hi - lo,
                    C.c_void_p(counts.data_ptr() + 4 * i * self.n_steps),
                    C.c_void_p(n_obs_ot.data_ptr() + 4 * lo), st))
            counts_host = counts.cpu().numpy()                        # one copy, one wait
        out = []
        for i in range(n):
            lo, hi = int(offs[i]), int(offs[i + 1])
            times, time_ptr = times_from_counts(counts_host[i], self.metadata['dt'])
            out.append({'times': times, 'time_ptr': time_ptr, 'idx': idx_dev[lo:hi],
                        'counts': counts[i], 'n_obs_ot': n_obs_ot[lo:hi], 'B': hi - lo})
        return out

    def fill_batch(self, prep, func_names=()):
        """Phase 2 for one descriptor of ``prepare_batches``: two kernel launches, no host
        wait.  Same batch, bit for bit, as ``collate`` builds."""
        L = _lib.lib()
        dev = self.device
        B = prep['B']
        powers = parse_powers(func_names)
        width = self.dim * (1 + len(powers))
        pw = (C.c_int32 * max(len(powers), 1))(*powers)
        n_obs = int(prep['time_ptr'][-1])
        start_X = torch.empty((B, width), dtype=torch.float32, device=dev)
        X = torch.empty((n_obs, width), dtype=torch.float32, device=dev)
        obs_idx = torch.empty(n_obs, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.njode_collate_fill(_ptr(self.paths_tm), _ptr(self.observed_tm),
                                            self.n_paths, self.dim, self.n_steps,
                                            _ptr(prep['idx']), B, _ptr(prep['counts']), pw,
                                            len(powers), _ptr(start_X),
                                            _ptr(X) if n_obs else C.c_void_p(0),
                                            _ptr(obs_idx) if n_obs else C.c_void_p(0),
                                            _stream(dev)))
        return {'times': prep['times'], 'time_ptr': prep['time_ptr'], 'obs_idx': obs_idx,
                'start_X': start_X, 'n_obs_ot': prep['n_obs_ot'], 'X': X}
