"""The dropout keep-bit stream of the matrix-core kernels ("nothing tests
the per-unit keep rate or cross-unit independence of that stream").  oracle/dropout_oracle.py
restates the device code in numpy; here its statistics are tested on CPU, and (-m gpu) the
device's words are compared with the restatement bit for bit."""
import numpy as np
import pytest

from oracle import dropout_oracle as do

NET_ODE, NET_ENC = 0, 1
W, P = 50, 0.1


def _keep(n_paths, steps, net=NET_ODE, layer=0, seed=0x1234567890ABCDEF):
    gid = np.arange(n_paths, dtype=np.uint64)[:, None]
    tkey = np.arange(steps, dtype=np.uint64)[None, :]
    return do.keep_units(seed, np.broadcast_to(gid, (n_paths, steps)),
                         np.broadcast_to(tkey, (n_paths, steps)), net, W, P, layer)


def test_keep_rate_per_unit():
    k = _keep(4000, 100).reshape(-1, W).astype(np.float64)      # 400 000 draws per unit
    n = k.shape[0]
    rate = k.mean(axis=0)
    p_keep = 1.0 - int(P * 65536 + 0.5) / 65536.0
    se = np.sqrt(p_keep * (1 - p_keep) / n)
    assert np.all(np.abs(rate - p_keep) < 4.5 * se), (rate.min(), rate.max(), p_keep, se)
    assert abs(rate.mean() - p_keep) < 4.5 * se / np.sqrt(W)


def test_units_are_pairwise_uncorrelated():
    k = _keep(2000, 100).reshape(-1, W).astype(np.float64)
    n = k.shape[0]
    c = np.corrcoef(k.T)
    off = c[~np.eye(W, dtype=bool)]
    # 2 450 pairs, null s.d. 1 / sqrt(n): the largest |rho| of that many should stay below ~4.7 sd
    assert np.max(np.abs(off)) < 5.0 / np.sqrt(n), np.max(np.abs(off)) * np.sqrt(n)
    # joint drops of neighbouring units (same random word: low / high half) at the product rate
    both = ((1 - k[:, 0::2][:, :24]) * (1 - k[:, 1::2][:, :24])).mean()
    pd = int(P * 65536 + 0.5) / 65536.0
    assert abs(both - pd * pd) < 5.0 * np.sqrt(pd * pd * (1 - pd * pd) / (n * 24))


def test_independent_across_steps_paths_layers_networks():
    a = _keep(3000, 64).astype(np.float64)                         # [paths, steps, W]
    n = a.shape[0] * (a.shape[1] - 1) * W
    def corr(x, y):
        x, y = x.reshape(-1), y.reshape(-1)
        return float(np.corrcoef(x, y)[0, 1])
    tol = 5.0 / np.sqrt(n)
    assert abs(corr(a[:, :-1], a[:, 1:])) < tol                    # consecutive Euler steps
    assert abs(corr(a[:-1, :-1], a[1:, :-1])) < tol                # neighbouring paths
    b = _keep(3000, 64, layer=1).astype(np.float64)
    assert abs(corr(a, b)) < 5.0 / np.sqrt(a.size)                 # the two hidden layers
    e = _keep(3000, 64, net=NET_ENC).astype(np.float64)
    assert abs(corr(a, e)) < 5.0 / np.sqrt(a.size)                 # two networks, same key
    s2 = _keep(3000, 64, seed=0x1234567890ABCDF0).astype(np.float64)
    assert abs(corr(a, s2)) < 5.0 / np.sqrt(a.size)                # neighbouring seeds


def test_stream_does_not_depend_on_the_batch_a_path_is_in():
    """keyed by the GLOBAL path id: a path's masks are the same whatever shard holds it"""
    full = _keep(64, 10)
    gid = np.arange(40, 64, dtype=np.uint64)[:, None]
    tkey = np.arange(10, dtype=np.uint64)[None, :]
    part = do.keep_units(0x1234567890ABCDEF, np.broadcast_to(gid, (24, 10)),
                         np.broadcast_to(tkey, (24, 10)), NET_ODE, W, P)
    assert np.array_equal(full[40:], part)


@pytest.mark.gpu
def test_device_words_equal_the_restatement():
    import ctypes
    import torch
    from njode_amd import _lib
    L = _lib.lib()
    seed, n_words = 0x9E3779B97F4A7C15, 14
    keys = [(0, 0, 0), (1, 0, 0), (12345, 7, 0), (2 ** 33 + 5, 99, 1), (19999, 100, 0),
            (3, 0xFFFFFFFF, 1), (77, 0x80000005, 4)]
    out = torch.zeros(4 * n_words, dtype=torch.int32, device='cuda')
    for gid, tkey, net in keys:
        _lib.check(L.njode_selftest_dropout_words(
            ctypes.c_uint64(seed), ctypes.c_uint64(gid), ctypes.c_uint32(tkey), ctypes.c_uint32(net),
            n_words, ctypes.c_void_p(out.data_ptr()), None))
        torch.cuda.synchronize()
        got = out.cpu().numpy().view(np.uint32).reshape(4, n_words)
        for g in range(4):
            st = do.mfma_group_state(seed, np.uint64(gid), g, tkey, net)
            ref = do.xorshift32_words(st, n_words).astype(np.uint32)
            assert np.array_equal(got[g], ref), (gid, tkey, net, g)
