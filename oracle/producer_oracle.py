"""
CPU restatement of the GPU batch producer's random streams -- TEST INFRASTRUCTURE ONLY
(imported by tests/, never by the product path).

The reference draws from numpy's legacy Mersenne-Twister stream (``data_utils.py:73-81``),
which a GPU kernel cannot reproduce; the producer uses Philox4x32-10 instead.  This module
restates that generator (Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy as
1, 2, 3", SC'11; the Random123 library's philox4x32-10) in numpy so the device stream can be
checked word for word, and is itself pinned to the algorithm's published known-answer
vectors (Random123 ``kat_vectors``) in tests/test_producer_oracle.py.

The SDE recurrences and the collate have no separate restatement here: the host versions in
``njode_amd/stock_model.py`` / ``njode_amd/data_utils.py`` are already pinned to the
reference's outputs by tests/golden (G4) and serve as the checker for the device kernels.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
STREAM_PATHS, STREAM_OBS = 0x70617468, 0x6f627376
MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(ctr, key):
    """ctr uint32 [n, 4], key uint32 [n, 2] -> uint32 [n, 4]."""
    c = np.array(ctr, dtype=np.uint32).reshape(-1, 4).copy()
    k = np.array(key, dtype=np.uint32).reshape(-1, 2).copy()
    with np.errstate(over='ignore'):
        for _ in range(10):
            p0 = M0 * c[:, 0].astype(np.uint64)
            p1 = M1 * c[:, 2].astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & MASK32).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & MASK32).astype(np.uint32)
            c = np.stack([hi1 ^ c[:, 1] ^ k[:, 0], lo1, hi0 ^ c[:, 3] ^ k[:, 1], lo0], axis=1)
            k = np.stack([k[:, 0] + W0, k[:, 1] + W1], axis=1)
    return c


def u53(a, b):
    """53-bit uniform in [0, 1) from two 32-bit words (numpy's ``random_double`` recipe)."""
    a = np.asarray(a, dtype=np.uint32).astype(np.float64)
    b = np.asarray(b, dtype=np.uint32).astype(np.float64)
    return (np.floor(a / 32.0) * 67108864.0 + np.floor(b / 64.0)) / 9007199254740992.0


def observation_uniforms(n_paths, n_steps, seed):
    """The uniforms njode_sample_observations draws: f64 [N, S+1]."""
    n = np.arange(n_paths, dtype=np.uint32)
    out = np.empty((n_paths, n_steps + 2), dtype=np.float64)
    lo, hi = seed & 0xFFFFFFFF, ((seed >> 32) & 0xFFFFFFFF) ^ STREAM_OBS
    for t2 in range((n_steps + 2) // 2):
        ctr = np.stack([n, np.zeros_like(n), np.full_like(n, t2), np.zeros_like(n)], axis=1)
        key = np.tile(np.array([[lo, hi]], dtype=np.uint32), (n_paths, 1))
        r = philox4x32_10(ctr, key)
        out[:, 2 * t2] = u53(r[:, 0], r[:, 1])
        out[:, 2 * t2 + 1] = u53(r[:, 2], r[:, 3])
    return out[:, :n_steps + 1]


def step_normals(n_paths, n_steps, dim, seed):
    """The per-step normals of the Black-Scholes / OU generators: the Box-Muller pair of
    counter (path, m, dim) feeds steps 2m - 1 (cos branch) and 2m (sin branch); f64
    [N, S, dim]."""
    m = (n_steps + 1) // 2
    z1, z2 = path_normals(n_paths, m, dim, seed)
    z = np.empty((n_paths, 2 * m, dim))
    z[:, 0::2, :] = z1
    z[:, 1::2, :] = z2
    return z[:, :n_steps, :]


def path_normals(n_paths, n_steps, dim, seed):
    """Box-Muller pairs of the counters (path, 1..n_steps, dim): (z1, z2) f64 [N, S, dim] each
    (Heston consumes one pair per step)."""
    lo, hi = seed & 0xFFFFFFFF, ((seed >> 32) & 0xFFFFFFFF) ^ STREAM_PATHS
    n, k, j = np.meshgrid(np.arange(n_paths, dtype=np.uint32),
                          np.arange(1, n_steps + 1, dtype=np.uint32),
                          np.arange(dim, dtype=np.uint32), indexing='ij')
    ctr = np.stack([n.ravel(), np.zeros(n.size, dtype=np.uint32), k.ravel(), j.ravel()], axis=1)
    key = np.tile(np.array([[lo, hi]], dtype=np.uint32), (n.size, 1))
    r = philox4x32_10(ctr, key)
    u1 = 1.0 - u53(r[:, 0], r[:, 1])
    u2 = u53(r[:, 2], r[:, 3])
    rad = np.sqrt(-2.0 * np.log(u1))
    z1 = (rad * np.cos(2.0 * np.pi * u2)).reshape(n_paths, n_steps, dim)
    z2 = (rad * np.sin(2.0 * np.pi * u2)).reshape(n_paths, n_steps, dim)
    return z1, z2
