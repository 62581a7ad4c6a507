"""
TEST INFRASTRUCTURE (oracle/): numpy restatement of the dropout keep-bit stream of the
matrix-core kernels (njode_amd/csrc/njode_device.h: fmix32, drop_state; njode_mfma.h:
keep_bits).  Only tests/ may import it.

The reference draws its masks with torch's bernoulli_ (models.py:148-160, nn.Dropout), which a
GPU kernel cannot reproduce bit for bit; parity of the dropout path is therefore statistical
(SURVEY.md section 7).  This file pins WHAT the kernels draw, so that the stream itself can be
tested: keep rate per unit, independence across units / Euler steps / paths / networks.

Stream of one network evaluation for lane group g (the lanes holding units 4q + g of the
wave's 16 chains):  state = drop_state(seed, gid, gid_hi + 0x5bd1e995 (g + 1), tkey, net);
word i = i-th xorshift32 (13, 17, 5) output; unit 4q + g of hidden layer 1 is kept iff the
(q & 1 ? high : low) 16 bits of word q >> 1 are >= thr16 = round(p 65536); layer 2 continues
the stream after ceil(Q / 2) words.
"""
import numpy as np

M32 = np.uint64(0xFFFFFFFF)


def _u32(x):
    return np.asarray(x, dtype=np.uint64) & M32


def fmix32(h):
    h = _u32(h)
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85ebca6b)) & M32
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xc2b2ae35)) & M32
    h ^= h >> np.uint64(16)
    return h


def drop_state(seed, gid_lo, gid_hi, tkey, net):
    seed_lo, seed_hi = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
    gid_lo, gid_hi, tkey, net = _u32(gid_lo), _u32(gid_hi), _u32(tkey), _u32(net)
    h = fmix32(seed_lo ^ ((gid_lo * np.uint64(0x9e3779b9)) & M32))
    h = fmix32(h ^ seed_hi ^ ((gid_hi * np.uint64(0x7f4a7c15)) & M32)
               ^ ((tkey * np.uint64(0x85ebca6b)) & M32))
    h = fmix32(h ^ ((net * np.uint64(0xc2b2ae35)) & M32) ^ np.uint64(0x27d4eb2f))
    return np.where(h == 0, np.uint64(0x9e3779b9), h)


def xorshift32_words(state, n):
    """n successive outputs of xorshift32 (13, 17, 5) for every state (vectorised);
    returns [n, ...]."""
    s = _u32(state).copy()
    out = []
    for _ in range(n):
        s ^= (s << np.uint64(13)) & M32
        s ^= s >> np.uint64(17)
        s ^= (s << np.uint64(5)) & M32
        out.append(s.copy())
    return np.stack(out)


def mfma_group_state(seed, gid, g, tkey, net):
    gid = np.asarray(gid, dtype=np.uint64)
    hi = ((gid >> np.uint64(32)) + np.uint64(0x5bd1e995) * np.uint64(g + 1)) & M32
    return drop_state(seed, gid & M32, hi, tkey, net)


def keep_units(seed, gid, tkey, net, width, p, layer=0):
    """0/1 keep indicators [..., width] of hidden layer `layer` (0 or 1) of one network
    evaluation, as the matrix-core kernels draw them."""
    thr = np.uint64(int(p * 65536.0 + 0.5))
    q_regs = (width + 1 + 3) // 4               # registers per lane (units + bias unit)
    n_words = (q_regs + 1) // 2
    gid = np.asarray(gid, dtype=np.uint64)
    keep = np.zeros(gid.shape + (width,), dtype=np.uint8)
    for g in range(4):
        st = mfma_group_state(seed, gid, g, tkey, net)
        words = xorshift32_words(st, 2 * n_words)[layer * n_words:(layer + 1) * n_words]
        for q in range(q_regs):
            u = 4 * q + g
            if u >= width:
                continue
            w = words[q >> 1]
            bits = (w >> np.uint64(16)) if (q & 1) else (w & np.uint64(0xFFFF))
            keep[..., u] = (bits >= thr).astype(np.uint8)
    return keep
