"""CPU tests of the batch producer's host logic and of the Philox restatement that checks the
device random streams (pinned to the algorithm's published known-answer vectors)."""
import os
import re

import numpy as np
import pytest

from njode_amd import _lib, data_utils, device_data
from oracle import producer_oracle as po

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# Random123 kat_vectors, philox4x32-10: (counter, key, expected)
KAT = [
    ((0x00000000,) * 4, (0x00000000,) * 2, (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def test_philox_restatement_matches_known_answer_vectors():
    for ctr, key, want in KAT:
        got = po.philox4x32_10(np.array([ctr], dtype=np.uint32), np.array([key], dtype=np.uint32))
        assert tuple(int(x) for x in got[0]) == want


def test_u53_is_numpys_double_recipe():
    a = np.array([0, 0xffffffff, 0x12345678], dtype=np.uint32)
    b = np.array([0, 0xffffffff, 0x9abcdef0], dtype=np.uint32)
    want = ((a >> 5).astype(np.float64) * 67108864.0 + (b >> 6)) / 9007199254740992.0
    np.testing.assert_array_equal(po.u53(a, b), want)
    assert po.u53(a, b).max() < 1.0


def test_oracle_streams_have_the_right_moments():
    u = po.observation_uniforms(2000, 100, seed=7)
    assert u.shape == (2000, 101) and 0.0 <= u.min() and u.max() < 1.0
    assert abs(u.mean() - 0.5) < 4 * np.sqrt(1 / 12 / u.size)
    z1, z2 = po.path_normals(500, 100, 1, seed=7)
    for z in (z1, z2):
        assert abs(z.mean()) < 4 / np.sqrt(z.size)
        assert abs(z.var() - 1.0) < 4 * np.sqrt(2.0 / z.size)
    assert abs(np.mean(z1 * z2)) < 4 / np.sqrt(z1.size)
    zs = po.step_normals(50, 7, 2, seed=3)          # odd step count: the last pair is half used
    a, b = po.path_normals(50, 4, 2, seed=3)
    assert zs.shape == (50, 7, 2)
    np.testing.assert_array_equal(zs[:, 0::2], a)
    np.testing.assert_array_equal(zs[:, 1::2], b[:, :3])


def test_times_from_counts_reproduces_the_collate_clock():
    paths, observed, nb_obs, hp = data_utils.create_dataset(
        'BlackScholes', dict(data_utils.hyperparam_default, nb_paths=23, nb_steps=40), seed=3)
    observed[:, 7] = 0            # a grid time nobody observes
    ref = data_utils.collate_arrays(paths, observed, observed[:, 1:].sum(1), hp['dt'])
    counts = observed[:, 1:].sum(0)
    times, time_ptr = device_data.times_from_counts(counts, hp['dt'])
    np.testing.assert_array_equal(times, ref['times'])      # bit-exact float64 clock
    np.testing.assert_array_equal(time_ptr, ref['time_ptr'])


def test_parse_powers():
    assert device_data.parse_powers(None) == []
    assert device_data.parse_powers(['power-2', 'power-3', 'exp']) == [2, 3, 0]
    with pytest.raises(ValueError):
        device_data.parse_powers(['power-0.5'])


def test_every_producer_symbol_is_declared_and_exported():
    header = open(os.path.join(REPO, 'include', 'njode_producer.h')).read()
    declared = set(re.findall(r'\b(njode_[a-z0-9_]+)\s*\(', header))
    assert declared and declared <= set(_lib.EXPORTS)
    lib = _lib.lib()
    for name in declared:
        assert hasattr(lib, name), name
