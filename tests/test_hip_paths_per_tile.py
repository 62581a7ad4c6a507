"""Paths per tile of the masked lockstep kernels (round 4).  The specialised kernels
(njode_mfma_lock4.h, `NJODE_LOCK4_PT`) and the shape-generic ones (njode_gen.h, `NJODE_GEN_PT`)
take the number of paths a 16-lane tile holds at run time -- 16 / 8 / 4 / 2 / 1, chosen from the
batch size; the other lanes idle.  The choice is a performance decision only: with dropout ON
(masks are keyed by the path, not by the tile) a training step must give the same loss, hT and
gradient whatever it is, up to fp32 summation order, and the keep bits drawn ahead of the forward
(`NJODE_DROP_BITS_AHEAD=0`: inside it) must be the same bits.  The switches are read once per
process, so every variant runs in a child process."""
import os
import subprocess
import sys

import numpy as np
import pytest

from hip_util import rel_l2

pytestmark = pytest.mark.gpu

TESTS = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(TESTS)

_SNIPPET = r'''
import sys
sys.path.insert(0, {tests!r}); sys.path.insert(0, {repo!r})
import numpy as np, torch
from njode_amd import models, synthetic_physionet
NN = (({width}, 'tanh'), ({width}, 'tanh'))
cfg = dict(input_size=41, hidden_size=41, output_size=41, ode_nn=NN, readout_nn=NN, enc_nn=NN,
           use_rnn=False, bias=True, dropout_rate=0.1, options={{'masked': True, 'device_outputs': True}})
b = synthetic_physionet.make_batch(batch_size=37, n_grid=60, n_obs_range=(3, 9), seed=3)
torch.manual_seed(0)
m = models.NJODE(**cfg).cuda().train()
m._step_counter = 7
args = (b['times'], b['time_ptr'], b['X'].cuda(), b['obs_idx'].cuda().int(), b['delta_t'], b['T'],
        b['start_X'].cuda(), b['n_obs_ot'].cuda().int())
hT, loss = m.loss_and_grad(*args, M=b['M'].cuda())
np.save({out!r}, np.concatenate([[float(loss)], hT.cpu().numpy().reshape(-1).astype(np.float64),
                                 m.flat_grad().cpu().numpy().astype(np.float64)]))
'''


def _run(tmp_path, tag, width, env_extra):
    out = str(tmp_path / (tag + '.npy'))
    env = dict(os.environ, **env_extra)
    p = subprocess.run([sys.executable, '-c', _SNIPPET.format(tests=TESTS, repo=REPO, out=out, width=width)],
                       env=env, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:]
    return np.load(out)


@pytest.mark.parametrize('family,width,var', [('specialised', 50, 'NJODE_LOCK4_PT'), ('generic', 72, 'NJODE_GEN_PT')])
def test_training_step_does_not_depend_on_the_paths_per_tile(tmp_path, family, width, var):
    # (NJODE_CHAIN_MAX=0: the matrix-core tiles, not the wave-per-path kernels that a batch of this
    # size runs by default since round 6 -- those are compared with the tiles below)
    ref = _run(tmp_path, 'pt16', width, {var: '16', 'NJODE_CHAIN_MAX': '0'})
    n_h = 37 * 41
    assert np.isfinite(ref).all() and abs(ref[0]) > 0
    for pt in ('4', '1', '0'):                      # ('0': the library's own choice)
        got = _run(tmp_path, 'pt' + pt, width, {var: pt, 'NJODE_CHAIN_MAX': '0'})
        assert got[0] == pytest.approx(ref[0], rel=2e-5), (family, pt)
        np.testing.assert_allclose(got[1:1 + n_h], ref[1:1 + n_h], atol=2e-5, rtol=1e-4)
        assert rel_l2(got[1 + n_h:], ref[1 + n_h:]) < 1e-4, (family, pt)


def test_keep_bits_drawn_ahead_are_the_bits_drawn_in_the_kernel(tmp_path):
    """specialised masked forward: k_q4_bits against q4_ode_keep / q4_row_keep -- the same masks,
    so the same loss, hT and gradient BIT FOR BIT (one path per tile in both runs)."""
    a = _run(tmp_path, 'ahead', 50, {'NJODE_LOCK4_PT': '1', 'NJODE_DROP_BITS_AHEAD': '1', 'NJODE_CHAIN_MAX': '0'})
    b = _run(tmp_path, 'inside', 50, {'NJODE_LOCK4_PT': '1', 'NJODE_DROP_BITS_AHEAD': '0', 'NJODE_CHAIN_MAX': '0'})
    assert np.array_equal(a, b)


def test_wave_per_path_kernels_draw_the_masks_of_the_matrix_core_tiles(tmp_path):
    """round 6, njode_chain.h: one wave per path with a lane per unit against one 16-lane tile over four
    waves (njode_mfma_lock4.h), dropout ON.  k_chain_bits re-assembles the lane-group streams into
    64-bit lane masks, so both runs drop the SAME units: loss, hT and the gradient agree to fp32
    summation order -- a wrong bit anywhere would show as a 1e-2 difference.  Blocks of one and of
    four waves (NJODE_CHAIN_WPB) must give the same numbers bit for bit: the waves of a block share
    nothing but read-only LDS tables."""
    tiles = _run(tmp_path, 'tiles', 50, {'NJODE_CHAIN_MAX': '0'})
    chain = _run(tmp_path, 'chain', 50, {})
    n_h = 37 * 41
    assert np.isfinite(chain).all() and abs(chain[0]) > 0
    assert chain[0] == pytest.approx(tiles[0], rel=2e-5)
    np.testing.assert_allclose(chain[1:1 + n_h], tiles[1:1 + n_h], atol=2e-5, rtol=1e-4)
    assert rel_l2(chain[1 + n_h:], tiles[1 + n_h:]) < 1e-4
    for wpb in ('1', '4', '8'):
        got = _run(tmp_path, 'chain_wpb' + wpb, 50, {'NJODE_CHAIN_WPB': wpb})
        assert np.array_equal(got[:1 + n_h], chain[:1 + n_h]), wpb
        # (the gradient's slab reduction does not depend on the block shape of the sweep either)
        assert np.array_equal(got, chain), wpb


_SNIPPET_BS = r'''
import sys
sys.path.insert(0, {tests!r}); sys.path.insert(0, {repo!r})
import numpy as np, torch
from hip_util import bs_batch, demo_cfg, hip_model
b, meta = bs_batch(100, seed=11)
torch.manual_seed(0)
m = hip_model(demo_cfg(dropout=0.1)).train()
m._step_counter = 3
_, loss = m.loss_and_grad(b['times'], b['time_ptr'], b['X'].cuda(), b['obs_idx'].cuda().int(), meta['dt'],
                          meta['maturity'], b['start_X'].cuda(), b['n_obs_ot'].cuda().int())
np.save({out!r}, np.concatenate([[float(loss)], m.flat_grad().cpu().numpy().astype(np.float64)]))
'''


def test_segment_plan_keep_bits_ahead_are_the_bits_of_the_stream(tmp_path):
    """small plans of the segment plan (every tile four waves wide): the keep bits drawn by the
    spare blocks of the pack launch (drop_bits_tile_steps) against split_keep_bits inside the
    kernel: bit-identical loss and gradient at B = 100."""
    res = {}
    for tag, val in (('ahead', '1'), ('inside', '0')):
        out = str(tmp_path / (tag + '_bs.npy'))
        # (NJODE_SEG_CHAIN_MAX=0: the four-wave tiles this test is about, not the wave-per-item
        # kernels a batch of this size runs by default since round 6)
        env = dict(os.environ, NJODE_DROP_BITS_AHEAD=val, NJODE_SEG_CHAIN_MAX='0')
        p = subprocess.run([sys.executable, '-c', _SNIPPET_BS.format(tests=TESTS, repo=REPO, out=out)],
                           env=env, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                           timeout=600)
        assert p.returncode == 0, p.stdout[-3000:]
        res[tag] = np.load(out)
    assert np.isfinite(res['ahead']).all() and abs(res['ahead'][0]) > 0
    assert np.array_equal(res['ahead'], res['inside'])


_SNIPPET_SEG = r'''
import sys
sys.path.insert(0, {tests!r}); sys.path.insert(0, {repo!r})
import numpy as np, torch
from hip_util import bs_batch, demo_cfg, hip_model
b, meta = bs_batch({B}, seed=11)
torch.manual_seed(0)
m = hip_model(demo_cfg(dropout=0.1)).train()
m._step_counter = 3
args = (b['times'], b['time_ptr'], b['X'].cuda(), b['obs_idx'].cuda().int(), meta['dt'],
        meta['maturity'], b['start_X'].cuda(), b['n_obs_ot'].cuda().int())
_, loss = m.loss_and_grad(*args)
g1 = m.flat_grad().cpu().numpy().astype(np.float64).copy()
# the reference's call sequence: hT comes back too (the tails ride in the forward's launch)
m._step_counter = 3
m.zero_grad()
hT, loss2 = m(*args)
loss2.backward()
g2 = np.concatenate([p.grad.detach().cpu().numpy().reshape(-1) for p in m.parameters()]).astype(np.float64)
np.save({out!r}, np.concatenate([[float(loss), float(loss2)], hT.detach().cpu().numpy().reshape(-1).astype(np.float64), g1, g2]))
'''


@pytest.mark.parametrize('B', [100, 200, 7, 1200])   # (1 200: more than 768 tiles of rows -- the lane masks' buffer)
def test_wave_per_item_ode_kernels_match_the_tiles(tmp_path, B):
    """round 6, njode_chain_seg.h: the segment plan's ODE kernels with one wave per item (the default up
    to 16 384 items + paths) against the 16-chain tiles over four waves, dropout ON -- the same keep
    masks, so loss, hT and gradient agree to fp32 summation order on the fused step and on the
    reference's call sequence (model(...); loss.backward()).  The ODE weight gradients of the new
    route come from the lockstep plan's (step, path) pair kernel on the stored adjoints."""
    res = {}
    for tag, env_extra in (('items', {}), ('tiles', {'NJODE_SEG_CHAIN_MAX': '0'})):
        out = str(tmp_path / (tag + '_seg.npy'))
        p = subprocess.run([sys.executable, '-c', _SNIPPET_SEG.format(tests=TESTS, repo=REPO, out=out, B=B)],
                           env=dict(os.environ, **env_extra), cwd=REPO, stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True, timeout=600)
        assert p.returncode == 0, p.stdout[-3000:]
        res[tag] = np.load(out)
    a, t = res['items'], res['tiles']
    assert np.isfinite(a).all() and abs(a[0]) > 0
    n_h = B * 10
    n_p = (a.size - 2 - n_h) // 2
    assert a[0] == pytest.approx(t[0], rel=2e-5) and a[1] == pytest.approx(t[1], rel=2e-5)
    np.testing.assert_allclose(a[2:2 + n_h], t[2:2 + n_h], atol=2e-5, rtol=1e-4)
    assert rel_l2(a[2 + n_h:2 + n_h + n_p], t[2 + n_h:2 + n_h + n_p]) < 1e-4
    assert rel_l2(a[2 + n_h + n_p:], t[2 + n_h + n_p:]) < 1e-4
    # both routes of the new kernels: the same step
    assert a[0] == pytest.approx(a[1], rel=1e-6)
    assert rel_l2(a[2 + n_h:2 + n_h + n_p], a[2 + n_h + n_p:]) < 1e-5


def test_weight_gradients_from_stored_operands_match_the_recomputing_kernel(tmp_path):
    """round 6, njode_chain_dw.h: behind the wave-per-chain sweeps the ODE network's weight gradients are
    outer products of operands the sweeps stored (delta1 / delta2 per pair, sums of delta1 per segment for
    the x / tau / time columns of W1).  NJODE_CHAIN_DELTA=0 keeps k_ode_dw_pairs_mfma, which recomputes the
    deltas from the adjoints and forms every column per pair -- the route the library also takes when the
    delta records do not fit the record budget.  Same forward (loss, hT bit for bit), same gradient to
    fp32 summation order; masked lockstep plan and the demo models' segment plan."""
    n_h = 37 * 41
    new = _run(tmp_path, 'stored', 50, {})
    old = _run(tmp_path, 'recomputed', 50, {'NJODE_CHAIN_DELTA': '0'})
    assert np.array_equal(new[:1 + n_h], old[:1 + n_h])
    print('masked: rel-L2 of the gradients', rel_l2(new[1 + n_h:], old[1 + n_h:]))
    assert rel_l2(new[1 + n_h:], old[1 + n_h:]) < 1e-6          # (measured: 9e-9)
    assert np.abs(new[1 + n_h:] - old[1 + n_h:]).max() <= 1e-6 * np.abs(old[1 + n_h:]).max()
    res = {}
    for tag, env_extra in (('stored', {}), ('recomputed', {'NJODE_CHAIN_DELTA': '0'})):
        out = str(tmp_path / (tag + '_seg.npy'))
        p = subprocess.run([sys.executable, '-c', _SNIPPET_SEG.format(tests=TESTS, repo=REPO, out=out, B=100)],
                           env=dict(os.environ, **env_extra), cwd=REPO, stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True, timeout=600)
        assert p.returncode == 0, p.stdout[-3000:]
        res[tag] = np.load(out)
    a, t = res['stored'], res['recomputed']
    n_h = 100 * 10
    assert np.array_equal(a[:2 + n_h], t[:2 + n_h])
    print('demo: rel-L2 of the gradients', rel_l2(a[2 + n_h:], t[2 + n_h:]))
    assert rel_l2(a[2 + n_h:], t[2 + n_h:]) < 1e-6              # (measured: 3e-9)


@pytest.mark.parametrize('shape', ['demo_b100', 'masked_b37', 'demo_b1600'])
def test_a_step_reads_nothing_it_did_not_write(shape):
    """The workspace is the caller's and arrives uninitialised.  The wave-per-item route hands (step, path)
    pairs to the lockstep plan's weight-gradient kernel, which reads the stored activations of EVERY pair
    -- also of the pairs behind a path's last observation, which no item owns: their records must have
    been filled (0 x NaN is NaN; found by a flaky resume test in round 6).  Every pooled workspace is
    poisoned with NaN bit patterns between two identical steps: same loss, same gradient, bit for bit."""
    import torch
    from hip_util import bs_batch, demo_cfg, hip_model
    from njode_amd import models, synthetic_physionet
    if shape in ('demo_b100', 'demo_b1600'):
        # (1 600 paths: the mixed matrix-core kernels -- the item records the saving forward packs for the
        # backward's tile prologue are the forward's to write, every step)
        b, meta = bs_batch(100 if shape == 'demo_b100' else 1600, seed=11,
                           obs_perc=0.05 if shape == 'demo_b100' else 0.1)   # (0.05: long tails behind the last observations)
        m = hip_model(demo_cfg(dropout=0.1)).train()
        args = (b['times'], b['time_ptr'], b['X'].cuda(), b['obs_idx'].cuda().int(), meta['dt'], meta['maturity'],
                b['start_X'].cuda(), b['n_obs_ot'].cuda().int())
        kw = {}
    else:
        NN = ((50, 'tanh'), (50, 'tanh'))
        cfg = dict(input_size=41, hidden_size=41, output_size=41, ode_nn=NN, readout_nn=NN, enc_nn=NN,
                   use_rnn=False, bias=True, dropout_rate=0.1, options={'masked': True, 'device_outputs': True})
        b = synthetic_physionet.make_batch(batch_size=37, n_grid=60, n_obs_range=(3, 9), seed=3)
        torch.manual_seed(0)
        m = models.NJODE(**cfg).cuda().train()
        args = (b['times'], b['time_ptr'], b['X'].cuda(), b['obs_idx'].cuda().int(), b['delta_t'], b['T'],
                b['start_X'].cuda(), b['n_obs_ot'].cuda().int())
        kw = {'M': b['M'].cuda()}
    res = []
    for rep in range(3):
        m._step_counter = 3
        _, loss = m.loss_and_grad(*args, **kw)
        res.append((float(loss), m.flat_grad().clone()))
        torch.cuda.synchronize()
        for slot in m._ws_pool:                     # poison what the next step will be handed
            if not slot[1]:
                slot[0].fill_(0xFF)
        torch.cuda.synchronize()
    assert np.isfinite(res[0][0]) and bool(torch.isfinite(res[0][1]).all())
    for r in res[1:]:
        assert r[0] == res[0][0] and torch.equal(r[1], res[0][1])
