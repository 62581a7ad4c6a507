"""njode_plan_f32 / NJODE.prefetch_plan: a plan built ahead on a helper stream is the plan the
call would have built itself -- loss and gradient are bit-identical, for the segment plan
(demo shape, counting-sort and Onesweep sizes) and the masked lockstep plan."""
import pytest
import torch

from hip_util import bs_batch, demo_cfg, to_dev
from njode_amd import models, synthetic_physionet

pytestmark = pytest.mark.gpu
NN = ((50, 'tanh'), (50, 'tanh'))


def _step(m, args, kw, prefetch, left=0):
    m._step_counter = 3
    if prefetch:
        m.prefetch_plan(*args, **kw)
        assert m._plans
    _, loss = m.loss_and_grad(*args, **kw)
    assert len(m._plans) == left   # one prefetched plan was picked up
    return float(loss), m.flat_grad().clone()


WIDE = ((100, 'tanh'), (100, 'tanh'))      # the shape-generic kernels (a plan of their own: items sorted by rocPRIM)


@pytest.mark.parametrize('nets', [None, WIDE], ids=['demo', 'generic_w100'])
@pytest.mark.parametrize('n_paths', [300, 3000])
def test_prefetched_plan_gives_identical_results_on_the_segment_plan(n_paths, nets):
    cfg = demo_cfg(dropout=0.1, device_outputs=True)
    if nets is not None:
        cfg.update(ode_nn=nets, enc_nn=nets, readout_nn=nets)
    torch.manual_seed(0)
    m = models.NJODE(**cfg).cuda().train()
    b, meta = bs_batch(n_paths, seed=4)
    b = to_dev(b)
    obs_idx = b['obs_idx'].cuda().int()
    args = (b['times'], b['time_ptr'], b['X'], obs_idx, meta['dt'], meta['maturity'], b['start_X'],
            b['n_obs_ot'])
    l0, g0 = _step(m, args, {}, False)
    l1, g1 = _step(m, args, {}, True)
    assert l1 == l0 and torch.equal(g1, g0)
    # two plans in flight, consumed in order
    m.prefetch_plan(*args)
    l2, g2 = _step(m, args, {}, True, left=1)
    l3, g3 = _step(m, args, {}, False)
    assert (l2, l3) == (l0, l0) and torch.equal(g2, g0) and torch.equal(g3, g0)


def test_prefetched_plan_gives_identical_results_on_the_masked_plan():
    cfg = dict(input_size=41, hidden_size=41, output_size=41, ode_nn=NN, readout_nn=NN, enc_nn=NN,
               use_rnn=False, bias=True, dropout_rate=0.0, options={'masked': True, 'device_outputs': True})
    b = synthetic_physionet.make_batch(batch_size=20, n_grid=120, n_obs_range=(3, 9), seed=3)
    torch.manual_seed(0)
    m = models.NJODE(**cfg).cuda().train()
    obs_idx = b['obs_idx'].cuda().int()
    args = (b['times'], b['time_ptr'], b['X'].cuda(), obs_idx, b['delta_t'], b['T'],
            b['start_X'].cuda(), b['n_obs_ot'].cuda().int())
    kw = {'M': b['M'].cuda()}
    l0, g0 = _step(m, args, kw, False)
    l1, g1 = _step(m, args, kw, True)
    assert l1 == l0 and torch.equal(g1, g0)


def test_autograd_route_picks_a_prefetched_plan_up():
    cfg = demo_cfg(dropout=0.0, device_outputs=True)
    torch.manual_seed(0)
    m = models.NJODE(**cfg).cuda().train()
    b, meta = bs_batch(200, seed=5)
    b = to_dev(b)
    obs_idx = b['obs_idx'].cuda().int()
    args = (b['times'], b['time_ptr'], b['X'], obs_idx, meta['dt'], meta['maturity'], b['start_X'],
            b['n_obs_ot'])

    def run(prefetch):
        for p in m.parameters():
            p.grad = None
        if prefetch:
            m.prefetch_plan(*args)         # default: made for a call that returns hT
        hT, loss = m(*args)
        loss.backward()
        assert not m._plans
        return hT.clone(), float(loss), torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone()

    h0, l0, g0 = run(False)
    h1, l1, g1 = run(True)
    assert l1 == l0 and torch.equal(h1, h0) and torch.equal(g1, g0)


def _demo_args(n_paths, seed):
    b, meta = bs_batch(n_paths, seed=seed)
    b = to_dev(b)
    obs_idx = b['obs_idx'].cuda().int()
    return (b['times'], b['time_ptr'], b['X'], obs_idx, meta['dt'], meta['maturity'], b['start_X'],
            b['n_obs_ot'])


def test_a_plan_is_only_taken_by_the_very_batch_it_was_made_for():
    """plans were keyed by id(obs_idx), id(time_ptr) without keeping the objects
    alive.  Now a plan holds its batch's objects, is matched by identity, and a plan that does
    not fit the call (no tail order although hT is wanted; obs_idx modified in place) is dropped
    instead of used or left to block the queue."""
    cfg = demo_cfg(dropout=0.0, device_outputs=True)
    torch.manual_seed(0)
    m = models.NJODE(**cfg).cuda().train()
    a = _demo_args(300, seed=4)
    b = _demo_args(300, seed=5)          # same sizes possible, different rows
    l_a = float(m.loss_and_grad(*a)[1])
    l_b = float(m.loss_and_grad(*b)[1])
    # 1. a prefetched plan of batch a is NOT picked up by batch b
    m.prefetch_plan(*a, need_hT=False)
    assert float(m.loss_and_grad(*b)[1]) == l_b and len(m._plans) == 1
    assert float(m.loss_and_grad(*a)[1]) == l_a and not m._plans
    # 2. a handle is accepted by the call it is passed to, exactly once
    h = m.prefetch_plan(*a, need_hT=False)
    assert float(m.loss_and_grad(*a, plan=h)[1]) == l_a and not m._plans
    with pytest.raises(RuntimeError):
        m.loss_and_grad(*a, plan=h)
    # 3. made without the tail order, then model(...) wants hT: dropped, the call plans in line
    m.prefetch_plan(*a, need_hT=False)
    with torch.no_grad():
        hT, loss = m(*a)
    assert float(loss) == pytest.approx(l_a, rel=1e-6) and not m._plans
    assert torch.isfinite(hT).all()
    # 4. obs_idx modified in place after the prefetch: dropped
    m.prefetch_plan(*a, need_hT=False)
    a[3].add_(0)
    assert float(m.loss_and_grad(*a)[1]) == l_a and not m._plans
    # 5. an unconsumed plan does not block a later one of another batch
    m.prefetch_plan(*a, need_hT=False)
    m.prefetch_plan(*b, need_hT=False)
    assert float(m.loss_and_grad(*b)[1]) == l_b and len(m._plans) == 1


def test_deferred_plan_flush_hosting_by_another_model_and_lifetime():
    """NJODE_C_PLAN_DEFER (round 5): the plan of a prefetched batch is launched by the NEXT forward call
    on the stream -- whoever makes it -- or by njode_plan_flush / the plan's destructor; in every case
    the consuming step gives the bits of a step that planned in line."""
    import gc
    import os
    from njode_amd import _lib
    if (os.environ.get('NJODE_PLAN_DEFER', '1') == '0' or os.environ.get('NJODE_PLAN_GRID', '1') == '0'
            or os.environ.get('NJODE_VALIDATE', '0') not in ('', '0')):
        # (NJODE_VALIDATE=1: the library checks the batch while it plans, so it plans in line -- fill_plan_job)
        pytest.skip('the deferred plan is switched off in this environment')
    cfg = demo_cfg(dropout=0.1, device_outputs=True)
    torch.manual_seed(0)
    m = models.NJODE(**cfg).cuda().train()
    torch.manual_seed(1)
    other = models.NJODE(**cfg).cuda().train()
    a = _demo_args(400, seed=4)
    b = _demo_args(900, seed=5)
    assert m.plan_defer_ok(int(a[1][-1]))
    l0, g0 = _step(m, a, {}, False)
    L = _lib.lib()
    assert L.njode_plan_flush() == 0
    # 1. flushed by hand
    m.prefetch_plan(*a, need_hT=False)
    assert m._plans[0].done is None and m._plans[0].pending
    assert L.njode_plan_flush() == 1 and L.njode_plan_flush() == 0
    l1, g1 = _step(m, a, {}, False, left=0)
    assert l1 == l0 and torch.equal(g1, g0)
    # 2. hosted by another model's step on another batch
    m.prefetch_plan(*a, need_hT=False)
    other.loss_and_grad(*b)
    assert L.njode_plan_flush() == 0          # (it rode in that call's ODE-forward launch)
    l2, g2 = _step(m, a, {}, False, left=0)
    assert l2 == l0 and torch.equal(g2, g0)
    # 3. the consuming call itself finds it pending: launched in front of it
    l3, g3 = _step(m, a, {}, True)
    assert l3 == l0 and torch.equal(g3, g0)
    # 3b. more prefetches in a row than the pinned ring has slots: every older job is launched by the next
    # njode_plan_f32 and its slot handed back; the newest four plans stay queued
    for _ in range(40):
        m.prefetch_plan(*a, need_hT=False)
    assert len(m._plans) == 4 and len(m._deferred_slots) == 1
    l4, g4 = _step(m, a, {}, False, left=3)
    assert l4 == l0 and torch.equal(g4, g0)
    m._plans.clear()
    # 4. a plan nobody launches dies with its model: its destructor launches it before the buffer goes
    m.prefetch_plan(*a, need_hT=False)
    del m
    gc.collect()
    assert L.njode_plan_flush() == 0
    torch.cuda.synchronize()
    assert float(other.loss_and_grad(*b)[1]) > 0


def test_deferred_plan_consumed_on_another_stream_is_ordered_behind_its_own():
    """a deferred plan is ordered by the stream prefetch_plan saw.  A step that runs
    on ANOTHER stream (a torch.cuda.stream context entered only around the step) must still see the
    finished plan: the consuming call launches the pending job on the plan's stream and waits for that
    stream.  The plan's stream is kept busy (a long elementwise kernel queued in front of the prefetch)
    so that an unordered read would be a read of a plan that does not exist yet."""
    import os
    if os.environ.get('NJODE_PLAN_DEFER', '1') == '0' or os.environ.get('NJODE_PLAN_GRID', '1') == '0':
        pytest.skip('the deferred plan is switched off in this environment')
    cfg = demo_cfg(dropout=0.1, device_outputs=True)
    torch.manual_seed(0)
    m = models.NJODE(**cfg).cuda().train()
    a = _demo_args(900, seed=4)
    l0, g0 = _step(m, a, {}, False)
    side = torch.cuda.Stream()
    busy = torch.ones(64 << 20, device='cuda')
    for rep in range(3):
        for _ in range(20):
            busy.mul_(1.0000001)                      # ~ms of work in front of the plan on the default stream
        m.prefetch_plan(*a, need_hT=False)            # described on the default stream, deferred
        assert m._plans[-1].done is None and m._plans[-1].stream is not None
        side.wait_stream(torch.cuda.current_stream())   # (the batch's arrays; NOT the plan: it is still pending)
        with torch.cuda.stream(side):
            m._step_counter = 3
            _, loss = m.loss_and_grad(*a)
            g = m.flat_grad().clone()
        torch.cuda.current_stream().wait_stream(side)
        assert float(loss) == l0 and torch.equal(g, g0), rep
    torch.cuda.synchronize()


_SNIPPET_STRESS = r'''
import sys, threading
sys.path.insert(0, {tests!r}); sys.path.insert(0, {repo!r})
import numpy as np, torch
from hip_util import bs_batch, demo_cfg, to_dev
from njode_amd import models, _lib
def args_of(n, seed):
    b, meta = bs_batch(n, seed=seed)
    b = to_dev(b)
    return (b['times'], b['time_ptr'], b['X'], b['obs_idx'].cuda().int(), meta['dt'], meta['maturity'], b['start_X'], b['n_obs_ot'])
cfg = demo_cfg(dropout=0.1, device_outputs=True)
batches = [args_of(600 + 50 * i, seed=20 + i) for i in range(4)]
torch.manual_seed(0)
ref_model = models.NJODE(**cfg).cuda().train()
ref = []
for a in batches:                                   # the reference: every plan in line, one thread
    ref_model._step_counter = 5
    _, loss = ref_model.loss_and_grad(*a)
    ref.append((float(loss), ref_model.flat_grad().clone()))
torch.cuda.synchronize()
errors = []
workers = []
for _ in range(2):                                  # (built here: torch's generator is process-global)
    torch.manual_seed(0)
    workers.append(models.NJODE(**cfg).cuda().train())
torch.cuda.synchronize()
def worker(tid):
    try:
        m = workers[tid]
        streams = [torch.cuda.Stream() for _ in range(2)]
        for rep in range(8):
            s = streams[rep % 2]
            with torch.cuda.stream(s):
                # more multi-block plans in flight than round 5's sixteen counter sets
                for a in batches:
                    m.prefetch_plan(*a, need_hT=False, defer=False)   # helper stream: P > 1 blocks each
                for i, a in enumerate(batches):
                    m._step_counter = 5
                    _, loss = m.loss_and_grad(*a)
                    g = m.flat_grad().clone()
                    s.synchronize()
                    if float(loss) != ref[i][0] or not torch.equal(g, ref[i][1]):
                        errors.append((tid, rep, i, float(loss), ref[i][0]))
    except Exception as e:
        errors.append((tid, repr(e)))
ts = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
for t in ts: t.start()
for t in ts: t.join()
torch.cuda.synchronize()
print('FAILURES', _lib.lib().njode_plan_barrier_failures(), 'ERRORS', errors[:3])
assert not errors, errors[:3]
assert _lib.lib().njode_plan_barrier_failures() == 0
'''


def test_many_multi_block_plans_in_flight_over_streams_and_threads(tmp_path):
    """VERDICT r5 item 6 / ADVICE r5: the grid barrier of the one-launch plan now counts in eight words
    of the PLAN'S OWN buffer (zeroed on the launch's stream) instead of sixteen shared sets handed out
    round-robin, and its spin is bounded.  Two host threads, two streams each, 64 four-block plans in
    flight (NJODE_PLAN_BLOCKS=4, built on the helper stream): every step must give the bits of the step
    that planned in line, and no barrier may have timed out."""
    import os
    import subprocess
    import sys
    tests = os.path.dirname(os.path.abspath(__file__))
    repo = os.path.dirname(tests)
    p = subprocess.run([sys.executable, '-c', _SNIPPET_STRESS.format(tests=tests, repo=repo)],
                       env=dict(os.environ, NJODE_PLAN_BLOCKS='4'), cwd=repo, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:]
    assert 'FAILURES 0' in p.stdout
