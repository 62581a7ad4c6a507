"""
NJ-ODE model with the operator surface of the reference's ``NJODE/models.py``,
executed by hand-written HIP kernels for gfx950 (libnjode_hip.so).

Same module-level names, constructor, ``forward`` signature and return
conventions, ``evaluate`` / ``get_pred`` / ``weight_decay_step``, checkpoint
helpers and ``state_dict`` keys as the reference (file:line cited per symbol),
so the class is constructed and called exactly like ``models.NJODE`` is by
``train.py`` / ``demo.py``.

What is different by construction:

* the sub-modules (``ode_f``, ``encoder_map``, ``readout_map``) only *hold*
  parameters under the reference's names; their math runs fused inside the
  kernels, so calling them directly raises.  There is no eager / CPU fallback:
  a model on a CPU device, or an unbuilt library, raises.
* all parameters are views of one flat fp32 vector (the layout of the C ABI),
  so the gradient comes back as one vector and the optimizer step can be fused.
* ``options['device_outputs']=True`` keeps ``loss`` / predictions on the GPU
  (no host sync); the default follows the reference harness, which calls
  ``.numpy()`` on them (``train.py:558,571,726``), and returns CPU tensors.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib
from .schedule import PinnedRing, ScheduleCache


# =====================================================================================
# module-level helpers with the reference's names
# =====================================================================================
def init_weights(m, bias=0.0):
    """Xavier-uniform weights, constant bias (reference ``models.py:21-26``)."""
    if type(m) == torch.nn.Linear:
        torch.nn.init.xavier_uniform_(m.weight)
        if m.bias is not None:
            m.bias.data.fill_(bias)


def save_checkpoint(model, optimizer, path, epoch):
    """``checkpt.tar`` with the reference's keys (``models.py:29-45``)."""
    if not os.path.exists(path):
        os.makedirs(path)
    torch.save({'epoch': epoch, 'weight': model.weight,
                'model_state_dict': model.state_dict(),
                'optimizer_state_dict': optimizer.state_dict()},
               os.path.join(path, 'checkpt.tar'))


def get_ckpt_model(ckpt_path, model, optimizer, device):
    """Load a checkpoint written by this build or by the reference, in place
    (``models.py:48-66``)."""
    ckpt_path = os.path.join(ckpt_path, 'checkpt.tar')
    if not os.path.exists(ckpt_path):
        raise Exception("Checkpoint " + ckpt_path + " does not exist.")
    checkpt = torch.load(ckpt_path, map_location='cpu', weights_only=False)
    if optimizer is not None:
        optimizer.load_state_dict(checkpt['optimizer_state_dict'])
    model.load_state_dict(checkpt['model_state_dict'])
    model.epoch = checkpt['epoch']
    model.weight = checkpt['weight']
    model.to(device)


def _norm(t, dim=1, eps=1e-10):
    return torch.sqrt(torch.sum(t, dim=dim) + eps)


def compute_loss(X_obs, Y_obs, Y_obs_bj, n_obs_ot, batch_size, eps=1e-10,
                 weight=0.5, M_obs=None):
    """Paper loss on tensors (reference ``models.py:71-106``).  Utility for
    callers that hold predictions; the training path computes the same
    expression inside the kernels."""
    m = 1.0 if M_obs is None else M_obs
    after = _norm(m * (X_obs - Y_obs) ** 2, eps=eps)
    before = _norm(m * (Y_obs_bj - Y_obs) ** 2, eps=eps)
    inner = (2 * weight * after + 2 * (1 - weight) * before) ** 2
    return torch.sum(inner / n_obs_ot) / batch_size


def compute_loss_2(X_obs, Y_obs, Y_obs_bj, n_obs_ot, batch_size, eps=1e-10,
                   weight=0.5, M_obs=None):
    """'easy' variant: second term uses X instead of Y (``models.py:109-126``)."""
    m = 1.0 if M_obs is None else M_obs
    after = _norm(m * (X_obs - Y_obs) ** 2, eps=eps)
    before = _norm(m * (Y_obs_bj - X_obs) ** 2, eps=eps)
    inner = (weight * after + (1 - weight) * before) ** 2
    return torch.sum(inner / n_obs_ot) / batch_size


LOSS_FUN_DICT = {'standard': compute_loss, 'easy': compute_loss_2}

nonlinears = {'tanh': torch.nn.Tanh, 'relu': torch.nn.ReLU}


def get_ffnn(input_size, output_size, nn_desc, dropout_rate, bias):
    """Parameter container with the reference's Sequential layout
    (``models.py:140-166``): Linear at indices 0, 3, 6, ... with activation and
    Dropout modules in between (kept so ``print(model)`` and the state_dict
    keys match)."""
    if nn_desc is None:
        layers = [torch.nn.Linear(input_size, output_size, bias=bias)]
    else:
        layers = [torch.nn.Linear(input_size, nn_desc[0][0], bias=bias)]
        for i in range(len(nn_desc) - 1):
            layers.append(nonlinears[nn_desc[i][1]]())
            layers.append(torch.nn.Dropout(p=dropout_rate))
            layers.append(torch.nn.Linear(nn_desc[i][0], nn_desc[i + 1][0], bias=bias))
        layers.append(nonlinears[nn_desc[-1][1]]())
        layers.append(torch.nn.Dropout(p=dropout_rate))
        layers.append(torch.nn.Linear(nn_desc[-1][0], output_size, bias=bias))
    return torch.nn.Sequential(*layers)


_FUSED_MSG = ('{} holds parameters only: its math runs inside the fused HIP kernels of '
              'NJODE.forward (libnjode_hip.so); there is no eager path.')


class ODEFunc(torch.nn.Module):
    """f_theta: parameters ``ode_f.f.*`` (reference ``models.py:170-199``)."""

    def __init__(self, input_size, hidden_size, ode_nn, dropout_rate=0.0, bias=True,
                 input_current_t=False):
        super().__init__()
        self.input_current_t = input_current_t
        add = 3 if input_current_t else 2
        self.f = get_ffnn(input_size=input_size + hidden_size + add, output_size=hidden_size,
                          nn_desc=ode_nn, dropout_rate=dropout_rate, bias=bias)

    def forward(self, x, h, tau, tdiff):
        raise RuntimeError(_FUSED_MSG.format('ODEFunc'))


class GRUCell(torch.nn.Module):
    """rho_theta: parameters ``obs_c.gru_d.*`` (reference ``models.py:202-217``)."""

    def __init__(self, input_size, hidden_size, bias=True):
        super().__init__()
        self.gru_d = torch.nn.GRUCell(input_size, hidden_size, bias=bias)
        self.input_size = input_size

    def forward(self, h, X_obs, i_obs):
        raise RuntimeError(_FUSED_MSG.format('GRUCell'))


class FFNN(torch.nn.Module):
    """Encoder / readout: parameters ``*.ffnn.*``; residual size rules and error
    messages of the reference (``models.py:220-276``)."""

    def __init__(self, input_size, output_size, nn_desc, dropout_rate=0.0, bias=True,
                 residual=False, masked=False):
        super().__init__()
        in_size = 2 * input_size if masked else input_size
        self.masked = masked
        self.ffnn = get_ffnn(input_size=in_size, output_size=output_size, nn_desc=nn_desc,
                             dropout_rate=dropout_rate, bias=bias)
        if residual:
            print('use residual network: input_size={}, output_size={}'.format(
                input_size, output_size))
            if input_size <= output_size:
                if output_size % input_size == 0:
                    self.case = 1
                    self.mult = int(output_size / input_size)
                else:
                    raise ValueError('for residual: output_size needs to be '
                                     'multiple of input_size')
            if input_size > output_size:
                if input_size % output_size == 0:
                    self.case = 2
                    self.mult = int(input_size / output_size)
                else:
                    raise ValueError('for residual: input_size needs to be '
                                     'multiple of output_size')
        else:
            self.case = 0

    def forward(self, nn_input, mask=None):
        raise RuntimeError(_FUSED_MSG.format('FFNN'))


# =====================================================================================
# autograd bridge
# =====================================================================================
class _Call:
    """Everything one library call needs, kept alive until the stream has used it.
    Dropping the last reference (e.g. an autograd graph that is never back-propagated)
    returns the workspace to the model's pool."""
    __slots__ = ('dims', 'batch', 'sched', 'flags', 'weight', 'p_drop', 'seed', 'ws',
                 'keep', 'ws_slot', 'sched_obj', 'time_ptr')

    def __del__(self):
        slot = getattr(self, 'ws_slot', None)
        if slot is not None:
            slot[1] = False


class _Plan:
    """A plan built ahead by ``NJODE.prefetch_plan``; its buffer returns to the model's pool
    when the last call that reads it is gone.  It holds the ORIGINAL ``obs_idx`` / ``time_ptr``
    objects of its batch (so their ids cannot be recycled for another batch while the plan is
    queued) and the version counter ``obs_idx`` had: a call only takes a plan of the very same,
    unmodified objects."""
    __slots__ = ('buf', 'done', 'flags', 'sizes', 'keep', 'pool', 'obs_idx', 'time_ptr',
                 'obs_version', 'taken', 'pending', 'stream')

    def __del__(self):
        # a deferred plan nobody has launched yet must not outlive its buffer
        if getattr(self, 'pending', False):
            try:
                _lib.lib().njode_plan_flush()
            except Exception:
                pass
        pool, buf = getattr(self, 'pool', None), getattr(self, 'buf', None)
        if pool is not None and buf is not None and len(pool) < 4:
            pool.append(buf)


def _hT_and_loss_grads(model, call, grad_loss, grad_hT):
    """Flat parameter gradient of  grad_loss * loss + <grad_hT, hT>  for a saved forward ``call``.
    The loss term is the call's own backward (whatever plan it ran).  The hT term -- the reference
    returns hT inside its autograd graph, models.py:414-518 -- is a second pass: the same step once
    more on the LOCKSTEP plan (whose adjoint sweep can start from an upstream gradient of the final
    state: NjodeBatch.grad_hT) with the loss switched off (loss_batch_size = inf), same dropout
    seed, i.e. the same masks.  It costs a lockstep forward + backward and is only paid by a
    backward pass that really reaches hT (``train.py`` never does)."""
    grad_flat = torch.empty_like(model._flat)
    dev = model._flat.device
    if grad_loss is None:
        grad_loss = torch.zeros(1, device=dev)
    g = grad_loss.to(device=dev, dtype=torch.float32).reshape(1).contiguous()
    model._run_backward(call, g, grad_flat)
    if grad_hT is not None:
        grad_flat += model._grad_through_hT(call, grad_hT)
    model._release_ws(call)
    return grad_flat


class _NJODEFunction(torch.autograd.Function):
    """(loss, hT) = F(params); backward = exact discrete adjoint (njode_backward_f32); a gradient
    that reaches hT adds the pass of ``_hT_and_loss_grads``."""

    @staticmethod
    def forward(ctx, model, call, loss, hT, *params):
        ctx.model = model
        ctx.call = call
        ctx.set_materialize_grads(False)
        # (views, not copies: `loss` and `hT` are this call's own fresh tensors; a clone was one more
        # 5 us launch on the chain between the forward's last kernel and the backward's first)
        return loss.view_as(loss), hT.view_as(hT)

    @staticmethod
    def backward(ctx, grad_loss, grad_hT):
        model, call = ctx.model, ctx.call
        grad_flat = _hT_and_loss_grads(model, call, grad_loss, grad_hT)
        grads = [grad_flat[off:off + n].view(shape) for (off, n, shape) in model._param_slices]
        return (None, None, None, None) + tuple(grads)


class _NJODEhTOnlyFunction(torch.autograd.Function):
    """hT = F(params) of a ``get_loss=False`` call: the reference's hT is differentiable there too
    (models.py:414-518).  Nothing was saved by the forward; a backward that reaches hT replays the
    step on the lockstep plan (``NJODE._grad_through_hT``)."""

    @staticmethod
    def forward(ctx, model, call, hT, *params):
        ctx.model = model
        ctx.call = call
        ctx.set_materialize_grads(False)
        return hT.view_as(hT)

    @staticmethod
    def backward(ctx, grad_hT):
        model = ctx.model
        if grad_hT is None:
            grad_flat = torch.zeros_like(model._flat)
        else:
            grad_flat = model._grad_through_hT(ctx.call, grad_hT)
        grads = [grad_flat[off:off + n].view(shape) for (off, n, shape) in model._param_slices]
        return (None, None, None) + tuple(grads)


# =====================================================================================
# the model
# =====================================================================================
def _desc_of(nn_desc):
    """(n_hidden, widths, acts) of a get_ffnn description (``nn_desc=None``: one Linear)."""
    if nn_desc is None:
        return (0, (), ())
    widths = tuple(int(w) for w, _ in nn_desc)
    for _, a in nn_desc:
        if a not in nonlinears:
            raise KeyError(a)
    acts = tuple(_lib.ACT_TANH if a == 'tanh' else _lib.ACT_RELU for _, a in nn_desc)
    return (len(nn_desc), widths, acts)


def _raw_current_stream(dev):
    """hipStream_t of torch's current stream on ``dev`` as an int, without building a Stream object."""
    get = getattr(torch._C, '_cuda_getCurrentRawStream', None)
    if get is not None and dev.index is not None:
        return get(dev.index)
    return torch.cuda.current_stream(dev).cuda_stream


class NJODE(torch.nn.Module):
    """NJ-ODE model (reference ``models.py:280-584``), HIP-backed."""

    def __init__(self, input_size, hidden_size, output_size, ode_nn, readout_nn, enc_nn,
                 use_rnn, bias=True, dropout_rate=0, solver="euler", weight=0.5,
                 weight_decay=1., **options):
        super().__init__()
        self.epoch = 1
        self.weight = weight
        self.weight_decay = weight_decay
        self.use_rnn = use_rnn

        # the harness passes its whole params_dict; the model's own switches sit
        # under options['options'] (reference models.py:321)
        options1 = options['options']
        self.which_loss = options1.get('which_loss', 'standard')
        assert self.which_loss in LOSS_FUN_DICT
        print('using loss: {}'.format(self.which_loss))
        self.residual_enc_dec = options1.get('residual_enc_dec', True)
        self.input_current_t = options1.get('input_current_t', False)
        self.masked = options1.get('masked', False)
        self.device_outputs = bool(options1.get('device_outputs', False))
        # route the forward through torch.ops.njode_amd.forward (njode_amd/ops.py)
        self.torch_library_op = bool(options1.get('torch_library_op', False))

        self.ode_f = ODEFunc(input_size, hidden_size, ode_nn, dropout_rate, bias,
                             input_current_t=self.input_current_t)
        self.encoder_map = FFNN(input_size, hidden_size, enc_nn, dropout_rate, bias,
                                masked=self.masked, residual=self.residual_enc_dec)
        self.readout_map = FFNN(hidden_size, output_size, readout_nn, dropout_rate, bias,
                                residual=self.residual_enc_dec)
        if self.use_rnn:
            self.obs_c = GRUCell(input_size, hidden_size, bias=bias)

        self.solver = solver
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.output_size = output_size
        self.dropout_rate = float(dropout_rate)
        self.bias = bias
        self._descs = (_desc_of(ode_nn), _desc_of(enc_nn), _desc_of(readout_nn))

        self.apply(init_weights)

        # data-parallel sharding (set by the harness): global batch size used in the
        # loss denominator and the global index of this shard's first path
        self.dp_global_batch = None
        self.dp_path_offset = 0
        self.seed = int(options1.get('dropout_seed', 0))
        self._step_counter = 0

        self._flat = None
        self._flat_checks = 0
        self._flat_grad = None
        self._grad_bucket = None
        self.dp_loss_in_bucket = False
        self._flat_present = None
        self._param_slices = None
        self._flat_params = None
        self._ones = None
        self._sched_cache = ScheduleCache()
        self._ring = None
        self._ws_pool = []
        self._dims = None
        self._plans = []
        self._last_stream = None
        self._plan_pool = []
        self._plan_stream = None
        self._deferred_slots = []    # pinned schedule slots of plans whose launch is deferred
        self._deferred_plans = []    # ... and the plans themselves (their buffers stay alive until launched)
        self._last_hT_replay = None

    # -- reference API ----------------------------------------------------------------
    def weight_decay_step(self):
        """weight <- 0.5 + (weight - 0.5) * weight_decay (``models.py:364-367``)."""
        inc = (self.weight - 0.5)
        self.weight = 0.5 + inc * self.weight_decay
        return self.weight

    def _apply(self, fn, *args, **kwargs):
        """.to() / .cuda() / .float() move the parameters: the flat vector is rebuilt on next use."""
        out = super()._apply(fn, *args, **kwargs)
        self._flat_checks = 63          # next _ensure_flat walks all parameters
        return out

    # -- flat parameter storage -----------------------------------------------------------
    def _flat_slots(self):
        """[(parameter or None, slot size, shape)] in the C ABI's order: the Linear
        layers of the three networks (bias slot always present), then the GRU cell."""
        slots = []
        for seq in (self.ode_f.f, self.encoder_map.ffnn, self.readout_map.ffnn):
            for m in seq:
                if isinstance(m, torch.nn.Linear):
                    slots.append((m.weight, m.weight.numel(), tuple(m.weight.shape)))
                    slots.append((m.bias, m.out_features, (m.out_features,)))
        if self.use_rnn:
            g = self.obs_c.gru_d
            H3 = 3 * self.hidden_size
            slots.append((g.weight_ih, g.weight_ih.numel(), tuple(g.weight_ih.shape)))
            slots.append((g.weight_hh, g.weight_hh.numel(), tuple(g.weight_hh.shape)))
            slots.append((getattr(g, 'bias_ih', None), H3, (H3,)))
            slots.append((getattr(g, 'bias_hh', None), H3, (H3,)))
        return slots

    def _ensure_flat(self, full=False):
        """Make every parameter a view of one flat vector laid out as the C ABI expects
        (state_dict order; bias slots always present).  ``full``: walk all parameters (once per
        library call, from ``_make_call``: a parameter re-pointed by ``p.data = ...`` or
        ``load_state_dict(assign=True)`` is caught before the kernels read the flat vector)."""
        # fast path (the hot loop calls this several times per step): first and last parameter
        # still sit where they were put; the full walk below runs once per call and after _apply()
        if self._flat is not None and self._flat_params and not full:
            self._flat_checks += 1
            if self._flat_checks & 63:
                base = self._flat.data_ptr()
                (o0, _, _), (o1, _, _) = self._param_slices[0], self._param_slices[-1]
                p0, p1 = self._flat_params[0], self._flat_params[-1]
                if (p0.data_ptr() == base + 4 * o0 and p1.data_ptr() == base + 4 * o1
                        and p0.dtype == torch.float32):
                    return
        if self._flat is not None and self._flat_params:
            base = self._flat.data_ptr()
            ok = True
            for (off, n, _), p in zip(self._param_slices, self._flat_params):
                if p.data_ptr() != base + 4 * off or p.dtype != torch.float32:
                    ok = False
                    break
            if ok:
                return
        slots = self._flat_slots()
        dev = slots[0][0].device
        total = sum(n for _, n, _ in slots)
        flat = torch.zeros(total, dtype=torch.float32, device=dev)
        slices, params, off = [], [], 0
        present = None
        for p, n, shape in slots:
            if p is not None:
                flat[off:off + n].copy_(p.data.reshape(-1).to(torch.float32))
                p.data = flat[off:off + n].view(shape)
                slices.append((off, n, shape))
                params.append(p)
            else:
                # a slot without a parameter (bias=False, models.py:140-166): the kernels read it
                # as zeros and WRITE its gradient; it is not a parameter, so the fused optimizer
                # must leave it alone (FusedAdam.step multiplies the gradient by this mask)
                if present is None:
                    present = torch.ones(total, dtype=torch.float32, device=dev)
                present[off:off + n] = 0.0
            off += n
        self._flat, self._param_slices, self._flat_params = flat, slices, params
        self._flat_present = present
        self._flat_grad = None
        self._grad_bucket = None

    def flat_parameters(self):
        """The flat parameter vector all Linear parameters are views of."""
        self._ensure_flat()
        return self._flat

    def flat_grad(self):
        """A flat gradient vector whose slices are installed as ``.grad`` of the
        parameters (for the fused training step).  It is the head of ``grad_bucket()``."""
        self._ensure_flat()
        if self._flat_grad is None:
            # ONE bucket: the flat gradient and, behind it, one slot for the step's scalar loss,
            # so that a data-parallel step all-reduces both with a single collective
            # ("one RCCL all-reduce (sum) of the flat fp32 gradient [P] + the scalar loss")
            self._grad_bucket = torch.zeros(self._flat.numel() + 1, dtype=torch.float32,
                                            device=self._flat.device)
            self._flat_grad = self._grad_bucket[:-1]
            for (off, n, shape), p in zip(self._param_slices, self._flat_params):
                p.grad = self._flat_grad[off:off + n].view(shape)
        return self._flat_grad

    def grad_bucket(self):
        """[P + 1] floats: ``flat_grad()`` followed by the loss slot (``loss_slot()``)."""
        self.flat_grad()
        return self._grad_bucket

    def loss_slot(self):
        """The last element of the bucket: where ``loss_and_grad`` writes the step's loss when
        ``dp_loss_in_bucket`` is set (``FusedAdam(distributed=True)`` sets it).  It then holds this
        rank's partial loss until ``FusedAdam.step()`` and the GLOBAL loss afterwards."""
        return self.grad_bucket()[-1:]

    # -- library plumbing ---------------------------------------------------------------
    def _get_dims(self):
        if self._dims is not None:
            return self._dims
        if self.solver != 'euler':
            raise ValueError("Unknown solver '{}'.".format(self.solver))
        flags = ((_lib.F_MASKED if self.masked else 0)
                 | (_lib.F_INPUT_CURRENT_T if self.input_current_t else 0)
                 | (_lib.F_RESIDUAL if self.residual_enc_dec else 0)
                 | (_lib.F_LOSS_EASY if self.which_loss == 'easy' else 0)
                 | (_lib.F_USE_RNN if self.use_rnn else 0))
        nh, widths, acts = self._descs[0]
        uniform = (len(set(self._descs)) == 1 and len(set(widths)) <= 1 and len(set(acts)) <= 1)
        if uniform:
            # the three networks share one hidden structure (every configuration of the
            # reference's own scripts): the compact description, which the shape-specialised
            # kernels of the build table are keyed by
            d = _lib.NjodeDims(self.input_size, self.hidden_size, self.output_size, nh,
                               widths[0] if nh else 0, acts[0] if nh else _lib.ACT_TANH, flags)
        else:
            # per-network descriptions (shape-generic kernels)
            d = _lib.NjodeDims(self.input_size, self.hidden_size, self.output_size, 0, 0, 0, flags)
            d.per_net = 1
            for i, (n, ws, as_) in enumerate(self._descs):
                if n > _lib.MAX_HIDDEN:
                    raise NotImplementedError(
                        'libnjode_hip.so runs networks with up to {} hidden layers; got {}'
                        .format(_lib.MAX_HIDDEN, n))
                d.nets[i].n_hidden = n
                for l in range(n):
                    d.nets[i].width[l] = ws[l]
                    d.nets[i].act[l] = as_[l]
        if not _lib.lib().njode_supported(ctypes.byref(d)):
            raise NotImplementedError(
                'libnjode_hip.so has no gfx950 kernels for input_size={}, hidden_size={}, '
                'output_size={}, ode/enc/readout nets (n_hidden, widths, acts)={}, masked={}, '
                'input_current_t={}, residual={}, use_rnn={} (the shape-generic kernels run '
                'every model whose widths are <= {}; with use_rnn 4 x hidden_size too).  {}'
                .format(self.input_size, self.hidden_size, self.output_size, self._descs,
                        self.masked, self.input_current_t, self.residual_enc_dec,
                        self.use_rnn, 1024, _lib.build_info()))
        self._dims = d
        return d

    def _acquire_ws(self, nbytes, device):
        for slot in self._ws_pool:
            if not slot[1] and slot[0].device == device and slot[0].numel() >= nbytes:
                slot[1] = True
                return slot
        slot = [torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device=device), True]
        # keep the pool small: drop free slots that are too small
        self._ws_pool = [s for s in self._ws_pool if s[1]] + [slot]
        return slot

    def _release_ws(self, call):
        if call.ws_slot is not None:
            call.ws_slot[1] = False
            call.ws_slot = None

    def _make_call(self, times, time_ptr, X, obs_idx, delta_t, T, start_X, n_obs_ot,
                   return_path, get_loss, until_T, M, save_bwd, plan_key=None, plan_only=False,
                   want_hT=True, plan=None, rows_in_fwd=False, stream=None):
        """``plan_key`` = the caller's original ``(obs_idx, time_ptr)`` objects (looked up among
        the prefetched plans by identity); ``plan`` = a handle returned by ``prefetch_plan``;
        ``stream`` = the caller's current stream if it already looked it up (a lookup costs
        several microseconds of a step that, at the reference's batch sizes, the host bounds)."""
        L = _lib.lib()
        dev = start_X.device
        if dev.type != 'cuda':
            raise RuntimeError(
                'njode_amd.NJODE runs on an MI355X only (inputs are on {}); there is no '
                'CPU path. Move the model and the batch to a cuda device.'.format(dev))
        dims = self._get_dims()
        self._ensure_flat(full=True)
        if self._flat.device != dev:
            raise RuntimeError('model parameters are on {} but the batch is on {}'.format(
                self._flat.device, dev))
        if self._ring is None:
            self._ring = PinnedRing()
        B = int(start_X.shape[0])
        sched = self._sched_cache.get(times, delta_t, T, until_T)
        time_ptr = np.asarray(time_ptr, dtype=np.int32)
        assert len(times) + 1 == len(time_ptr)
        n_obs = int(time_ptr[-1])

        f32 = torch.float32
        start_X = start_X.to(f32).contiguous()
        X = X.to(device=dev, dtype=f32).contiguous()
        obs_idx_d = obs_idx.to(device=dev, dtype=torch.int32, non_blocking=True).contiguous()
        keep = [start_X, X, obs_idx_d]
        M_ptr = None
        if self.masked:
            assert M is not None
            M = M.to(device=dev, dtype=f32).contiguous()
            keep.append(M)
            M_ptr = M.data_ptr()
        n_ptr = None
        if get_loss:
            n_d = n_obs_ot.to(device=dev, dtype=torch.int32, non_blocking=True).contiguous()
            keep.append(n_d)
            n_ptr = n_d.data_ptr()

        slot_i, pinned = self._ring.acquire(sched.packed_nbytes())
        buf = pinned.numpy()
        K, nt = sched.pack_into(buf, time_ptr)
        base = pinned.data_ptr()
        cs = _lib.NjodeSchedule(K, nt, base, base + 4 * K, base + 8 * K, base + 8 * K + 4 * nt,
                                base + 8 * K + 8 * nt)
        gb = float(self.dp_global_batch if self.dp_global_batch else B)
        cb = _lib.NjodeBatch(B, n_obs, start_X.data_ptr(), X.data_ptr() if n_obs else None,
                             M_ptr, obs_idx_d.data_ptr() if n_obs else None, n_ptr, gb,
                             int(self.dp_path_offset), None)
        flags = ((_lib.C_TRAIN if self.training else 0) | (_lib.C_GET_LOSS if get_loss else 0)
                 | (_lib.C_RETURN_PATH if return_path else 0)
                 | (_lib.C_SAVE_BWD if save_bwd else 0)
                 # autograd route: the saving forward is followed by its backward, so the forward
                 # runs the backward's pass over the observation rows itself (the readouts are
                 # evaluated once instead of twice); carried to the backward by the flags
                 | (_lib.C_ROWS_IN_FWD if (rows_in_fwd and save_bwd) else 0)
                 # the plan decision travels with the call: the backward never re-reads the
                 # pinned schedule buffer, which the ring may have handed to a later forward
                 | _lib.C_SCHED_KNOWN | (_lib.C_SCHED_TAIL if sched.has_tail() else 0)
                 # shape-generic kernels, A/B switch NJODE_GEN_PLAN=lock: read HERE, once per step,
                 # and carried by the flags to the plan, the forward and the backward alike (a
                 # plan prefetched under the other setting no longer matches and is dropped)
                 | (_lib.C_GEN_LOCKSTEP if os.environ.get('NJODE_GEN_PLAN') == 'lock' else 0))
        if plan_only:   # prefetch_plan: the structs of the call, nothing allocated or counted
            return dims, cb, cs, flags, keep + [pinned], slot_i, (B, n_obs, nt, K)
        if plan is not None or (plan_key is not None and self._plans):
            plan = self._take_plan(plan, plan_key, flags, (B, n_obs, nt, K), want_hT)
        if plan is not None:
            cb.plan = plan.buf.data_ptr()
            flags |= _lib.C_PLAN_READY | (plan.flags & _lib.C_NEED_HT)
            use = stream if stream is not None else torch.cuda.current_stream(dev)
            if plan.done is not None:
                use.wait_event(plan.done)
            elif plan.stream is not None and plan.stream.cuda_stream != use.cuda_stream:
                # A deferred plan is ordered by its stream alone -- the one prefetch_plan saw.  A call
                # that consumes it on ANOTHER stream (loss_and_grad(stream=...), a torch.cuda.stream
                # context entered only around the step) would read a plan that may still be half
                # built: launch the job if it is still pending (the library runs it on its own
                # stream) and wait for that stream here.
                L.njode_plan_flush()
                ev = torch.cuda.Event()
                ev.record(plan.stream)
                use.wait_event(ev)
            keep.append(plan)
        need = ctypes.c_size_t(0)
        _lib.check(L.njode_workspace_bytes(ctypes.byref(dims), B, n_obs, nt, K, flags,
                                           ctypes.byref(need)))
        call = _Call()
        call.dims, call.batch, call.sched, call.flags = dims, cb, cs, flags
        call.weight, call.p_drop = float(self.weight), float(self.dropout_rate)
        # a fresh dropout stream per training forward, reproducible from `seed`
        call.seed = (self.seed * 0x9E3779B97F4A7C15 + self._step_counter) & 0xFFFFFFFFFFFFFFFF
        if self.training:
            self._step_counter += 1
        call.ws_slot = self._acquire_ws(need.value, dev)
        call.ws = call.ws_slot[0]
        call.keep = keep + [pinned]
        call.sched_obj, call.time_ptr = sched, time_ptr      # (what _grad_through_hT re-packs)
        return call, sched, slot_i, B

    def _run_forward(self, call, hT, loss, path_h, path_y, slot_i, stream=None):
        L = _lib.lib()
        if stream is None:
            stream = torch.cuda.current_stream()
        rc = L.njode_forward_f32(
            ctypes.byref(call.dims), self._flat.data_ptr(), ctypes.byref(call.batch),
            ctypes.byref(call.sched), call.flags, call.weight, call.p_drop, call.seed,
            hT.data_ptr() if hT is not None else None,
            loss.data_ptr() if loss is not None else None,
            path_h.data_ptr() if path_h is not None else None,
            path_y.data_ptr() if path_y is not None else None,
            call.ws.data_ptr(), call.ws.numel(), stream.cuda_stream)
        self._release_slots(slot_i, stream)
        _lib.check(rc)

    def _release_slots(self, slot_i, stream):
        """After an ``njode_forward_f32`` call: its pinned schedule slot is free once ``stream`` has
        passed it -- and so are the slots of deferred plans, which that call hosted or launched."""
        self._ring.release_after(slot_i, stream)
        ds = getattr(self, '_deferred_stream', None)
        for p in self._deferred_slots:
            if ds is not None and ds.cuda_stream != stream.cuda_stream:
                self._ring.release_after(p, ds)     # (the job ran on the stream it was described on)
            else:
                self._ring.release_with(p, slot_i)
        del self._deferred_slots[:]
        for pl in self._deferred_plans:
            pl.pending = False
        del self._deferred_plans[:]

    def _run_backward(self, call, grad_loss, grad_flat, loss=None, stream=None):
        """``loss`` given: the fused step's backward, which may also produce the loss
        (``NJODE_C_LOSS_IN_BWD``, see ``loss_and_grad``)."""
        L = _lib.lib()
        if stream is None:
            stream = torch.cuda.current_stream()
        if call.ws_slot is None:
            raise RuntimeError(
                'the workspace saved by this forward was already released: a second backward '
                'through the same NJODE forward (retain_graph) is not supported -- run the '
                'forward again')
        if loss is not None:
            _lib.check(L.njode_backward_loss_f32(
                ctypes.byref(call.dims), self._flat.data_ptr(), ctypes.byref(call.batch),
                ctypes.byref(call.sched), call.flags, call.weight, call.p_drop, call.seed,
                grad_loss.data_ptr(), grad_flat.data_ptr(), loss.data_ptr(), call.ws.data_ptr(),
                call.ws.numel(), stream.cuda_stream))
            return
        _lib.check(L.njode_backward_f32(
            ctypes.byref(call.dims), self._flat.data_ptr(), ctypes.byref(call.batch),
            ctypes.byref(call.sched), call.flags, call.weight, call.p_drop, call.seed,
            grad_loss.data_ptr(), grad_flat.data_ptr(), call.ws.data_ptr(), call.ws.numel(),
            stream.cuda_stream))

    def _grad_through_hT(self, call, grad_hT, stream=None):
        """d <grad_hT, hT> / d params (flat) for the saved forward ``call``: see
        ``_hT_and_loss_grads``."""
        L = _lib.lib()
        dev = self._flat.device
        if stream is None:
            stream = torch.cuda.current_stream(dev)
        sched, time_ptr = call.sched_obj, call.time_ptr
        b0 = call.batch
        B, H = int(b0.batch_size), self.hidden_size
        gh = grad_hT.to(device=dev, dtype=torch.float32).reshape(B, H).contiguous()
        n_ptr, n_dummy = b0.n_obs_ot, None
        if not n_ptr:     # (a get_loss=False call: the switched-off loss still wants a divisor)
            n_dummy = torch.ones(B, dtype=torch.int32, device=dev)
            n_ptr = n_dummy.data_ptr()
        # loss_batch_size = inf: every loss term (and its gradient) is scaled by 1 / inf = 0
        cb = _lib.NjodeBatch(b0.batch_size, b0.n_obs, b0.start_X, b0.X, b0.M, b0.obs_idx, n_ptr,
                             float('inf'), b0.path_id_offset, None, None)
        flags = ((call.flags & (_lib.C_TRAIN | _lib.C_SCHED_KNOWN | _lib.C_SCHED_TAIL))
                 | _lib.C_GET_LOSS | _lib.C_SAVE_BWD | _lib.C_GEN_LOCKSTEP)
        slot_i, pinned = self._ring.acquire(sched.packed_nbytes())
        slot = None
        try:
            try:
                K, nt = sched.pack_into(pinned.numpy(), time_ptr)
                base = pinned.data_ptr()
                cs = _lib.NjodeSchedule(K, nt, base, base + 4 * K, base + 8 * K, base + 8 * K + 4 * nt,
                                        base + 8 * K + 8 * nt)
                need = ctypes.c_size_t(0)
                _lib.check(L.njode_workspace_bytes(ctypes.byref(call.dims), B, int(b0.n_obs), nt, K, flags,
                                                   ctypes.byref(need)))
                slot = self._acquire_ws(need.value, dev)
                ws = slot[0]
                hT2 = torch.empty(B, H, dtype=torch.float32, device=dev)
                loss2 = torch.zeros(1, dtype=torch.float32, device=dev)
                rc = L.njode_forward_f32(
                    ctypes.byref(call.dims), self._flat.data_ptr(), ctypes.byref(cb), ctypes.byref(cs), flags,
                    call.weight, call.p_drop, call.seed, hT2.data_ptr(), loss2.data_ptr(), None, None,
                    ws.data_ptr(), ws.numel(), stream.cuda_stream)
            finally:
                self._release_slots(slot_i, stream)   # (the pinned schedule slot: whatever happened)
            _lib.check(rc)
            self._last_hT_replay = hT2       # (the replayed hT: tests compare it with the call's own)
            out = torch.empty_like(self._flat)
            one = torch.ones(1, dtype=torch.float32, device=dev)
            cb.grad_hT = gh.data_ptr()
            _lib.check(L.njode_backward_f32(
                ctypes.byref(call.dims), self._flat.data_ptr(), ctypes.byref(cb), ctypes.byref(cs), flags,
                call.weight, call.p_drop, call.seed, one.data_ptr(), out.data_ptr(), ws.data_ptr(),
                ws.numel(), stream.cuda_stream))
        finally:
            if slot is not None:
                slot[1] = False
        return out

    # -- plan ahead (njode_plan_f32) --------------------------------------------------------
    def plan_defer_ok(self, n_obs):
        """Whether ``prefetch_plan`` defers a plan of ``n_obs`` observation rows into the next
        forward call's ODE-forward launch (``NJODE_C_PLAN_DEFER``) instead of building it on a helper
        stream.  Inside that launch the plan blocks share the memory system with a kernel that
        streams ~2 TB/s and take about twice their time on an idle chip; as long as they end before the
        forward does the step only pays their share of its slots: B = 100 0.250 -> 0.235 ms per step,
        B = 1 000 0.332 -> 0.317, 20 000 paths 0.877 -> 0.85; from 50 000 paths on the plan outlasts the
        forward (1.93 -> 1.95 ms; 125 000: 4.79 -> 4.85), so very large batches keep the helper stream
        (``profiles/r05_plan_in_forward.txt``).  ``NJODE_PLAN_DEFER=0`` switches it off,
        ``NJODE_PLAN_DEFER_MAX`` moves the limit (rows)."""
        if self.masked or self.use_rnn or os.environ.get('NJODE_PLAN_DEFER', '1') == '0':
            return False
        return 0 < n_obs <= int(os.environ.get('NJODE_PLAN_DEFER_MAX', '262144'))

    def prefetch_plan(self, times, time_ptr, X, obs_idx, delta_t, T, start_X, n_obs_ot,
                      M=None, need_hT=None, defer=None):
        """Build the execution plan of a coming ``loss_and_grad`` / training ``forward`` call
        for this batch NOW, on a helper stream, beside whatever the current stream is running
        (``include/njode_hip.h``: ``njode_plan_f32``).  The plan depends on the batch and the
        schedule only, not on the parameters; the call that later receives the same batch
        (same ``obs_idx`` tensor and ``time_ptr`` array, in the order they were prefetched)
        picks it up and skips its own plan stage.  Call it for batch i+1 right before the
        step on batch i.  Returns a handle that may be passed to that call as ``plan=``
        (otherwise the call finds it by the identity of ``obs_idx`` and ``time_ptr``).
        ``need_hT``: the coming call returns hT.  Default True -- ``forward`` always does;
        ``loss_and_grad`` of an unmasked model does not and simply ignores the tail order.
        ``defer`` (round 5; default: on for unmasked models without ``use_rnn``, ``NJODE_PLAN_DEFER=0``
        turns it off): no helper stream and no events -- the plan is built by the first blocks of the
        ODE-forward launch of the NEXT forward call on the current stream (``NJODE_C_PLAN_DEFER``,
        ``include/njode_hip.h``), i.e. of the step on batch i when this is called for batch i+1
        right before it; the batch's arrays must be ready on the current stream."""
        need_hT = True if need_hT is None else bool(need_hT)
        if defer is None:
            defer = self.plan_defer_ok(int(np.asarray(time_ptr)[-1]))
        dims, cb, cs, flags, keep, slot_i, (B, n_obs, nt, K) = self._make_call(
            times, time_ptr, X, obs_idx, delta_t, T, start_X, n_obs_ot, False, True, False, M,
            save_bwd=True, plan_only=True)
        if need_hT:
            flags |= _lib.C_NEED_HT
        L = _lib.lib()
        need = ctypes.c_size_t(0)
        _lib.check(L.njode_plan_bytes(ctypes.byref(dims), B, n_obs, nt, K, flags,
                                      ctypes.byref(need)))
        dev = start_X.device
        buf = None
        for i, cand in enumerate(self._plan_pool):
            if cand.device == dev and cand.numel() >= need.value:
                buf = self._plan_pool.pop(i)     # (list.remove would compare tensors by value)
                break
        if buf is None:
            buf = torch.empty(int(need.value * 1.25) + 4096, dtype=torch.uint8, device=dev)
        if defer:
            # one queue: the job is only described now; the next forward call on this stream carries
            # it (or launches it in front of itself).  Its pinned schedule slot stays held until then.
            cur = torch.cuda.current_stream(dev)
            older, older_plans = self._deferred_slots, self._deferred_plans
            self._deferred_slots, self._deferred_plans = [slot_i], []
            self._deferred_stream = cur
            self._ring.hold(slot_i)
            rc = L.njode_plan_f32(ctypes.byref(dims), ctypes.byref(cb), ctypes.byref(cs),
                                  flags | _lib.C_PLAN_DEFER, buf.data_ptr(), buf.numel(), cur.cuda_stream)
            # (a job that was still pending -- two prefetches without a forward between them -- has
            # just been launched on `cur` by the library: its schedule slot is free behind that)
            for p in older:
                self._ring.release_after(p, cur)
            for pl in older_plans:
                pl.pending = False
            _lib.check(rc)
            done = None
        else:
            if self._plan_stream is None:
                self._plan_stream = torch.cuda.Stream(device=dev)
            side = self._plan_stream
            side.wait_stream(torch.cuda.current_stream(dev))   # the batch's arrays are ready by now
            # (the library call, the ring and the event all take the stream explicitly: no
            # `with torch.cuda.stream(side)` -- entering and leaving it costs ~15 us of host time)
            rc = L.njode_plan_f32(ctypes.byref(dims), ctypes.byref(cb), ctypes.byref(cs), flags,
                                  buf.data_ptr(), buf.numel(), side.cuda_stream)
            self._ring.release_after(slot_i, side)
            _lib.check(rc)
            done = torch.cuda.Event()
            done.record(side)
        plan = _Plan()
        plan.buf, plan.done, plan.flags, plan.sizes, plan.keep = buf, done, flags, (B, n_obs, nt, K), keep
        plan.pool = self._plan_pool
        plan.obs_idx, plan.time_ptr, plan.taken = obs_idx, time_ptr, False
        plan.obs_version = getattr(obs_idx, '_version', 0)
        plan.pending = bool(defer)
        plan.stream = cur if defer else None
        if defer:
            self._deferred_plans.append(plan)
        self._plans.append(plan)
        # plans nobody picks up (a prefetched batch that is then never stepped on) must not pile
        # up: keep the four newest
        while len(self._plans) > 4:
            self._plans.pop(0)
        return plan

    def _take_plan(self, plan, key, flags, sizes, want_hT):
        """The prefetched plan of this call, or None (the call then builds its own in line).
        ``plan`` given: the handle ``prefetch_plan`` returned; else the oldest queued plan whose
        ``obs_idx`` / ``time_ptr`` ARE (identity) the caller's objects.  A plan that was made for
        a different kind of call (sizes, flags, no tail order although hT is wanted) or whose
        ``obs_idx`` was modified in place since is dropped, never used and never left to block
        the plans queued behind it."""
        if plan is None:
            obs_idx, time_ptr = key
            for cand in self._plans:
                if cand.obs_idx is obs_idx and cand.time_ptr is time_ptr:
                    plan = cand
                    break
            if plan is None:
                return None
        if plan.taken:
            raise RuntimeError('this prefetched plan was already consumed by an earlier call')
        if plan in self._plans:
            self._plans.remove(plan)
        plan.taken = True
        want = flags & ~(_lib.C_LOSS_IN_BWD | _lib.C_PLAN_READY | _lib.C_ROWS_IN_FWD)
        ok = (plan.sizes == sizes and (plan.flags & ~_lib.C_NEED_HT) == want
              and getattr(plan.obs_idx, '_version', 0) == plan.obs_version
              and (not want_hT or self.masked or (plan.flags & _lib.C_NEED_HT)))
        return plan if ok else None

    # -- forward -------------------------------------------------------------------------
    def forward(self, times, time_ptr, X, obs_idx, delta_t, T, start_X, n_obs_ot,
                return_path=False, get_loss=True, until_T=False, M=None, plan=None):
        """Same contract as the reference's ``NJODE.forward`` (``models.py:379-518``):
        returns ``(hT, loss)`` or ``(hT, loss, path_t, path_h, path_y)``; ``loss`` is
        the Python int 0 when ``get_loss=False``.  ``plan`` (optional, not in the reference):
        the handle ``prefetch_plan`` returned for this batch."""
        want_grad = (torch.is_grad_enabled() and get_loss
                     and any(p.requires_grad for p in self.parameters()))
        if self.torch_library_op and not return_path:
            return self._forward_via_op(times, time_ptr, X, obs_idx, delta_t, T, start_X,
                                        n_obs_ot, get_loss, until_T, M, want_grad)
        call, sched, slot_i, B = self._make_call(
            times, time_ptr, X, obs_idx, delta_t, T, start_X, n_obs_ot, return_path,
            get_loss, until_T, M, save_bwd=want_grad,
            plan_key=(obs_idx, time_ptr), want_hT=True, plan=plan, rows_in_fwd=True)
        dev = start_X.device
        hT = torch.empty(B, self.hidden_size, dtype=torch.float32, device=dev)
        loss = torch.zeros(1, dtype=torch.float32, device=dev) if get_loss else None
        path_h = path_y = None
        if return_path:
            path_h = torch.empty(sched.n_rows, B, self.hidden_size, dtype=torch.float32,
                                 device=dev)
            path_y = torch.empty(sched.n_rows, B, self.output_size, dtype=torch.float32,
                                 device=dev)
        try:
            self._run_forward(call, hT, loss, path_h, path_y, slot_i)
        except Exception:
            self._release_ws(call)
            raise
        if want_grad:
            self._ensure_flat()
            loss_out, hT = _NJODEFunction.apply(self, call, loss, hT, *self._flat_params)
            loss_out = loss_out.reshape(())
        else:
            self._release_ws(call)
            loss_out = loss.reshape(()) if get_loss else 0
            if (not get_loss and torch.is_grad_enabled()
                    and any(p.requires_grad for p in self.parameters())):
                self._ensure_flat()
                hT = _NJODEhTOnlyFunction.apply(self, call, hT, *self._flat_params)
        if get_loss and not self.device_outputs:
            loss_out = loss_out.cpu()       # reference harness calls .numpy() on it
        if return_path:
            return hT, loss_out, sched.path_t.copy(), path_h, path_y
        return hT, loss_out

    def _forward_via_op(self, times, time_ptr, X, obs_idx, delta_t, T, start_X, n_obs_ot,
                        get_loss, until_T, M, want_grad):
        """The same forward as one ``torch.library`` operator (``njode_amd/ops.py``)."""
        from . import ops
        self._ensure_flat()
        hT, loss, _ = torch.ops.njode_amd.forward(
            list(self._flat_params), start_X, X, obs_idx, n_obs_ot if get_loss else None,
            M if self.masked else None, torch.as_tensor(np.asarray(times, dtype=np.float64)),
            torch.as_tensor(np.asarray(time_ptr, dtype=np.int64)), float(delta_t), float(T),
            ops.register_model(self), bool(get_loss), bool(until_T), bool(want_grad))
        if not get_loss:
            return hT, 0
        loss_out = loss.reshape(())
        if not self.device_outputs:
            loss_out = loss_out.cpu()
        return hT, loss_out

    # -- fused training step (no autograd graph) ------------------------------------------
    def loss_and_grad(self, times, time_ptr, X, obs_idx, delta_t, T, start_X, n_obs_ot,
                      M=None, plan=None):
        """Forward + exact gradient in two library calls, no autograd bookkeeping:
        returns ``(None, loss)`` (device loss tensor; hT is not computed -- it would
        cost a per-path tail evolve nobody reads) and fills ``flat_grad()`` (whose
        slices are the parameters' ``.grad``).  The calls carry ``NJODE_C_LOSS_IN_BWD``:
        on the segment plan the forward skips its readout/loss pass over the observation
        rows and the backward, which evaluates the same readouts anyway, writes the loss.
        Used by the build's harness and bench."""
        grad = self.flat_grad()
        dev = start_X.device
        stream = torch.cuda.current_stream(dev)      # (one lookup per step: ~7 us each)
        call, sched, slot_i, B = self._make_call(
            times, time_ptr, X, obs_idx, delta_t, T, start_X, n_obs_ot, False, True, False,
            M, save_bwd=True, plan_key=(obs_idx, time_ptr), want_hT=self.masked, plan=plan,
            stream=stream)
        # (always written: sum of the terms; data parallel: written INTO the gradient bucket, see
        # loss_slot())
        loss = (self.loss_slot() if self.dp_loss_in_bucket
                else torch.empty(1, dtype=torch.float32, device=dev))
        # hT is only skipped on the segment plan (unmasked): there it would cost an extra
        # per-path tail evolve; the lockstep plan produces it anyway
        hT = (torch.empty(B, self.hidden_size, dtype=torch.float32, device=dev)
              if self.masked else None)
        call.flags |= _lib.C_LOSS_IN_BWD
        self._last_stream = stream
        try:
            self._run_forward(call, hT, loss, None, None, slot_i, stream)
            if self._ones is None or self._ones.device != dev:
                self._ones = torch.ones(1, dtype=torch.float32, device=dev)
            self._run_backward(call, self._ones, grad, loss=loss, stream=stream)
        finally:
            self._release_ws(call)
        return hT, loss.reshape(())

    # -- evaluation helpers ---------------------------------------------------------------
    def evaluate(self, times, time_ptr, X, obs_idx, delta_t, T, start_X, n_obs_ot,
                 stockmodel, cond_exp_fun_kwargs=None,
                 diff_fun=lambda x, y: np.mean((x - y) ** 2), return_paths=False, M=None):
        """Distance of the predicted path to the true conditional expectation
        (reference ``models.py:521-562``)."""
        self.eval()
        _, _, path_t, path_h, path_y = self.forward(
            times, time_ptr, X, obs_idx, delta_t, T, start_X, None, return_path=True,
            get_loss=False, until_T=True, M=M)
        _, true_path_t, true_path_y = stockmodel.compute_cond_exp(
            times, time_ptr, X.detach().cpu().numpy(), obs_idx.detach().cpu().numpy(),
            delta_t, T, start_X.detach().cpu().numpy(), n_obs_ot.detach().cpu().numpy(),
            return_path=True, get_loss=False)
        eval_loss = diff_fun(path_y.detach().cpu().numpy(), true_path_y)
        if return_paths:
            return eval_loss, path_t, true_path_t, path_y, true_path_y
        return eval_loss

    def get_pred(self, times, time_ptr, X, obs_idx, delta_t, T, start_X, M=None):
        """Predicted path (reference ``models.py:564-584``)."""
        self.eval()
        _, _, path_t, path_h, path_y = self.forward(
            times, time_ptr, X, obs_idx, delta_t, T, start_X, None, return_path=True,
            get_loss=False, until_T=True, M=M)
        if not self.device_outputs:
            path_y = path_y.cpu()           # reference harness calls .numpy() on it
        return {'pred': path_y, 'pred_t': path_t}


class FusedAdam:
    """``torch.optim.Adam(lr, betas, eps, weight_decay)`` on the model's flat
    parameter vector in one kernel (``njode_adam_step_f32``), optionally preceded by
    the data-parallel gradient all-reduce (one RCCL all-reduce of P floats)."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 process_group=None, distributed=False):
        self.model = model
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.step_count = 0
        flat = model.flat_parameters()
        self.exp_avg = torch.zeros_like(flat)
        self.exp_avg_sq = torch.zeros_like(flat)
        self.distributed = distributed
        # data parallel: the step's loss travels in the gradient's bucket (one collective)
        model.dp_loss_in_bucket = bool(distributed)
        self.group = process_group
        self.time_allreduce = False
        self._allreduce_events = []

    def zero_grad(self):
        pass  # loss_and_grad overwrites the flat gradient

    def step(self):
        m = self.model
        flat, grad = m.flat_parameters(), m.flat_grad()
        if m._flat_present is not None:
            # slots without a parameter (bias=False) take no update: zero gradient, zero moments,
            # zero L2 term -> they stay exactly 0, as the C ABI requires (include/njode_hip.h)
            grad.mul_(m._flat_present)
        if self.exp_avg.device != flat.device:
            self.exp_avg = self.exp_avg.to(flat.device)
            self.exp_avg_sq = self.exp_avg_sq.to(flat.device)
        if self.distributed:
            bucket = m.grad_bucket()      # gradient [P] + the scalar loss: ONE all-reduce
            if self.time_allreduce:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
                torch.distributed.all_reduce(bucket, group=self.group)
                ev[1].record()
                self._allreduce_events.append(ev)
            else:
                torch.distributed.all_reduce(bucket, group=self.group)
        self.step_count += 1
        _lib.check(_lib.lib().njode_adam_step_f32(
            flat.data_ptr(), grad.data_ptr(), self.exp_avg.data_ptr(),
            self.exp_avg_sq.data_ptr(), flat.numel(), self.lr, self.betas[0], self.betas[1],
            self.eps, self.weight_decay, self.step_count, 1.0,
            _raw_current_stream(flat.device)))

    def allreduce_ms(self):
        """Mean device time of the gradient all-reduce since the last call (``time_allreduce``
        must be set; synchronises)."""
        if not self._allreduce_events:
            return None
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b in self._allreduce_events]
        self._allreduce_events = []
        return sum(ms) / len(ms)

    # -- checkpoints: torch.optim.Adam's layout, so the reference's get_ckpt_model
    # (optimizer.load_state_dict, models.py:63) reads a checkpoint of the fused loop and
    # this class reads the reference's
    def _param_index(self):
        """[(offset, size, shape)] of the flat slices in ``model.parameters()`` order."""
        m = self.model
        m._ensure_flat()
        by_id = {id(p): sl for sl, p in zip(m._param_slices, m._flat_params)}
        return [by_id[id(p)] for p in m.parameters()]

    def state_dict(self):
        idx = self._param_index()
        state = {}
        if self.step_count > 0:
            for i, (off, n, shape) in enumerate(idx):
                state[i] = {'step': torch.tensor(float(self.step_count)),
                            'exp_avg': self.exp_avg[off:off + n].view(shape).clone(),
                            'exp_avg_sq': self.exp_avg_sq[off:off + n].view(shape).clone()}
        group = {'lr': self.lr, 'betas': tuple(self.betas), 'eps': self.eps,
                 'weight_decay': self.weight_decay, 'amsgrad': False, 'maximize': False,
                 'foreach': None, 'capturable': False, 'differentiable': False, 'fused': None,
                 'decoupled_weight_decay': False, 'params': list(range(len(idx)))}
        return {'state': state, 'param_groups': [group]}

    def load_state_dict(self, sd):
        if 'param_groups' not in sd:      # round-1 private layout
            self.step_count = int(sd['step'])
            self.exp_avg.copy_(sd['exp_avg'])
            self.exp_avg_sq.copy_(sd['exp_avg_sq'])
            return
        idx = self._param_index()
        group = sd['param_groups'][0]
        if len(sd['param_groups']) != 1 or len(group['params']) != len(idx):
            raise ValueError('optimizer state has {} groups / {} parameters, the model has {} '
                             'parameters'.format(len(sd['param_groups']), len(group['params']),
                                                 len(idx)))
        if group.get('amsgrad'):
            raise ValueError('FusedAdam does not implement amsgrad')
        self.lr, self.betas = group['lr'], tuple(group['betas'])
        self.eps, self.weight_decay = group['eps'], group['weight_decay']
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        steps = set()
        for pos, key in enumerate(group['params']):
            st = sd['state'].get(key)
            if st is None:
                continue
            off, n, shape = idx[pos]
            steps.add(int(float(st['step'])))
            self.exp_avg[off:off + n].copy_(st['exp_avg'].reshape(-1))
            self.exp_avg_sq[off:off + n].copy_(st['exp_avg_sq'].reshape(-1))
        if len(steps) > 1:
            raise ValueError('parameters with different step counts: {}'.format(sorted(steps)))
        self.step_count = steps.pop() if steps else 0
