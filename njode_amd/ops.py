"""
``torch.library`` registration of the NJ-ODE forward (BASELINE north_star: "called from Python
via PyTorch-ROCm custom ops"): ``torch.ops.njode_amd.forward`` is one dispatcher-visible,
``torch.compile``-opaque operator around ``njode_forward_f32`` with a registered fake (shape)
implementation and a registered autograd formula that calls ``njode_backward_f32``.

The model class uses it when constructed with ``options={'torch_library_op': True}``; the
default route is the ``torch.autograd.Function`` of ``models.py`` (same two library calls).
Both routes produce the same numbers (``tests/test_hip_torch_op.py``).

Operator schema::

    njode_amd::forward(Tensor[] params, Tensor start_X, Tensor X, Tensor obs_idx,
                       Tensor? n_obs_ot, Tensor? M, Tensor times, Tensor time_ptr,
                       float delta_t, float T, int model_id, bool get_loss, bool until_T,
                       bool save_bwd) -> (Tensor hT, Tensor loss, Tensor call_id)

* ``params``: the model's parameters in ``state_dict`` order (views of its flat vector); their
  gradients are what the autograd formula returns;
* ``times`` (float64) / ``time_ptr`` (int64): CPU tensors, the reference's host-side arrays;
* ``model_id``: handle of the ``NJODE`` instance (shapes, options and the workspace pool live
  there; the registry holds weak references);
* ``call_id`` (CPU int64 scalar): handle of the saved forward, consumed by the backward (and
  released with the autograd node if no backward ever runs).

Hidden state: a training-mode call advances the model's dropout step counter (the seed of the
next call's masks) although the schema declares no mutated argument -- two calls with equal
arguments are NOT interchangeable, so do not let a graph pass deduplicate or reorder them
(the op is opaque to ``torch.compile``, which keeps the calls in program order).
"""
import weakref
from typing import List, Optional, Tuple

import torch
from torch import Tensor

_MODELS = weakref.WeakValueDictionary()
_CALLS = {}
_next_call = [1]


def register_model(model):
    _MODELS[id(model)] = model
    return id(model)


@torch.library.custom_op('njode_amd::forward', mutates_args=())
def forward(params: List[Tensor], start_X: Tensor, X: Tensor, obs_idx: Tensor,
            n_obs_ot: Optional[Tensor], M: Optional[Tensor], times: Tensor, time_ptr: Tensor,
            delta_t: float, T: float, model_id: int, get_loss: bool, until_T: bool,
            save_bwd: bool) -> Tuple[Tensor, Tensor, Tensor]:
    model = _MODELS[model_id]
    call, sched, slot_i, B = model._make_call(
        times.numpy(), time_ptr.numpy(), X, obs_idx, delta_t, T, start_X, n_obs_ot, False,
        get_loss, until_T, M, save_bwd=save_bwd)
    dev = start_X.device
    hT = torch.empty(B, model.hidden_size, dtype=torch.float32, device=dev)
    loss = torch.zeros(1, dtype=torch.float32, device=dev)
    try:
        model._run_forward(call, hT, loss if get_loss else None, None, None, slot_i)
    except Exception:
        model._release_ws(call)
        raise
    cid = 0
    if save_bwd:
        cid = _next_call[0]
        _next_call[0] += 1
        _CALLS[cid] = call
    else:
        model._release_ws(call)
    return hT, loss, torch.tensor(cid, dtype=torch.int64)


@forward.register_fake
def _(params, start_X, X, obs_idx, n_obs_ot, M, times, time_ptr, delta_t, T, model_id,
      get_loss, until_T, save_bwd):
    model = _MODELS[model_id]
    hT = start_X.new_empty((start_X.shape[0], model.hidden_size), dtype=torch.float32)
    return hT, start_X.new_empty((1,), dtype=torch.float32), torch.empty((), dtype=torch.int64)


def _drop_call(call_id):
    """The autograd node of a saving forward is gone (back-propagated, or its graph dropped
    without a backward, e.g. a validation loss computed without no_grad): release the saved
    call, which returns its workspace to the model's pool."""
    call = _CALLS.pop(call_id, None)
    if call is not None and call.ws_slot is not None:
        call.ws_slot[1] = False
        call.ws_slot = None


def _setup_context(ctx, inputs, output):
    ctx.model_id = inputs[10]
    ctx.n_params = len(inputs[0])
    ctx.call_id = int(output[2])
    ctx.set_materialize_grads(False)     # an unused output's gradient arrives as None, not zeros
    if ctx.call_id:
        weakref.finalize(ctx, _drop_call, ctx.call_id)


def _backward(ctx, grad_hT, grad_loss, grad_id):
    from .models import _hT_and_loss_grads
    model = _MODELS[ctx.model_id]
    call = _CALLS.pop(ctx.call_id, None)
    if call is None:
        raise RuntimeError('njode_amd::forward was not run with save_bwd=True, or its backward '
                           'already ran (a second backward is not supported)')
    # (round 4 received grad_hT here and dropped it; now the hT term is differentiated too)
    grad_flat = _hT_and_loss_grads(model, call, grad_loss, grad_hT)
    grads = [grad_flat[off:off + n].view(shape) for (off, n, shape) in model._param_slices]
    return (grads,) + (None,) * 13


torch.library.register_autograd('njode_amd::forward', _backward, setup_context=_setup_context)
