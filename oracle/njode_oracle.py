"""
ORACLE -- test infrastructure, not product code.

CPU restatement (plain PyTorch fp32, ATen CPU ops + autograd) of the NJ-ODE
forward/training path of the reference, ``NJODE/models.py``.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; the product path (``njode_amd``) never does and fails loudly when
its HIP library is missing.

Parity pinning: ``tests/test_oracle_golden.py`` checks this restatement against
golden vectors produced by importing the reference itself in the build
container (``tests/golden/make_golden.py``; inputs, outputs, gradients, Adam
steps, and the known-answer triples of the reference's shipped checkpoints).

What each function follows (reference file:line):

* ``mlp`` / ``OracleNet``       -- ``models.py:140-166`` (get_ffnn) parameter
  naming ``<prefix>.{0,3,6,..}.{weight,bias}``
* ``ode_rhs``                   -- ``models.py:188-199`` (ODEFunc.forward)
* ``ffnn``                      -- ``models.py:261-276`` (FFNN.forward, residual
  cases 0/1/2, masked input)
* ``gru_jump``                  -- ``models.py:202-217`` (GRUCell)
* ``paper_loss``                -- ``models.py:71-126`` (compute_loss / _2)
* ``euler_clock``               -- ``models.py:430-439, 497-505`` (float64 clock)
* ``OracleNJODE.forward``       -- ``models.py:379-518``
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

EPS = 1e-10


def layer_sizes(in_size, out_size, nn_desc):
    """[(in, out, activation-or-None), ...] of a get_ffnn network."""
    if nn_desc is None:
        return [(in_size, out_size, None)]
    sizes = []
    prev = in_size
    for width, act in nn_desc:
        sizes.append((prev, int(width), act))
        prev = int(width)
    sizes.append((prev, out_size, None))
    return sizes


class OracleNet:
    """One feed-forward net: Linear (+act +dropout) ... Linear.  Parameters live
    in the owning model's dict under the reference's state_dict names: layer k
    of the Sequential sits at index 3*k (Linear, act, Dropout triples)."""

    def __init__(self, prefix, in_size, out_size, nn_desc, bias=True):
        self.prefix = prefix
        self.layers = layer_sizes(in_size, out_size, nn_desc)
        self.bias = bias

    def names(self):
        out = []
        for k in range(len(self.layers)):
            out.append('{}.{}.weight'.format(self.prefix, 3 * k))
            if self.bias:
                out.append('{}.{}.bias'.format(self.prefix, 3 * k))
        return out

    def shapes(self):
        out = {}
        for k, (i, o, _) in enumerate(self.layers):
            out['{}.{}.weight'.format(self.prefix, 3 * k)] = (o, i)
            if self.bias:
                out['{}.{}.bias'.format(self.prefix, 3 * k)] = (o,)
        return out

    def __call__(self, params, x, p_drop, training):
        for k, (_, _, act) in enumerate(self.layers):
            w = params['{}.{}.weight'.format(self.prefix, 3 * k)]
            b = params.get('{}.{}.bias'.format(self.prefix, 3 * k))
            x = F.linear(x, w, b)
            if act is not None:
                x = torch.tanh(x) if act == 'tanh' else torch.relu(x)
                x = F.dropout(x, p_drop, training)
        return x


def paper_loss(which, X_obs, Y_obs, Y_obs_bj, n_obs_ot, batch_size, weight,
               M_obs=None):
    """standard: sum_obs (2w|X-Y| + 2(1-w)|Ybj-Y|)^2 / n_obs / B;
    easy: (w|X-Y| + (1-w)|Ybj-X|)^2; norms are sqrt(sum_d m (.)^2 + eps)."""
    m = 1.0 if M_obs is None else M_obs
    after = torch.sqrt(torch.sum(m * (X_obs - Y_obs) ** 2, dim=1) + EPS)
    if which == 'standard':
        before = torch.sqrt(torch.sum(m * (Y_obs_bj - Y_obs) ** 2, dim=1) + EPS)
        inner = (2 * weight * after + 2 * (1 - weight) * before) ** 2
    elif which == 'easy':
        before = torch.sqrt(torch.sum(m * (Y_obs_bj - X_obs) ** 2, dim=1) + EPS)
        inner = (weight * after + (1 - weight) * before) ** 2
    else:
        raise ValueError(which)
    return torch.sum(inner / n_obs_ot) / batch_size


def euler_clock(current_time, target, delta_t):
    """Yield (step_size, time_before_step) until ``target`` is reached, with
    the reference's float64 rules: continue while ``t < target - 1e-10*dt``;
    a full ``dt`` while ``t < target - dt``, else the remainder."""
    while current_time < target - 1e-10 * delta_t:
        if current_time < target - delta_t:
            step = delta_t
        else:
            step = target - current_time
        yield step, current_time
        current_time = current_time + step


class OracleNJODE:
    """Functional NJ-ODE.  ``params``: dict name -> fp32 tensor with the
    reference's state_dict keys (``ode_f.f.*``, ``encoder_map.ffnn.*``,
    ``readout_map.ffnn.*``, ``obs_c.gru_d.*``)."""

    def __init__(self, input_size, hidden_size, output_size, ode_nn, readout_nn,
                 enc_nn, use_rnn=False, bias=True, dropout_rate=0.0,
                 weight=0.5, which_loss='standard', residual_enc_dec=True,
                 input_current_t=False, masked=False):
        self.d, self.H, self.d_out = input_size, hidden_size, output_size
        self.p_drop = dropout_rate
        self.weight = weight
        self.which_loss = which_loss
        self.input_current_t = input_current_t
        self.masked = masked
        self.use_rnn = use_rnn
        self.bias = bias
        self.training = False
        extra = 3 if input_current_t else 2
        self.ode = OracleNet('ode_f.f', input_size + hidden_size + extra,
                             hidden_size, ode_nn, bias)
        self.enc = OracleNet('encoder_map.ffnn',
                             2 * input_size if masked else input_size,
                             hidden_size, enc_nn, bias)
        self.dec = OracleNet('readout_map.ffnn', hidden_size, output_size,
                             readout_nn, bias)
        self.enc_res = self._residual_case(input_size, hidden_size,
                                           residual_enc_dec)
        self.dec_res = self._residual_case(hidden_size, output_size,
                                           residual_enc_dec)

    @staticmethod
    def _residual_case(n_in, n_out, residual):
        if not residual:
            return (0, 1)
        if n_in <= n_out:
            if n_out % n_in:
                raise ValueError('for residual: output_size needs to be '
                                 'multiple of input_size')
            return (1, n_out // n_in)
        if n_in % n_out:
            raise ValueError('for residual: input_size needs to be '
                             'multiple of output_size')
        return (2, n_in // n_out)

    # -- parameters -------------------------------------------------------------
    def param_shapes(self):
        shapes = {}
        for net in (self.ode, self.enc, self.dec):
            shapes.update(net.shapes())
        if self.use_rnn:
            shapes['obs_c.gru_d.weight_ih'] = (3 * self.H, self.d)
            shapes['obs_c.gru_d.weight_hh'] = (3 * self.H, self.H)
            if self.bias:
                shapes['obs_c.gru_d.bias_ih'] = (3 * self.H,)
                shapes['obs_c.gru_d.bias_hh'] = (3 * self.H,)
        return shapes

    def init_params(self, seed=0):
        """Xavier-uniform weights, zero bias (``models.py:21-26``); own RNG
        stream -- tests that need the reference's exact init load its
        state_dict from a golden file instead."""
        g = torch.Generator().manual_seed(seed)
        params = {}
        for name, shape in self.param_shapes().items():
            if name.startswith('obs_c'):
                k = 1.0 / math.sqrt(self.H)
                t = (torch.rand(shape, generator=g) * 2 - 1) * k
            elif name.endswith('weight'):
                bound = math.sqrt(6.0 / (shape[0] + shape[1]))
                t = (torch.rand(shape, generator=g) * 2 - 1) * bound
            else:
                t = torch.zeros(shape)
            params[name] = t.float()
        return params

    # -- sub-maps ---------------------------------------------------------------
    def ffnn(self, net, res, params, x, mask=None):
        inp = torch.tanh(x)
        if mask is not None:
            inp = torch.cat((inp, mask), 1)
        out = net(params, inp, self.p_drop, self.training)
        case, mult = res
        if case == 1:
            return x.repeat(1, mult) + out
        if case == 2:
            return torch.mean(torch.stack(x.chunk(mult, dim=1)), dim=0) + out
        return out

    def encode(self, params, x, mask=None):
        return self.ffnn(self.enc, self.enc_res, params, x, mask)

    def readout(self, params, h):
        return self.ffnn(self.dec, self.dec_res, params, h)

    def ode_rhs(self, params, x, h, tau, tdiff):
        parts = [torch.tanh(x), torch.tanh(h), tau, tdiff]
        if self.input_current_t:
            parts.append(tau + tdiff)
        return self.ode(params, torch.cat(parts, dim=1), self.p_drop,
                        self.training)

    def gru_jump(self, params, h, X_obs, i_obs):
        gi = F.linear(torch.tanh(X_obs), params['obs_c.gru_d.weight_ih'],
                      params.get('obs_c.gru_d.bias_ih'))
        hp = torch.tanh(h[i_obs])
        gh = F.linear(hp, params['obs_c.gru_d.weight_hh'],
                      params.get('obs_c.gru_d.bias_hh'))
        i_r, i_z, i_n = gi.chunk(3, 1)
        h_r, h_z, h_n = gh.chunk(3, 1)
        r = torch.sigmoid(i_r + h_r)
        z = torch.sigmoid(i_z + h_z)
        n = torch.tanh(i_n + r * h_n)
        new = (1 - z) * n + z * hp
        out = h.clone()
        out[i_obs] = new
        return out

    # -- the path -----------------------------------------------------------------
    def forward(self, params, times, time_ptr, X, obs_idx, delta_t, T, start_X,
                n_obs_ot, return_path=False, get_loss=True, until_T=False,
                M=None):
        B = start_X.shape[0]
        if self.masked:
            h = self.encode(params, start_X, torch.zeros_like(start_X))
        else:
            h = self.encode(params, start_X)
        last_X = start_X
        tau = torch.zeros(B, 1)
        now = 0.0
        loss = 0
        rec_t, rec_h, rec_y = [], [], []

        def record(t, h_):
            if return_path:
                rec_t.append(t)
                rec_h.append(h_)
                rec_y.append(self.readout(params, h_))

        record(0, h)
        assert len(times) + 1 == len(time_ptr)

        def evolve(h_, now_, target):
            for step, t0 in euler_clock(now_, target, delta_t):
                # python-float minus fp32 tensor: ATen casts the scalar to fp32
                h_ = h_ + step * self.ode_rhs(params, last_X, h_, tau, t0 - tau)
                now_ = t0 + step
                record(now_, h_)
            return h_, now_

        for i, obs_time in enumerate(times):
            h, now = evolve(h, now, obs_time)
            lo, hi = int(time_ptr[i]), int(time_ptr[i + 1])
            X_obs = X[lo:hi]
            i_obs = obs_idx[lo:hi]
            M_obs = M[lo:hi] if self.masked else None

            Y_bj = self.readout(params, h)
            if self.use_rnn:
                h = self.gru_jump(params, h, X_obs, i_obs)
            else:
                if self.masked:
                    x_in = X_obs * M_obs + (1 - M_obs) * Y_bj[i_obs]
                    new = self.encode(params, x_in, M_obs)
                else:
                    new = self.encode(params, X_obs)
                h = h.clone()
                h[i_obs] = new
            Y = self.readout(params, h)

            if get_loss:
                loss = loss + paper_loss(
                    self.which_loss, X_obs, Y[i_obs], Y_bj[i_obs],
                    n_obs_ot[i_obs], B, self.weight, M_obs)

            last_X = last_X.clone()
            last_X[i_obs] = Y[i_obs] if self.masked else X_obs
            tau = tau.clone()
            tau[i_obs] = float(obs_time)
            if return_path:
                rec_t.append(obs_time)
                rec_h.append(h)
                rec_y.append(Y)

        if until_T:
            h, now = evolve(h, now, T)

        if return_path:
            return (h, loss, np.array(rec_t), torch.stack(rec_h),
                    torch.stack(rec_y))
        return h, loss


def make_oracle(cfg):
    """Build from a config dict with the NJODE constructor's keys."""
    opts = cfg.get('options', {})
    return OracleNJODE(
        cfg['input_size'], cfg['hidden_size'], cfg['output_size'],
        cfg['ode_nn'], cfg['readout_nn'], cfg['enc_nn'],
        use_rnn=cfg.get('use_rnn', False), bias=cfg.get('bias', True),
        dropout_rate=cfg.get('dropout_rate', 0.0),
        weight=cfg.get('weight', 0.5),
        which_loss=opts.get('which_loss', 'standard'),
        residual_enc_dec=opts.get('residual_enc_dec', True),
        input_current_t=opts.get('input_current_t', False),
        masked=opts.get('masked', False))


def train_step(model, params, opt, batch, delta_t, T):
    """One optimizer step with the reference loop's semantics
    (``train.py:492-523``): zero_grad, recount n_obs_ot, forward, backward,
    Adam step.  ``params`` tensors must have requires_grad and be in ``opt``."""
    opt.zero_grad()
    B = batch['start_X'].shape[0]
    n_obs_ot = torch.bincount(batch['obs_idx'], minlength=B)
    model.training = True
    _, loss = model.forward(params, batch['times'], batch['time_ptr'],
                            batch['X'], batch['obs_idx'], delta_t, T,
                            batch['start_X'], n_obs_ot, return_path=False,
                            get_loss=True, M=batch.get('M'))
    loss.backward()
    opt.step()
    return loss.detach()
