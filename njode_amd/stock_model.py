"""
Synthetic SDE data and analytic conditional expectations for the NJ-ODE hot path.

Host-side (numpy, float64) producer of the inputs the HIP path consumes and of
the analytic ground truth it is checked against.  Mirrors the *interface* of the
reference's ``NJODE/stock_model.py`` (class names, ctor keywords, method names,
return conventions) for the models BASELINE.json's configs use:

* ``BlackScholes``      -- reference ``stock_model.py:339-375``
* ``OrnsteinUhlenbeck`` -- reference ``stock_model.py:378-418``
* ``Heston``            -- reference ``stock_model.py:161-221``
* ``compute_cond_exp``  -- reference ``stock_model.py:50-151``
* ``compute_loss``      -- reference ``stock_model.py:471-481``

The generators are vectorised over paths (the reference runs a Python double
loop, 12.6 s for 20 000 paths) but consume ``numpy``'s legacy global RNG in
exactly the reference's order, so ``np.random.seed(s)`` followed by
``generate_paths()`` reproduces the reference's arrays bit for bit
(``tests/test_data_layer.py`` pins this against golden vectors).

Deliberate difference: the reference's tail loop (``stock_model.py:139``) calls
``next_cond_exp(y, delta_t_)`` without ``current_t`` and raises ``TypeError``
whenever the last observation time is < T.  Here the current time is passed.
"""
import copy

import numpy as np


class StockModel:
    """Base class: shared hyper-parameters and the conditional-expectation walk
    (reference ``stock_model.py:15-158``)."""

    def __init__(self, drift, volatility, S0, nb_paths, nb_steps, maturity,
                 sine_coeff=None, **kwargs):
        self.drift = drift
        self.volatility = volatility
        self.S0 = S0
        self.nb_paths = nb_paths
        self.nb_steps = nb_steps
        self.maturity = maturity
        self.dimensions = np.size(S0)
        if sine_coeff is None:
            self.periodic_coeff = lambda t: 1
        else:
            self.periodic_coeff = lambda t: (1 + np.sin(sine_coeff * t))

    # -- to be provided by the concrete model --------------------------------
    def generate_paths(self, **options):
        raise ValueError("not implemented yet")

    def next_cond_exp(self, y, delta_t, current_t):
        raise ValueError("not implemented yet")

    # -- helpers -----------------------------------------------------------------
    def _init_paths(self, start_X):
        paths = np.empty((self.nb_paths, self.dimensions, self.nb_steps + 1))
        if start_X is not None:
            paths[:, :, 0] = start_X
        else:
            paths[:, :, 0] = self.S0
        return paths

    def _walk(self, y, current_time, target, delta_t, path_t, path_y):
        """Advance the conditional expectation from ``current_time`` to
        ``target`` with the model's Euler-grid bookkeeping (same clock rules as
        ``NJODE.forward``: full steps while more than one ``delta_t`` away, then
        one partial step)."""
        while current_time < target - 1e-10 * delta_t:
            if current_time < target - delta_t:
                step = delta_t
            else:
                step = target - current_time
            y = self.next_cond_exp(y, step, current_time)
            current_time = current_time + step
            if path_t is not None:
                path_t.append(current_time)
                path_y.append(y)
        return y, current_time

    def compute_cond_exp(self, times, time_ptr, X, obs_idx, delta_t, T, start_X,
                         n_obs_ot, return_path=True, get_loss=False,
                         weight=0.5, start_time=None, **kwargs):
        """True conditional expectation along the batch's observation schedule
        (reference ``stock_model.py:50-151``).  Returns ``loss`` or
        ``(loss, path_t, path_y)`` with ``path_y`` of shape [n_t, B, d]."""
        y = start_X
        batch_size = start_X.shape[0]
        current_time = start_time if start_time else 0.0
        loss = 0
        path_t = path_y = None
        if return_path:
            path_t, path_y = ([], []) if start_time else ([0.], [y])

        for i, obs_time in enumerate(times):
            if obs_time > T + 1e-10:
                break
            if obs_time <= current_time:
                continue
            y, current_time = self._walk(y, current_time, obs_time, delta_t,
                                         path_t, path_y)
            lo, hi = time_ptr[i], time_ptr[i + 1]
            X_obs = X[lo:hi]
            i_obs = obs_idx[lo:hi]
            Y_bj = y
            y = copy.copy(y)
            y[i_obs] = X_obs
            if get_loss:
                loss = loss + compute_loss(
                    X_obs=X_obs, Y_obs=y[i_obs], Y_obs_bj=Y_bj[i_obs],
                    n_obs_ot=n_obs_ot[i_obs], batch_size=batch_size,
                    weight=weight)
            if return_path:
                path_t.append(obs_time)
                path_y.append(y)

        y, current_time = self._walk(y, current_time, T, delta_t, path_t, path_y)

        if return_path:
            return loss, np.array(path_t), np.array(path_y)
        return loss

    def get_optimal_loss(self, times, time_ptr, X, obs_idx, delta_t, T, start_X,
                         n_obs_ot, weight=0.5):
        """Loss of the true conditional expectation = the floor for the model
        (reference ``stock_model.py:153-158``)."""
        return self.compute_cond_exp(
            times, time_ptr, X, obs_idx, delta_t, T, start_X, n_obs_ot,
            return_path=False, get_loss=True, weight=weight)


class BlackScholes(StockModel):
    """dS = mu S dt + sigma S dW, Euler-Maruyama (reference
    ``stock_model.py:339-375``)."""

    def __init__(self, drift, volatility, nb_paths, nb_steps, S0, maturity,
                 sine_coeff=None, **kwargs):
        super().__init__(drift=drift, volatility=volatility, nb_paths=nb_paths,
                         nb_steps=nb_steps, S0=S0, maturity=maturity,
                         sine_coeff=sine_coeff)

    def next_cond_exp(self, y, delta_t, current_t):
        return y * np.exp(self.drift * self.periodic_coeff(current_t) * delta_t)

    def generate_paths(self, start_X=None):
        dt = self.maturity / self.nb_steps
        paths = self._init_paths(start_X)
        # one draw per (path, step, dim), path-major: the reference's RNG order
        z = np.random.normal(0, 1, (self.nb_paths, self.nb_steps,
                                    self.dimensions))
        sq = np.sqrt(dt)
        for k in range(1, self.nb_steps + 1):
            prev = paths[:, :, k - 1]
            dW = z[:, k - 1, :] * sq
            mu = self.drift * self.periodic_coeff((k - 1) * dt) * prev
            sig = self.volatility * prev
            paths[:, :, k] = prev + mu * dt + sig * dW
        return paths, dt


class OrnsteinUhlenbeck(StockModel):
    """dX = -k (X - m) dt + sigma dW (reference ``stock_model.py:378-418``)."""

    def __init__(self, volatility, nb_paths, nb_steps, S0, mean, speed,
                 maturity, sine_coeff=None, **kwargs):
        super().__init__(volatility=volatility, nb_paths=nb_paths, drift=None,
                         nb_steps=nb_steps, S0=S0, maturity=maturity,
                         sine_coeff=sine_coeff)
        self.mean = mean
        self.speed = speed

    def next_cond_exp(self, y, delta_t, current_t):
        decay = np.exp(-self.speed * self.periodic_coeff(current_t) * delta_t)
        return y * decay + self.mean * (1 - decay)

    def generate_paths(self, start_X=None):
        dt = self.maturity / self.nb_steps
        paths = self._init_paths(start_X)
        z = np.random.normal(0, 1, (self.nb_paths, self.nb_steps,
                                    self.dimensions))
        sq = np.sqrt(dt)
        for k in range(1, self.nb_steps + 1):
            prev = paths[:, :, k - 1]
            dW = z[:, k - 1, :] * sq
            mu = -self.speed * self.periodic_coeff((k - 1) * dt) * (
                prev - self.mean)
            paths[:, :, k] = prev + mu * dt + self.volatility * dW
        return paths, dt


class Heston(StockModel):
    """Heston stochastic-volatility model (reference ``stock_model.py:161-221``).
    Note the reference's scheme: the spot diffusion uses the *updated* variance
    ``v_k`` and the spot drift is evaluated at ``(k-1) dt``."""

    def __init__(self, drift, volatility, mean, speed, correlation, nb_paths,
                 nb_steps, S0, maturity, sine_coeff=None, **kwargs):
        super().__init__(drift=drift, volatility=volatility, nb_paths=nb_paths,
                         nb_steps=nb_steps, S0=S0, maturity=maturity,
                         sine_coeff=sine_coeff)
        self.mean = mean
        self.speed = speed
        self.correlation = correlation

    def next_cond_exp(self, y, delta_t, current_t):
        return y * np.exp(self.drift * self.periodic_coeff(current_t) * delta_t)

    def generate_paths(self, start_X=None):
        dt = self.maturity / self.nb_steps
        spot = self._init_paths(start_X)
        var = np.empty_like(spot)
        var[:, :, 0] = self.mean
        # two draws per (path, step): [.., 0, :] drives the spot, [.., 1, :] is
        # mixed in for the variance
        z = np.random.normal(0, 1, (self.nb_paths, self.nb_steps, 2,
                                    self.dimensions))
        sq = np.sqrt(dt)
        rho = self.correlation
        for k in range(1, self.nb_steps + 1):
            z1 = z[:, k - 1, 0, :]
            z2 = z[:, k - 1, 1, :]
            dW = z1 * sq
            dZ = (rho * z1 + np.sqrt(1 - rho ** 2) * z2) * sq
            v_prev = var[:, :, k - 1]
            s_prev = spot[:, :, k - 1]
            var[:, :, k] = (v_prev + (-self.speed * (v_prev - self.mean)) * dt
                            + (self.volatility * np.sqrt(v_prev)) * dZ)
            spot[:, :, k] = (
                s_prev
                + (self.drift * self.periodic_coeff((k - 1) * dt) * s_prev) * dt
                + (np.sqrt(var[:, :, k]) * s_prev) * dW)
        return spot, dt


def compute_loss(X_obs, Y_obs, Y_obs_bj, n_obs_ot, batch_size, eps=1e-10,
                 weight=0.5):
    """Paper loss in numpy (reference ``stock_model.py:471-481``)."""
    after = np.sqrt(np.sum((X_obs - Y_obs) ** 2, axis=1) + eps)
    before = np.sqrt(np.sum((Y_obs_bj - Y_obs) ** 2, axis=1) + eps)
    inner = (2 * weight * after + 2 * (1 - weight) * before) ** 2
    return np.sum(inner / n_obs_ot) / batch_size


STOCK_MODELS = {
    "BlackScholes": BlackScholes,
    "Heston": Heston,
    "OrnsteinUhlenbeck": OrnsteinUhlenbeck,
    "sine_BlackScholes": BlackScholes,
    "sine_Heston": Heston,
    "sine_OrnsteinUhlenbeck": OrnsteinUhlenbeck,
}
