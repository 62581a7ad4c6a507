"""
Host-side time grid of one NJ-ODE pass.

``NJODE.forward`` in the reference keeps a float64 Python clock
(``models.py:430-439, 497-505``): Euler steps of ``delta_t`` while the clock is
more than one step away from the next observation time, then one partial step
onto it; the loop guard is ``current_time < obs_time - 1e-10 * delta_t``.  The
kernels consume that clock as fp32 arrays, rounded exactly where ATen rounds
(python-float operands of fp32 tensor ops are cast to fp32):

    step_dt[k]  = fp32(delta_t_k)                 (h + delta_t_ * f(...))
    step_t[k]   = fp32(current_time before k)     (current_time - tau)
    time_f32[i] = fp32(times[i])                  (tau[i_obs] = obs_time)
    k_jump[i]   = number of Euler steps completed when jump i happens

``path_t`` is the reference's ``path_t`` output (float64, first entry 0).
"""
import collections

import numpy as np
import torch


class Schedule:
    __slots__ = ('step_dt', 'step_t', 'k_jump', 'time_f32', 'path_t', 'n_steps', 'n_times',
                 'row_of_jump')

    def __init__(self, times, delta_t, T, until_T):
        times = np.asarray(times, dtype=np.float64)
        dts, ts, k_jump, path_t, row_of_jump = [], [], [], [0.0], []
        now = 0.0

        def walk(now, target):
            guard = target - 1e-10 * delta_t
            while now < guard:
                step = delta_t if now < target - delta_t else target - now
                dts.append(step)
                ts.append(now)
                now = now + step
                path_t.append(now)
            return now

        for obs_time in times:
            now = walk(now, obs_time)
            k_jump.append(len(dts))
            row_of_jump.append(len(path_t))
            path_t.append(obs_time)
        if until_T:
            now = walk(now, T)
        self.step_dt = np.asarray(dts, dtype=np.float64).astype(np.float32)
        self.step_t = np.asarray(ts, dtype=np.float64).astype(np.float32)
        self.k_jump = np.asarray(k_jump, dtype=np.int32)
        self.time_f32 = times.astype(np.float32)
        self.path_t = np.asarray(path_t, dtype=np.float64)
        self.row_of_jump = np.asarray(row_of_jump, dtype=np.int64)
        self.n_steps = len(dts)
        self.n_times = len(times)

    @property
    def n_rows(self):
        return 1 + self.n_steps + self.n_times

    def has_tail(self):
        """Euler steps after the last jump (until_T) -- or no jump at all."""
        if self.n_times == 0:
            return self.n_steps > 0
        return int(self.k_jump[-1]) < self.n_steps

    def packed_nbytes(self):
        return 4 * (2 * self.n_steps + 3 * self.n_times + 1)

    def pack_into(self, buf, time_ptr):
        """Write [step_dt | step_t | k_jump | time_f32 | time_ptr] into the int32
        numpy view ``buf`` (the layout the library copies with one memcpy)."""
        K, nt = self.n_steps, self.n_times
        f = buf.view(np.float32)
        f[0:K] = self.step_dt
        f[K:2 * K] = self.step_t
        buf[2 * K:2 * K + nt] = self.k_jump
        f[2 * K + nt:2 * K + 2 * nt] = self.time_f32
        buf[2 * K + 2 * nt:2 * K + 3 * nt + 1] = time_ptr
        return K, nt


class ScheduleCache:
    """LRU of schedules keyed by (times, delta_t, T, until_T)."""

    def __init__(self, capacity=64):
        self.capacity = capacity
        self._d = collections.OrderedDict()

    def get(self, times, delta_t, T, until_T):
        times = np.ascontiguousarray(times, dtype=np.float64)
        key = (times.tobytes(), float(delta_t), float(T), bool(until_T))
        s = self._d.get(key)
        if s is None:
            s = Schedule(times, delta_t, T, until_T)
            self._d[key] = s
            if len(self._d) > self.capacity:
                self._d.popitem(last=False)
        else:
            self._d.move_to_end(key)
        return s


class PinnedRing:
    """Round-robin pinned host buffers for the asynchronous schedule upload.  A
    slot is reused only after the event recorded behind its last upload has
    completed."""

    def __init__(self, slots=16):
        self.slots = [None] * slots
        self.events = [None] * slots
        self.next = 0
        self.held = set()

    def acquire(self, nbytes):
        i = self.next
        while i in self.held and len(self.held) < len(self.slots):
            i = (i + 1) % len(self.slots)
        self.next = (i + 1) % len(self.slots)
        ev = self.events[i]
        if ev is not None:
            ev.synchronize()
        t = self.slots[i]
        n_int = (nbytes + 3) // 4 + 16
        if t is None or t.numel() < n_int:
            t = torch.empty(max(n_int, 1024), dtype=torch.int32).pin_memory()
            self.slots[i] = t
        return i, t

    def hold(self, i):
        """Slot ``i`` is read by work that is not enqueued yet (a deferred plan): ``acquire``
        passes it over until ``release_with`` / ``release_after`` frees it."""
        self.held.add(i)

    def release_with(self, i, j):
        """Slot ``i`` is free when slot ``j`` is: both were read by the same enqueued call (they
        share ``j``'s event; recording it again for either only moves the guard later)."""
        self.held.discard(i)
        self.events[i] = self.events[j]

    def release_after(self, i, stream):
        self.held.discard(i)
        ev = self.events[i]
        if ev is None:
            ev = torch.cuda.Event()
            self.events[i] = ev
        ev.record(stream)
