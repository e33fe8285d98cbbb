"""One worker thread for fit()'s host prefetch whose hand-overs are POLLED, not slept on.

RecommenderABC.fit() draws batch s + 1 on a worker thread while batch s trains (`_host_prefetch`; the reference draws inline:
recommender_abc.py:186-188).  With concurrent.futures both hand-overs — job to the worker, result back — are futex sleeps, and at a batch
every 90 - 250 us the wake-ups cost as much as the batch: DMF.fit() at B = 256 ran at 0.09 or at 0.23 ms per step from one window to the
next (r06, profiles/r06_host_handover.log).  Here the worker and the client poll two counters through `drx_spin_until` — a C loop, called
through ctypes, so WITHOUT the interpreter lock — for a few hundred microseconds before they fall back to sleeping: the same protocol as the
native draw-ahead workers of CDAE's reference mode (csrc/drx_host.cpp DrxDrawAhead).  One job in flight at a time."""
import threading
import time

import numpy as np


class _Ticket:
    __slots__ = ('pool', 'n')

    def __init__(self, pool, n):
        self.pool, self.n = pool, n

    def result(self):
        return self.pool._result(self.n)


class SpinWorker:
    POLL_US = 400

    def __init__(self, spin=None):
        # spin(ptr, at_least, poll_us) -> 1 / 0; default: the library's drx_spin_until
        if spin is None:
            from . import _lib
            spin = _lib.lib().drx_spin_until
        self._spin = spin
        self._c = np.zeros(4, np.int64)                   # submitted, done, worker sleeping, stop
        base = self._c.ctypes.data
        self._p_sub, self._p_done = base, base + 8
        self._go = threading.Event()
        self._job = self._res = None
        self._n = self._taken = 0
        self._th = threading.Thread(target=self._run, name='drx-prefetch', daemon=True)
        self._th.start()

    def _run(self):
        c, nxt = self._c, 1
        while True:
            while c[0] < nxt:
                if c[3]:
                    return
                if self._spin(self._p_sub, nxt, self.POLL_US) > 0:         # (1: there; 0: polled long enough; < 0: a bad pointer — treated as 0)
                    break
                c[2] = 1                                  # nothing for a while: sleep — the client sets `go` when it sees this flag
                if c[0] < nxt and not c[3]:
                    self._go.wait(0.05)
                self._go.clear()
                c[2] = 0
            fn, a, k = self._job
            try:
                self._res = (True, fn(*a, **k))
            except BaseException as e:                    # noqa: BLE001  (handed to the client)
                self._res = (False, e)
            c[1] = nxt
            nxt += 1

    def submit(self, fn, *a, **k):
        if self._n != self._taken:
            raise RuntimeError('SpinWorker runs one job at a time: take the result of the previous one first')
        if self._c[3] or not self._th.is_alive():
            raise RuntimeError('SpinWorker is closed')
        self._job = (fn, a, k)
        self._n += 1
        self._c[0] = self._n
        if self._c[2]:
            self._go.set()
        return _Ticket(self, self._n)

    def _result(self, n):
        if n != self._n or n == self._taken:
            raise RuntimeError('result of a job that is not the one in flight')
        while self._spin(self._p_done, n, self.POLL_US) <= 0:
            if not self._th.is_alive():
                raise RuntimeError('the prefetch worker died')
            time.sleep(50e-6)
        self._taken = n
        ok, v = self._res
        self._res = self._job = None
        if ok:
            return v
        raise v

    def close(self):
        self._c[3] = 1
        self._go.set()
        if self._th.is_alive() and threading.current_thread() is not self._th:
            self._th.join(timeout=1.0)

    shutdown = close

    def __del__(self):
        try:
            self.close()
        except Exception:                                 # noqa: BLE001
            pass
