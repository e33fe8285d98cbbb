"""drecpy_amd/_spinpool.SpinWorker: fit()'s host-prefetch worker whose hand-overs are polled through drx_spin_until (host code only)."""
import threading
import time

import pytest


def test_jobs_run_in_order_on_one_worker_thread_and_results_come_back():
    from drecpy_amd._spinpool import SpinWorker
    w = SpinWorker()
    seen = []

    def job(i):
        seen.append((i, threading.current_thread().name))
        return i * i
    try:
        for i in range(3000):
            t = w.submit(job, i)
            assert t.result() == i * i
        assert [i for i, _ in seen] == list(range(3000)) and {n for _, n in seen} == {'drx-prefetch'}
    finally:
        w.close()


def test_the_worker_sleeps_when_idle_and_wakes_for_the_next_job():
    from drecpy_amd._spinpool import SpinWorker
    w = SpinWorker()
    try:
        assert w.submit(lambda: 1).result() == 1
        time.sleep(0.15)                                   # far beyond POLL_US: the worker is asleep on its event now
        assert int(w._c[2]) == 1
        t0 = time.perf_counter()
        assert w.submit(lambda: 2).result() == 2
        assert time.perf_counter() - t0 < 0.04            # woken by the client, not by the 50 ms timeout
    finally:
        w.close()


def test_exceptions_reach_the_client_and_the_worker_survives_them():
    from drecpy_amd._spinpool import SpinWorker
    w = SpinWorker()
    try:
        def boom():
            raise KeyError('x')
        with pytest.raises(KeyError):
            w.submit(boom).result()
        assert w.submit(lambda a, b=0: a + b, 2, b=3).result() == 5
        t = w.submit(lambda: 7)
        with pytest.raises(RuntimeError):
            w.submit(lambda: 8)                            # one job at a time
        assert t.result() == 7
        with pytest.raises(RuntimeError):
            t.result()                                     # a result is taken once
    finally:
        w.close()
    assert not w._th.is_alive()
    with pytest.raises(RuntimeError):
        w.submit(lambda: 1)


def test_slow_jobs_are_waited_for_beyond_the_polling_window():
    from drecpy_amd._spinpool import SpinWorker
    w = SpinWorker()
    try:
        def slow():
            time.sleep(0.05)
            return 'done'
        assert w.submit(slow).result() == 'done'
    finally:
        w.close()
