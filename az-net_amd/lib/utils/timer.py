"""Stopwatch with a running mean, the interface test_proposals / test_net_shared use
(reference: lib/utils/timer.py:10-32): tic(), toc(average=True), and the attributes
total_time, calls, diff, average_time."""
from time import perf_counter


class Timer(object):
    __slots__ = ("total_time", "calls", "diff", "average_time", "_t0")

    def __init__(self):
        self.reset()

    def reset(self):
        self.total_time = self.diff = self.average_time = 0.0
        self.calls = 0
        self._t0 = None

    def tic(self):
        self._t0 = perf_counter()

    def toc(self, average=True):
        if self._t0 is None:
            raise RuntimeError("Timer.toc() without tic()")
        self.diff = perf_counter() - self._t0
        self.calls += 1
        self.total_time += self.diff
        self.average_time = self.total_time / self.calls
        return self.average_time if average else self.diff
