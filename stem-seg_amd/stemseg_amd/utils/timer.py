"""Named wall-clock accumulators with the interface ``inference/main.py`` and ``inference_model.py`` use from the reference's
``stemseg/utils/timer.py``: ``Timer.log_duration(*names)`` / ``Timer.exclude_duration(*names)`` decorators,
``Timer.get_duration(name)``, ``Timer.get_durations_sum()``, ``Timer.print_durations()``, plus tic / toc / context manager.
"""
import functools
import time


class Timer(object):
    _TIMERS = {}

    def __init__(self, name):
        self._name = name
        self._started = None
        self._total = 0.0

    running = property(lambda self: self._started is not None)
    paused = property(lambda self: self._started is None)
    total_duration = property(lambda self: self._total)

    def tic(self):
        assert self._started is None, "tic() has already been called for timer '{}'".format(self._name)
        self._started = time.time()

    def toc(self):
        assert self._started is not None, "tic() has not been called for timer '{}'".format(self._name)
        self._total += time.time() - self._started
        self._started = None

    __enter__ = tic

    def __exit__(self, *exc):
        self.toc()

    @classmethod
    def create(cls, name):
        assert name not in cls._TIMERS, "Timer with name '{}' already exists".format(name)
        cls._TIMERS[name] = cls(name)
        return cls._TIMERS[name]

    @classmethod
    def get(cls, name):
        return cls._TIMERS[name] if name in cls._TIMERS else cls.create(name)

    @classmethod
    def get_duration(cls, name):
        assert name in cls._TIMERS, "No timer named '{}' exists".format(name)
        return cls._TIMERS[name].total_duration

    @classmethod
    def get_durations_sum(cls):
        return sum(t.total_duration for t in cls._TIMERS.values())

    @classmethod
    def print_durations(cls):
        for name, t in cls._TIMERS.items():
            print(" - {}: {:03f} sec".format(name, t.total_duration))
        print(" - TOTAL: {:03f} sec".format(cls.get_durations_sum()))

    @classmethod
    def _switching(cls, names, start):
        """Decorator factory: while the wrapped call runs, the named timers are started (``start``) or paused (not ``start``) if
        they were not already in that state, and put back afterwards."""
        def deco(fn):
            @functools.wraps(fn)
            def wrapped(*args, **kwargs):
                if start:
                    flipped = [t for t in map(cls.get, names) if t.paused]
                else:
                    flipped = [cls._TIMERS[n] for n in names if n in cls._TIMERS and cls._TIMERS[n].running]
                for t in flipped:
                    t.tic() if start else t.toc()
                try:
                    return fn(*args, **kwargs)
                finally:
                    for t in flipped:
                        t.toc() if start else t.tic()
            return wrapped
        return deco

    @classmethod
    def log_duration(cls, *timer_names):
        return cls._switching(timer_names, True)

    @classmethod
    def exclude_duration(cls, *timer_names):
        return cls._switching(timer_names, False)
