"""Minimal Box / Discrete spaces with the attributes the reference's callers read
(rllabwrapper/__init__.py:16-27, runners/rurltools.py:29-38): .shape, .low, .high, .n.
`gym` itself is not a dependency of this package."""
import numpy as np


class Box(object):
    def __init__(self, low, high, shape=None):
        if shape is None:
            self.low = np.asarray(low, dtype=np.float64)
            self.high = np.asarray(high, dtype=np.float64)
            assert self.low.shape == self.high.shape
        else:
            self.low = np.zeros(shape) + low
            self.high = np.zeros(shape) + high

    @property
    def shape(self):
        return self.low.shape

    def sample(self):
        lo = np.maximum(self.low, -1e3)
        hi = np.minimum(self.high, 1e3)
        return np.random.uniform(lo, hi, size=self.low.shape)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool((x >= self.low).all() and (x <= self.high).all())

    def __repr__(self):
        return "Box%s" % (self.shape,)


class Discrete(object):
    def __init__(self, n):
        self.n = int(n)

    @property
    def shape(self):
        return ()

    def sample(self):
        return int(np.random.randint(self.n))

    def contains(self, x):
        return 0 <= int(x) < self.n

    def __repr__(self):
        return "Discrete(%d)" % self.n
