import numpy as np


class Box(object):
    def __init__(self, low, high, shape=None):
        if shape is None:
            low = np.asarray(low)
            high = np.asarray(high)
            assert low.shape == high.shape
            self.low, self.high = low, high
        else:
            self.low = np.zeros(shape) + low
            self.high = np.zeros(shape) + high

    @property
    def shape(self):
        return self.low.shape

    def sample(self):
        return np.random.uniform(low=np.maximum(self.low, -1e3), high=np.minimum(self.high, 1e3),
                                 size=self.low.shape)

    def contains(self, x):
        return x.shape == self.shape and (x >= self.low).all() and (x <= self.high).all()


class Discrete(object):
    def __init__(self, n):
        self.n = n

    @property
    def shape(self):
        return ()

    def sample(self):
        return np.random.randint(self.n)

    def contains(self, x):
        return 0 <= int(x) < self.n
