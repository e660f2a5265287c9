"""Minimal Box / Discrete spaces with the attributes the reference's callers read
(rllabwrapper/__init__.py:16-27, runners/rurltools.py:29-38): .shape, .low, .high, .n.
`gym` itself is not a dependency of this package; when it IS importable the two classes derive from gym's, so
that `isinstance(agent.observation_space, spaces.Box)` in the reference's own wrappers
(madrl_environments/__init__.py:156, :225) holds for the drop-in envs."""
import numpy as np

try:  # pragma: no cover - depends on the deployment
    from gym import spaces as _gym_spaces
    _BoxBase, _DiscreteBase = _gym_spaces.Box, _gym_spaces.Discrete
except Exception:
    _BoxBase = _DiscreteBase = object


class Box(_BoxBase):
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


class Discrete(_DiscreteBase):
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
