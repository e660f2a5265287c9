import numpy as np


def np_random(seed=None):
    if seed is None:
        seed = int(np.random.randint(2**31 - 1))
    return np.random.RandomState(seed), seed
