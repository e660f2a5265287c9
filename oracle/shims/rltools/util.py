import numpy as np


class EzPickle(object):
    """Pickle by constructor arguments (what rltools.util.EzPickle does)."""

    def __init__(self, *args, **kwargs):
        self._ezpickle_args = args
        self._ezpickle_kwargs = kwargs

    def __getstate__(self):
        return {"_ezpickle_args": self._ezpickle_args, "_ezpickle_kwargs": self._ezpickle_kwargs}

    def __setstate__(self, d):
        out = type(self)(*d["_ezpickle_args"], **d["_ezpickle_kwargs"])
        self.__dict__.update(out.__dict__)


def stack_dict_list(dict_list):
    ret = {}
    if not dict_list:
        return ret
    for k in dict_list[0].keys():
        ret[k] = np.asarray([d[k] for d in dict_list])
    return ret
