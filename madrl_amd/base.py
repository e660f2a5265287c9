"""Env API mirror of madrl_environments/__init__.py:9-119 (Agent, AbstractMAEnv)."""
import numpy as np


def stack_dict_list(dict_list):
    """rltools.util.stack_dict_list as used by AbstractMAEnv.animate (:106): list of info dicts -> dict of arrays."""
    ret = {}
    if not dict_list:
        return ret
    for k in dict_list[0].keys():
        ret[k] = np.asarray([d[k] for d in dict_list])
    return ret


class Agent(object):
    """madrl_environments/__init__.py:9-24"""

    @property
    def observation_space(self):
        raise NotImplementedError()

    @property
    def action_space(self):
        raise NotImplementedError()

    def __str__(self):
        return "<{} instance>".format(type(self).__name__)


class AbstractMAEnv(object):
    """madrl_environments/__init__.py:27-119 (render/animate are out of scope: matplotlib)."""

    def setup(self):
        pass

    def seed(self, seed=None):
        return []

    @property
    def agents(self):
        raise NotImplementedError()

    @property
    def reward_mech(self):
        raise NotImplementedError()

    def reset(self):
        raise NotImplementedError()

    def step(self, actions):
        raise NotImplementedError()

    @property
    def is_terminal(self):
        raise NotImplementedError()

    def set_param_values(self, lut):
        # madrl_environments/__init__.py:64-67
        for k, v in lut.items():
            setattr(self, k, v)
        self.setup()

    def render(self, *args, **kwargs):
        """:69-70.  Rendering (matplotlib / pyglet on the host) is out of scope; animate() runs without frames."""
        return None

    def animate(self, act_fn, nsteps, **kwargs):
        """:72-107 without the video encoder: one policy function per agent (or one for all), reset, then up to nsteps
        steps or until done; returns (summed rewards per agent, stacked info dicts) like the reference."""
        if not isinstance(act_fn, list):
            act_fn = [act_fn for _ in range(len(self.agents))]
        assert len(act_fn) == len(self.agents)
        obs = self.reset()
        rew = np.zeros((len(self.agents)))
        traj_info_list = []
        for step in range(nsteps):
            a = list(map(lambda afn, o: afn(o), act_fn, obs))
            obs, r, done, info = self.step(a)
            rew += r
            if info:
                traj_info_list.append(info)
            if done:
                break
        return rew, stack_dict_list(traj_info_list)

    @property
    def unwrapped(self):
        return self

    def __str__(self):
        return "<{} instance>".format(type(self).__name__)


class SingleEnvDelegate(object):
    """Mixin of the N == 1 drop-in classes: attributes the reference's callers read off the env object
    (n_pursuers, catchr, map_matrix, ...) come from the one-env batched engine held in `_env`.

    pickle / copy.deepcopy probe `__setstate__`, `__reduce_ex__`, ... on an instance whose __dict__ is still empty
    (the reference samplers hand pickled env copies to their workers, rltools.util.EzPickle): those lookups must end
    in AttributeError, not in a KeyError from the delegation."""

    def __getattr__(self, name):
        env = self.__dict__.get("_env")
        if env is None or (name.startswith("__") and name.endswith("__")):
            raise AttributeError(name)
        return getattr(env, name)

    def __getstate__(self):
        return dict(self.__dict__)  # `_env`, the batched engine, pickles by constructor arguments (EzPickle-style)

    def __setstate__(self, state):
        self.__dict__.update(state)
        self._after_unpickle()

    def _after_unpickle(self):
        """EzPickle re-runs the reference constructor; classes whose constructor ends in reset() redo it here."""
