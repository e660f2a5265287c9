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
        """Contract of AbstractMAEnv.animate (madrl_environments/__init__.py:72-107) for headless use: `act_fn` is one policy
        callable per controllable agent, or a single callable shared by all of them; the env is reset and stepped with
        `[policy_i(observation_i)]` until it reports done or `nsteps` steps have run.  Returns (per-agent reward totals,
        the non-empty info dicts of the episode stacked key by key).  Frame capture and video encoding are out of scope
        (`vid=` / `fps=` are accepted and ignored)."""
        n_agents = len(self.agents)
        policies = list(act_fn) if isinstance(act_fn, (list, tuple)) else [act_fn] * n_agents
        if len(policies) != n_agents:
            raise AssertionError("animate: %d policy functions for %d agents" % (len(policies), n_agents))
        totals = np.zeros(n_agents)
        infos = []
        observations = self.reset()
        t = 0
        finished = False
        while t < nsteps and not finished:
            actions = [policy(o) for policy, o in zip(policies, observations)]
            observations, rewards, finished, info = self.step(actions)
            totals += rewards
            if info:
                infos.append(info)
            t += 1
        return totals, stack_dict_list(infos)

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

    def set_param_values(self, lut):
        """madrl_environments/__init__.py:64-67 (setattr + setup()) on the engine that holds the parameters -- not on this shell, where the
        mixin's own AbstractMAEnv.set_param_values would put them"""
        self._env.set_param_values(lut)
        self._after_set_params()

    def _after_set_params(self):
        """classes whose reference setup() ends in reset() (multi_walker.py:303) redo it here"""

    def __getstate__(self):
        return dict(self.__dict__)  # `_env`, the batched engine, pickles by constructor arguments (EzPickle-style)

    def __setstate__(self, state):
        self.__dict__.update(state)
        self._after_unpickle()

    def _after_unpickle(self):
        """EzPickle re-runs the reference constructor; classes whose constructor ends in reset() redo it here."""
