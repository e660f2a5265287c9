"""Env API mirror of madrl_environments/__init__.py:9-119 (Agent, AbstractMAEnv)."""


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

    @property
    def unwrapped(self):
        return self

    def __str__(self):
        return "<{} instance>".format(type(self).__name__)
