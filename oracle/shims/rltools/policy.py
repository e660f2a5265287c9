"""Stand-in for rltools.policy (un-vendored submodule): the base class the reference heuristics derive from."""


class Policy(object):
    def __init__(self, observation_space, action_space):
        self.observation_space, self.action_space = observation_space, action_space
