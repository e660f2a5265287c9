"""Hand-written policies of the reference (heuristics/pursuit.py, heuristics/waterworld.py, heuristics/multi_walker.py)
on the device: `policy(obs)` maps the batched observation tensor [N, A, ...] of the matching env straight to its action
tensor, so an NN-free rollout (env.step -> policy -> env.step) never leaves the GPU.  `sample_actions(obs)` keeps the
reference's method name and its (actions, action-dist-placeholder) return."""
import math

import numpy as np
import torch

from . import _lib

LEFT, RIGHT, UP, DOWN, STAY = 0, 1, 2, 3, 4


def pursuit_decision_table(R):
    """action for "nearest evader at window cell (xc, yc)", the reference's own float64 expression (pursuit.py:23-52);
    255 where it falls through to action_space.sample()"""
    x = y = R // 2  # :23, Python 2 integer division
    t = np.full(R * R, 255, np.uint8)
    for xc in range(R):
        for yc in range(R):
            if xc == x and yc == y:
                a = STAY
            else:
                ang = math.atan2(yc - y, xc - x)
                ang = (ang + np.pi) % (2 * np.pi) - np.pi
                if -np.pi / 4 <= ang < np.pi / 4:
                    a = RIGHT
                elif np.pi / 4 <= ang < 3 / 4. * np.pi:
                    a = UP
                elif ang >= 3 / 4. * np.pi or ang < -3 / 4. * np.pi:
                    a = LEFT
                elif -3 / 4. * np.pi <= ang < -np.pi / 4:
                    a = DOWN
                else:
                    a = 255
            t[xc * R + yc] = a
    return t


class _DevicePolicy(object):
    def __call__(self, obs):
        return self.sample_actions(obs)[0]

    def get_state(self):
        return []

    def set_state(self, *args):
        pass


class PursuitHeuristicPolicy(_DevicePolicy):
    """obs: flatten rows [N, P, 3*R*R+1] or windows [N, P, R, R, 4] of BatchedPursuitEvade -> int32 [N, P]"""

    def __call__(self, obs, out=None):
        """out: optional contiguous int32 destination of the actions (e.g. a slot of a trajectory tensor): no copy afterwards"""
        return self.sample_actions(obs, out=out)[0]

    def __init__(self, obs_range, flatten=True, seed=0, row_id_base=0):
        self.R, self.flatten, self.seed, self.row_id_base = int(obs_range), bool(flatten), int(seed), int(row_id_base)
        self._tick = None  # [draw counter, workgroup counts ...] on the device: every launch advances the counter itself (a host counter would be frozen into a captured hipGraph)
        self._table_host = pursuit_decision_table(self.R)
        self._table = None
        self._act = None

    def sample_actions(self, obs, deterministic=True, out=None):
        obs = obs.contiguous()
        R = self.R
        n_rows = obs.shape[0] * obs.shape[1]
        if self._table is None or self._table.device != obs.device:
            self._table = torch.as_tensor(self._table_host, device=obs.device)
        if out is not None:
            assert out.dtype == torch.int32 and out.is_contiguous() and out.numel() == n_rows and out.device == obs.device
            act = out
        else:
            if self._act is None or self._act.numel() != n_rows or self._act.device != obs.device:
                self._act = torch.empty(obs.shape[:2], dtype=torch.int32, device=obs.device)
            act = self._act
        if self._tick is None or self._tick.device != obs.device:
            self._tick = torch.zeros(_lib.POLICY_COUNTER_WORDS, dtype=torch.int32, device=obs.device)
        row_stride = obs.numel() // n_rows
        cell_stride, ch_off = (1, 2 * R * R) if self.flatten else (4, 2)
        _lib.check(_lib.lib().madrl_heuristic_pursuit(_lib.ptr(obs), n_rows, R, row_stride, cell_stride, ch_off, _lib.ptr(self._table),
                                                      self.seed, self.row_id_base, 0, _lib.ptr(self._tick), _lib.ptr(act),
                                                      _lib.current_stream(obs.device)))
        return act, None


class WaterworldHeuristicPolicy(_DevicePolicy):
    """obs float32 [N, Np, D] of BatchedMAWaterWorld -> float32 [N, Np, 2]"""

    def __init__(self):
        self._cs = None
        self._act = None

    def sample_actions(self, obs, deterministic=True):
        obs = obs.contiguous()
        D = obs.shape[-1]
        K = D // 7
        n_rows = obs.numel() // D
        if self._cs is None or self._cs.device != obs.device or self._cs.shape[0] != K:
            ang = np.linspace(0., 2. * np.pi, K + 1)[:-1]  # waterworld.py:27-29
            self._cs = torch.as_tensor(np.c_[np.cos(ang), np.sin(ang)].copy(), dtype=torch.float64, device=obs.device)
        if self._act is None or self._act.numel() != 2 * n_rows or self._act.device != obs.device:
            self._act = torch.empty(tuple(obs.shape[:-1]) + (2,), dtype=torch.float32, device=obs.device)
        _lib.check(_lib.lib().madrl_heuristic_waterworld(_lib.ptr(obs), n_rows, D, _lib.ptr(self._cs), _lib.ptr(self._act),
                                                         _lib.current_stream(obs.device)))
        return self._act, None


class MultiWalkerHeuristicPolicy(_DevicePolicy):
    """obs float32 [N, W, 32] of BatchedMultiWalkerEnv -> float32 [N, W, 4]"""

    def __init__(self):
        self._act = None

    def sample_actions(self, obs, deterministic=True):
        obs = obs.contiguous()
        D = obs.shape[-1]
        n_rows = obs.numel() // D
        if self._act is None or self._act.numel() != 4 * n_rows or self._act.device != obs.device:
            self._act = torch.empty(tuple(obs.shape[:-1]) + (4,), dtype=torch.float32, device=obs.device)
        _lib.check(_lib.lib().madrl_heuristic_multiwalker(_lib.ptr(obs), n_rows, D, _lib.ptr(self._act), _lib.current_stream(obs.device)))
        return self._act, None
