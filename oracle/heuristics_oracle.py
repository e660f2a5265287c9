"""NumPy restatement of the reference's hand-written policies, vectorised over rows.  TEST INFRASTRUCTURE ONLY.
Pinned by tests/test_oracle_heuristics.py against tests/golden/heuristics.npz (outputs of the unmodified reference
classes, oracle/make_golden_heuristics.py).  Reference: /root/reference/heuristics/{pursuit,waterworld,multi_walker}.py."""
import math

import numpy as np

LEFT, RIGHT, UP, DOWN, STAY, RANDOM = 0, 1, 2, 3, 4, -1


def pursuit_actions(win):
    """win [B, R, R, C>=3] -> int32 [B]; RANDOM (-1) where the reference calls action_space.sample()  (pursuit.py:18-54)"""
    B, R = win.shape[0], win.shape[1]
    x = y = R // 2                                           # :23, Python 2 integer division
    out = np.full(B, RANDOM, np.int32)
    for b in range(B):
        xev, yev = np.nonzero(win[b, ..., 2])                # :27, row-major order
        if len(xev) == 0:
            continue
        d = np.sqrt((xev - x) ** 2 + (yev - y) ** 2)         # :28
        k = int(np.argmin(d))                                # :29, first minimum
        xc, yc = xev[k], yev[k]
        if xc == x and yc == y:
            out[b] = STAY
            continue
        ang = math.atan2(yc - y, xc - x)                     # :33-34
        ang = (ang + np.pi) % (2 * np.pi) - np.pi
        if -np.pi / 4 <= ang < np.pi / 4:
            out[b] = RIGHT
        elif np.pi / 4 <= ang < 3 / 4. * np.pi:
            out[b] = UP
        elif ang >= 3 / 4. * np.pi or ang < -3 / 4. * np.pi:
            out[b] = LEFT
        elif -3 / 4. * np.pi <= ang < -np.pi / 4:
            out[b] = DOWN
    return out


def waterworld_actions(obs):
    """obs [B, D] -> float64 [B, 2], every row normalised on its own (the reference is called with B = 1; its
    np.linalg.norm over the whole batch, waterworld.py:46, is then the row norm)"""
    obs = np.asarray(obs, np.float64)
    K = obs.shape[1] // 7                                    # :26
    ang = np.linspace(0., 2. * np.pi, K + 1)[:-1]
    vec = np.c_[np.cos(ang), np.sin(ang)]
    s = lambda a: np.sum(obs[:, a * K:(a + 1) * K][..., None] * vec[None], axis=1)
    oa, ev, po, pu = -s(0), s(1), -s(3), s(5) / 2            # :31-41
    ev[obs[:, 7 * K] > 0] *= 1.5                             # :43-44
    po[obs[:, 7 * K + 1] > 0] *= 1.5
    act = oa + ev + po + pu
    n = np.sqrt((act ** 2).sum(axis=1, keepdims=True))
    return np.where(n > 0, act / np.where(n > 0, n, 1.0), 0.0)


def multiwalker_actions(obs):
    """obs [B, >=14] -> float64 [B, 4]  (multi_walker.py:16-86; the gait state machine restarts on every call, :23-25)"""
    obs = np.asarray(obs, np.float64)
    out = np.zeros((obs.shape[0], 4))
    SPEED, SKA = 0.29, 0.1
    for i, s in enumerate(obs):
        state, moving, sup = 1, 0, 1
        ska = SKA
        hip_t, knee_t = [None, None], [None, None]
        hip, knee = [0.0, 0.0], [0.0, 0.0]
        mb, sb = 4 + 5 * moving, 4 + 5 * sup
        hip_t[moving] = 1.1; knee_t[moving] = -0.6
        ska += 0.03
        if s[2] > SPEED:
            ska += 0.03
        ska = min(ska, SKA)
        knee_t[sup] = ska
        if s[sb + 0] < 0.10:
            state = 2
        if state == 2:
            hip_t[moving] = 0.1; knee_t[moving] = SKA; knee_t[sup] = ska
            if s[mb + 4]:
                state = 3
                ska = min(s[mb + 2], SKA)
        if state == 3:
            knee_t[moving] = ska; knee_t[sup] = 1.0
        if hip_t[0]: hip[0] = 0.9 * (hip_t[0] - s[4]) - 0.25 * s[5]        # `if target:` -- a 0.0 target counts as unset
        if hip_t[1]: hip[1] = 0.9 * (hip_t[1] - s[9]) - 0.25 * s[10]
        if knee_t[0]: knee[0] = 4.0 * (knee_t[0] - s[6]) - 0.25 * s[7]
        if knee_t[1]: knee[1] = 4.0 * (knee_t[1] - s[11]) - 0.25 * s[12]
        hip[0] -= 0.9 * (0 - s[0]) - 1.5 * s[1]; hip[1] -= 0.9 * (0 - s[0]) - 1.5 * s[1]
        knee[0] -= 15.0 * s[3]; knee[1] -= 15.0 * s[3]
        out[i] = np.clip(0.5 * np.array([hip[0], knee[0], hip[1], knee[1]]), -1.0, 1.0)
    return out
