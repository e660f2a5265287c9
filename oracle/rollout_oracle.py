"""NumPy restatement of the discounted-return / GAE post-processing the reference's external samplers apply
(rllab `special.discount_cumsum` = scipy.signal.lfilter([1], [1, -discount], x[::-1])[::-1], used for
`returns` with `discount` and for `advantages` on the TD residuals with `discount * gae_lambda`;
runners/rurllab.py:298-305).  TEST INFRASTRUCTURE ONLY.  The libraries are not vendored in the reference
("parity anchored on the published formula"): tests pin this file against scipy.signal.lfilter per episode."""
import numpy as np


def gae(rew, done, values, gamma, lam):
    """rew [T,N,A], done [T,N] (nonzero = boundary after step t), values [T+1,N,A] or None -> returns, adv"""
    T = rew.shape[0]
    rew = rew.astype(np.float64)
    cut = (np.asarray(done) != 0)[..., None]
    ret = np.zeros(rew.shape)
    adv = np.zeros(rew.shape) if values is not None else None
    nxt = values[T].astype(np.float64) if values is not None else np.zeros(rew.shape[1:])
    a = np.zeros(rew.shape[1:])
    for t in range(T - 1, -1, -1):
        nxt = rew[t] + np.where(cut[t], 0.0, gamma * nxt)
        ret[t] = nxt
        if values is not None:
            v1 = values[t + 1].astype(np.float64)
            delta = rew[t] + np.where(cut[t], 0.0, gamma * v1) - values[t]
            a = delta + np.where(cut[t], 0.0, gamma * lam * a)
            adv[t] = a
    return ret, adv
