"""NumPy restatement of the reference's env wrappers, vectorised over env instances.
TEST INFRASTRUCTURE ONLY.  Pinned by tests/test_oracle_wrappers.py against tests/golden/wrappers_replay.npz
(outputs of the unmodified reference wrappers, oracle/make_golden_wrappers.py).
Reference: /root/reference/madrl_environments/__init__.py (file:line in the comments)."""
import numpy as np


class StdOracle(object):
    def __init__(self, obs_shape, rew_shape, scale_reward=1., enable_obsnorm=False, enable_rewnorm=False, obs_alpha=0.001,
                 rew_alpha=0.001, eps=1e-8):
        self.c = dict(scale=scale_reward, on=enable_obsnorm, rn=enable_rewnorm, oa=obs_alpha, ra=rew_alpha, eps=eps)
        self.om, self.ov = np.zeros(obs_shape), np.ones(obs_shape)   # :229-230
        self.rm, self.rv = np.zeros(rew_shape), np.ones(rew_shape)   # :231-232

    def obs(self, o):
        if not self.c["on"]:
            return o
        o = np.asarray(o, np.float64)
        a = self.c["oa"]
        self.om = (1 - a) * self.om + a * o                          # :245-246
        self.ov = (1 - a) * self.ov + a * np.square(o - self.om)     # :247-249
        return (o - self.om) / (np.sqrt(self.ov) + self.c["eps"])    # :262-263

    def rew(self, r):
        r = np.asarray(r, np.float64)
        if self.c["rn"]:
            a = self.c["ra"]
            self.rm = (1 - a) * self.rm + a * r                      # :253-254
            self.rv = (1 - a) * self.rv + a * np.square(r - self.rm) # :255-257
            r = r / (np.sqrt(self.rv) + self.c["eps"])               # :268-271
        return self.c["scale"] * r                                   # :290


class BufOracle(object):
    def __init__(self, obs_shape, k):
        self.k = k
        self.buf = np.zeros(tuple(obs_shape) + (k,))                 # :150

    def reset(self, o, mask=None):
        o = np.asarray(o, np.float64)
        if mask is None:
            self.buf[...] = o[..., None]                             # :190-192
        else:
            m = np.asarray(mask).astype(bool)
            self.buf[m] = o[m][..., None]
        return self.buf.copy()

    def step(self, o, reset_mask=None):
        o = np.asarray(o, np.float64)
        new = np.concatenate([self.buf[..., 1:], o[..., None]], axis=-1)   # :179-181
        if reset_mask is not None:
            m = np.asarray(reset_mask).astype(bool)
            new[m] = o[m][..., None]
        self.buf = new
        return self.buf.copy()


class DiagOracle(object):
    def __init__(self, n_envs, n_agents, discount=0.99, max_traj_len=500):
        self.g, self.mtl = discount, max_traj_len
        self.ep_rew = np.zeros((n_envs, n_agents)); self.ep_len = np.zeros(n_envs, np.int64)
        self.disc = np.zeros(n_envs); self.pw = np.ones(n_envs)

    def reset(self):
        self.ep_rew[:] = 0; self.ep_len[:] = 0; self.disc[:] = 0     # :328-333

    def step(self, rew, done):
        rew = np.asarray(rew, np.float64)
        self.pw[self.ep_len == 0] = 1.0
        self.ep_rew += rew                                           # :350
        self.disc += rew.mean(axis=1) * self.pw                      # :360-361, _discount_sum :392-393
        self.pw *= self.g
        self.ep_len += 1                                             # :351
        fin = np.asarray(done).astype(bool) | (self.ep_len >= self.mtl)   # :354
        out = dict(finished=fin.copy(), reward=self.ep_rew.copy(), disc=self.disc.copy(), length=self.ep_len.copy())
        self.ep_rew[fin] = 0; self.disc[fin] = 0; self.ep_len[fin] = 0    # :365-367
        return out
