"""GPU tests (-m gpu): the wrapper epilogue kernels (madrl_wrap_*) through the Python mirrors,
against the reference's golden outputs and against the NumPy oracle on a live batched env."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden", "wrappers_replay.npz")


class ReplayEnv(object):
    """plays the golden obs / reward sequence back as a 1-env batched env"""

    def __init__(self, g, n_copies=3):
        from madrl_amd.pursuit import PursuitAgent
        self.g, self.t, self.n_envs, self.device = g, -1, n_copies, torch.device(DEV)
        self.P, self.D = g["obs"].shape[1:]
        self.agents = [PursuitAgent((self.D,)) for _ in range(self.P)]
        self.reward_mech, self.auto_reset = "local", False

    def _o(self):
        return torch.as_tensor(np.repeat(self.g["obs"][self.t][None], self.n_envs, 0), device=DEV)

    def reset(self):
        self.t += 1
        assert self.g["op"][self.t] == 0
        return self._o()

    def step(self, a):
        self.t += 1
        assert self.g["op"][self.t] == 1
        rew = torch.as_tensor(np.repeat(self.g["rew"][self.t][None], self.n_envs, 0).astype(np.float32), device=DEV)
        done = torch.full((self.n_envs,), bool(self.g["done"][self.t]), device=DEV)
        return self._o(), rew, done, {}


def test_wrappers_match_reference_golden():
    from madrl_amd.wrappers import StandardizedEnv, ObservationBuffer, DiagnosticsWrapper
    g = np.load(G)
    T = len(g["op"])
    std = StandardizedEnv(ReplayEnv(g), scale_reward=float(g["std_cfg_scale_reward"]), enable_obsnorm=True, enable_rewnorm=True,
                          obs_alpha=float(g["std_cfg_obs_alpha"]), rew_alpha=float(g["std_cfg_rew_alpha"]), eps=float(g["std_cfg_eps"]))
    buf = ObservationBuffer(ReplayEnv(g), int(g["buf_k"]))
    diag = DiagnosticsWrapper(ReplayEnv(g), discount=float(g["diag_discount"]), max_traj_len=int(g["diag_max_traj_len"]))
    k = 0
    for t in range(T):
        if g["op"][t] == 0:
            so = std.reset(); bo = buf.reset(); diag.reset()
        else:
            so, sr, _, _ = std.step(None)
            bo, _, _, _ = buf.step(None)
            _, _, _, log = diag.step(None)
            assert np.abs(sr.cpu().numpy() - g["std_rew"][t][None]).max() < 1e-5 * max(1.0, np.abs(g["std_rew"][t]).max())
            if bool(log["finished"][0]):
                assert t == g["diag_at"][k]
                assert np.abs(log["global/episode_reward_agents"][1].cpu().numpy() - g["diag_reward"][k]).max() < 1e-6
                assert abs(float(log["global/episode_disc_return"][2]) - g["diag_disc"][k]) < 1e-6
                assert int(log["global/episode_length"][0]) == g["diag_len"][k]
                assert abs(float(log["global/episode_avg_reward"][0]) - g["diag_avg"][k]) < 1e-6
                k += 1
        assert np.abs(so.cpu().numpy() - g["std_obs"][t][None]).max() < 1e-5, "standardised obs, op %d" % t
        assert np.array_equal(bo.cpu().numpy(), np.repeat(g["buf_obs"][t][None], 3, 0)), "frame stack, op %d" % t
    assert k == len(g["diag_at"])
    assert buf.agents[0].observation_space.shape == (g["obs"].shape[2], int(g["buf_k"]))


def test_wrappers_on_live_auto_reset_env_match_numpy_oracle():
    from madrl_amd.maps import rectangle_map
    from madrl_amd.pursuit import BatchedPursuitEvade
    from madrl_amd.wrappers import StandardizedEnv, ObservationBuffer, DiagnosticsWrapper
    from oracle import wrappers_oracle as wo
    N, P, H = 512, 8, 15
    mk = lambda: BatchedPursuitEvade([rectangle_map(16, 16)], n_envs=N, device=DEV, seed=3, max_steps=H, auto_reset=True,
                                     n_pursuers=P, n_evaders=30, obs_range=7, reward_mech="local", catchr=0.1)
    env = DiagnosticsWrapper(StandardizedEnv(ObservationBuffer(mk(), 3), scale_reward=2.0, enable_obsnorm=True, enable_rewnorm=True,
                                             obs_alpha=0.05, rew_alpha=0.05), discount=0.9, max_traj_len=10)
    raw = mk()
    D = raw.obs_dim
    bo, so, do = wo.BufOracle((N, P, D), 3), wo.StdOracle((N, P, D, 3), (N, P), scale_reward=2.0, enable_obsnorm=True,
                                                         enable_rewnorm=True, obs_alpha=0.05, rew_alpha=0.05), wo.DiagOracle(N, P, 0.9, 10)
    obs = env.reset()
    ref = so.obs(bo.reset(raw.reset().cpu().numpy()))
    assert np.abs(obs.cpu().numpy() - ref).max() < 1e-5
    g = torch.Generator(device="cpu").manual_seed(0)
    nfin = 0
    for t in range(40):
        act = torch.randint(0, 5, (N, P), generator=g, dtype=torch.int32).to(DEV)
        obs, rew, done, log = env.step(act)
        ro, rr, rd, rinfo = raw.step(act)
        bits = rinfo["done_bits"].cpu().numpy()
        ref_o = so.obs(bo.step(ro.cpu().numpy(), reset_mask=bits != 0))
        ref_r = so.rew(rr.cpu().numpy())
        out = do.step(ref_r, bits != 0)   # Diagnostics sits above StandardizedEnv: it sees the scaled reward
        assert np.abs(obs.cpu().numpy() - ref_o).max() < 1e-5, t
        assert np.abs(rew.cpu().numpy() - ref_r).max() < 1e-4, t
        fin = log["finished"].cpu().numpy()
        assert np.array_equal(fin, out["finished"])
        if fin.any():
            nfin += int(fin.sum())
            assert np.abs(log["global/episode_reward_agents"].cpu().numpy()[fin] - out["reward"][fin]).max() < 1e-3
            assert np.abs(log["global/episode_disc_return"].cpu().numpy()[fin] - out["disc"][fin]).max() < 1e-3
            assert np.array_equal(log["global/episode_length"].cpu().numpy()[fin], out["length"][fin])
    assert nfin > N


def test_fused_standardized_waterworld_matches_unfused_and_oracle():
    """StandardizedEnv fused into the Waterworld step / reset kernels (madrl_waterworld_set_standardize): same values as the
    stand-alone epilogue kernels on an identical env, and both within 1e-5 of the NumPy restatement of the reference wrapper."""
    from madrl_amd.waterworld import BatchedMAWaterWorld
    from madrl_amd.wrappers import StandardizedEnv
    from oracle import wrappers_oracle as wo
    N, H = 256, 12
    mk = lambda: BatchedMAWaterWorld(5, 10, n_envs=N, device=DEV, seed=9, max_steps=H, auto_reset=True)
    cfg = dict(scale_reward=0.7, enable_obsnorm=True, enable_rewnorm=True, obs_alpha=0.05, rew_alpha=0.05)
    fused, plain, raw = StandardizedEnv(mk(), **cfg), StandardizedEnv(mk(), fused=False, **cfg), mk()
    assert fused._fused and not plain._fused
    D = raw.obs_dim
    so = wo.StdOracle((N, 5, D), (N, 5), **cfg)
    of, op = fused.reset(), plain.reset()
    ref = so.obs(raw.reset().cpu().numpy())
    assert torch.equal(of, op) and np.abs(of.cpu().numpy() - ref).max() < 1e-5
    g = torch.Generator(device="cpu").manual_seed(2)
    for t in range(30):
        a = (torch.rand((N, 5, 2), generator=g) * 2 - 1).to(DEV)
        of, rf, df, _ = fused.step(a)
        op, rp, dp, _ = plain.step(a)
        ro, rr, rd, _ = raw.step(a)
        assert torch.equal(of, op) and torch.equal(rf, rp) and torch.equal(df, dp), "step %d: fused != epilogue kernels" % t
        assert np.abs(of.cpu().numpy() - so.obs(ro.cpu().numpy())).max() < 1e-5, t
        want = so.rew(rr.cpu().numpy())
        assert np.abs(rf.cpu().numpy() - want).max() < 1e-5 * max(1.0, np.abs(want).max()), t
    assert torch.equal(fused._obs_mean, plain._obs_mean) and torch.equal(fused._rew_var, plain._rew_var)


def test_vectorised_observation_buffer_equals_the_scalar_kernel():
    """The k = 4 observation buffer moves one 16-byte word per element whenever the buffer is 16-byte aligned (643 -> 551 us at
    65 536 x 8 x 148), the scalar kernel otherwise: identical results (a buffer shifted by one float forces the scalar path)."""
    import torch
    from madrl_amd import _lib
    L, P = _lib.lib(), _lib.ptr
    dev = torch.device(DEV)
    st = _lib.current_stream(dev)
    N, E = 257, 30
    g = torch.Generator(device="cpu").manual_seed(5)
    obs = torch.randn((5, N * E), generator=g).to(dev)
    rm = (torch.rand(N, generator=g) < 0.3).to(torch.uint8).to(dev)
    am = (torch.rand(N, generator=g) < 0.9).to(torch.uint8).to(dev)
    def runbuf(off):
        buf = torch.zeros(N * E * 4 + 4, device=dev)
        for t in range(5):
            _lib.check(L.madrl_wrap_obsbuffer(P(obs[t]), P(buf[off:]), N * E, E, 4, P(rm) if t in (0, 3) else None, P(am) if t == 3 else None, st))
        return buf[off:off + N * E * 4].clone()
    a, b = runbuf(0), runbuf(1)
    assert torch.equal(a, b) and float(a.abs().sum()) > 0
