"""Debug aid: group kernel vs generic kernel, step by step, first differences printed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from madrl_amd.pursuit import BatchedPursuitEvade
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
DEV = "cuda:0"
g = np.load("tests/golden/pursuit_pool16_sample_maps.npz")
maps = list(g["maps"])
kw = dict(n_pursuers=20, n_evaders=50, obs_range=5, n_catch=2, surround=True, flatten=True, reward_mech="local", sample_maps=True)
N = int(os.environ.get("DBG_N", "512"))
envs = {k: BatchedPursuitEvade(maps, n_envs=N, device=DEV, seed=2024, env_id_base=1000, max_steps=25, auto_reset=True, kernel=k, **kw) for k in ("generic", "auto")}
print({k: e.kernel_kind for k, e in envs.items()})
o = {k: e.reset().clone() for k, e in envs.items()}
print("reset obs equal:", torch.equal(o["generic"], o["auto"]))
rng = np.random.RandomState(5)
for t in range(40):
    act = torch.as_tensor(rng.randint(5, size=(N, 20)), device=DEV)
    pre = {k: {a: b.cpu().numpy() for a, b in e.get_state().items()} for k, e in envs.items()}
    out = {k: e.step(act) for k, e in envs.items()}
    st = {k: {a: b.cpu().numpy() for a, b in e.get_state().items()} for k, e in envs.items()}
    ra, rb = out["generic"][1].cpu().numpy(), out["auto"][1].cpu().numpy()
    oa, ob = out["generic"][0].cpu().numpy().reshape(N, 20, -1), out["auto"][0].cpu().numpy().reshape(N, 20, -1)
    bad = np.argwhere(ra != rb)
    print("step", t, "rew diffs", len(bad), "obs diffs", int((oa != ob).sum()), "state diffs", {a: int((st["generic"][a] != st["auto"][a]).sum()) for a in st["generic"]})
    for n, p in bad[:4]:
        print("  env", n, "pursuer", p, "generic", ra[n, p], "group", rb[n, p], "pos pre", pre["generic"]["pos_p"][n, p], "post", st["generic"]["pos_p"][n, p],
              "removed", int(out["generic"][3]["removed"][n]), int(out["auto"][3]["removed"][n]), "map", pre["generic"]["map_id"][n])
        ev = pre["generic"]["pos_e"][n][pre["generic"]["gone"][n] == 0]
        d = np.abs(ev - pre["generic"]["pos_p"][n, p]).sum(1)
        print("   evaders within 1 (pre):", ev[d <= 1].tolist(), " evader slots:", np.nonzero((pre["generic"]["gone"][n] == 0))[0][d <= 1].tolist())
    if len(bad):
        n = bad[0][0]
        for k in ("generic", "auto"):
            print(k, "gone post", np.nonzero(st[k]["gone"][n])[0].tolist(), "term_e post", np.nonzero(st[k]["term_e"][n])[0].tolist())
        print("pre gone", np.nonzero(pre["auto"]["gone"][n])[0].tolist(), "pre term_e", np.nonzero(pre["auto"]["term_e"][n])[0].tolist(), "pre term_p", np.nonzero(pre["auto"]["term_p"][n])[0].tolist())
        dif = np.nonzero((st["generic"]["pos_e"][n] != st["auto"]["pos_e"][n]).any(1))[0]
        print("evaders whose pos differ", dif.tolist())
        for i in dif[:10]:
            print("   slot", i, "pre", pre["auto"]["pos_e"][n, i], "generic", st["generic"]["pos_e"][n, i], "group", st["auto"]["pos_e"][n, i])
        print("envs with any diff:", np.nonzero((st["generic"]["gone"] != st["auto"]["gone"]).any(1) | (ra != rb).any(1))[0].tolist())
        break
