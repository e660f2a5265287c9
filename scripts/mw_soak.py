"""CPU soak (no GPU): the product's MultiWalker source (CPU build) against the independent Box2D-ordered oracle, free-running without any
re-synchronisation, 96 000 env-steps per walker count (--walkers 5 8 10: the larger capacity classes) with stretches of zero actions; every body state, joint state, fat AABB, sleep time, done flag must stay
identical and the sticky overflow bits zero.   python scripts/mw_soak.py   (~15 s)"""
import sys, numpy as np, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import multiwalker as mwo, multiwalker_ref as mwr
import argparse
ap = argparse.ArgumentParser(); ap.add_argument('--walkers', type=int, nargs='*', default=[3, 4, 2]); ap.add_argument('--steps', type=int, default=1500); ap.add_argument('--seed', type=int, default=0, help='added to the per-walker-count seeds: another soak')
ap.add_argument('--no-terminate', action='store_true', help='terminate_on_fall off: fallen walkers stay (more sleeping bodies, package dropped)'); ap.add_argument('--rev', type=int, default=0, help='b2CollidePolygons revision')
args = ap.parse_args()
for W in args.walkers:
    seed = {3: 101, 4: 102, 2: 103}.get(W, 100 + W) + args.seed
    N, T = 64, args.steps
    ref = mwr.MultiWalkerRef(n_walkers=W, n_envs=N, seed=seed, position_noise=0, angle_noise=0, poly=True, terminate_on_fall=not args.no_terminate, polygon_revision=args.rev)
    core = mwo.MultiWalkerOracle(n_walkers=W, n_envs=N, seed=seed, position_noise=0.0, angle_noise=0.0, lanes_descending=(W == 4), terminate_on_fall=not args.no_terminate, polygon_revision=args.rev)
    ref.reset(); core.reset()
    rng = np.random.RandomState(seed)
    t0 = time.time(); nd = 0
    for t in range(T):
        a = rng.uniform(-1, 1, (N, W, 4)).astype(np.float32)
        if (t // 40) % 5 == 4: a[:] = 0
        ro, rr, rd = ref.step(a); co, cr, cd = core.step(a)
        assert np.array_equal(ref.bodies(), core.bodies()[0]) and np.array_equal(rd, cd), (W, t)
        assert np.array_equal(ref.joints(), core.joints()) and np.array_equal(ref.aux(), core.aux()), (W, t)
        nd += int(rd.sum())
        if rd.any(): ref.reset(mask=rd); core.reset(mask=rd)
    assert not core.overflow().any()
    print("W=%d: %d free-running env-steps identical, %d episodes, %d continuous-pass events, %.0f s" % (W, N * T, nd, ref.stats()["toi_events"], time.time() - t0), flush=True)
