"""CPU soak (no GPU): the product's MultiWalker source (CPU build) against the independent Box2D-ordered oracle, free-running without any
re-synchronisation, 96 000 env-steps per walker count (--walkers 5 8 10: the larger capacity classes) with stretches of zero actions; every body state, joint state, fat AABB, sleep time, done flag must stay
identical and the sticky overflow bits zero.   python scripts/mw_soak.py   (~15 s)"""
import sys, numpy as np, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import multiwalker as mwo, multiwalker_ref as mwr
import argparse
ap = argparse.ArgumentParser(); ap.add_argument('--walkers', type=int, nargs='*', default=[3, 4, 2]); ap.add_argument('--steps', type=int, default=1500); ap.add_argument('--seed', type=int, default=0, help='added to the per-walker-count seeds: another soak')
ap.add_argument('--no-terminate', action='store_true', help='terminate_on_fall off: fallen walkers stay (more sleeping bodies, package dropped)'); ap.add_argument('--rev', type=int, default=0, help='b2CollidePolygons revision'); ap.add_argument('--hold', type=int, default=1, help='actions drawn anew every HOLD steps (held in between: other gaits)'); ap.add_argument('--scale', type=float, default=1.0, help='action amplitude'); ap.add_argument('--global-reward', action='store_true'); ap.add_argument('--one-hot', action='store_true'); ap.add_argument('--noise', type=float, default=0.0, help='position_noise = angle_noise (the reference\'s default is 1e-3)'); ap.add_argument('--contacts', type=int, default=0, help='every CONTACTS steps also compare every env\'s whole contact list: pairs in list order, touching, feature ids, warm-start impulses'); ap.add_argument('--descending', action='store_true', help='the CPU build runs an env\'s lanes in descending order (the result must not depend on it)'); ap.add_argument('--gait', type=float, default=0.0, help='fraction of the walkers that follow the reference\'s hand-written gait (heuristics/multi_walker.py) instead of random actions')
args = ap.parse_args()
for W in args.walkers:
    seed = {3: 101, 4: 102, 2: 103}.get(W, 100 + W) + args.seed
    N, T = 64, args.steps
    ref = mwr.MultiWalkerRef(n_walkers=W, n_envs=N, seed=seed, position_noise=args.noise, angle_noise=args.noise, poly=True, terminate_on_fall=not args.no_terminate, polygon_revision=args.rev, reward_mech='global' if args.global_reward else 'local', one_hot=args.one_hot)
    core = mwo.MultiWalkerOracle(n_walkers=W, n_envs=N, seed=seed, position_noise=args.noise, angle_noise=args.noise, lanes_descending=(W == 4) != args.descending, terminate_on_fall=not args.no_terminate, polygon_revision=args.rev, reward_mech='global' if args.global_reward else 'local', one_hot=args.one_hot)
    ref.reset(); core.reset()
    rng = np.random.RandomState(seed); gait_rng = np.random.RandomState(seed + 1); pick = None
    t0 = time.time(); nd = 0; n_ov = 0
    for t in range(T):
        if t % args.hold == 0: a0 = (args.scale * rng.uniform(-1, 1, (N, W, 4))).astype(np.float32)
        a = a0.copy()
        if args.gait > 0 and t > 0:
            from oracle import heuristics_oracle as ho
            g = ho.multiwalker_actions(np.asarray(ro, np.float64).reshape(N * W, -1)).reshape(N, W, 4).astype(np.float32)
            pick = gait_rng.rand(N, W) < args.gait if t % 50 == 1 else pick
            a = np.where(pick[:, :, None], g, a)
        if (t // 40) % 5 == 4: a[:] = 0
        ro, rr, rd = ref.step(a); co, cr, cd = core.step(a)
        ov = core.overflow() != 0   # the product's sticky capacity flag (a contact did not fit the step's manifold pool / its cache): that env is no
        if ov.any():                # longer comparable -- counted, and restarted on both sides
            n_ov += int(ov.sum()); rd = rd | ov; cd = cd | ov
            keep = ~ov
            assert np.array_equal(ref.bodies()[keep], core.bodies()[0][keep]), (W, t)
            ref.reset(mask=rd.astype(np.uint8)); core.reset(mask=rd.astype(np.uint8)); nd += int(rd.sum())
            continue
        assert np.array_equal(ref.bodies(), core.bodies()[0]) and np.array_equal(rd, cd), (W, t)
        assert np.array_equal(ref.joints(), core.joints()) and np.array_equal(ref.aux(), core.aux()), (W, t)
        # the env layer on top (float64 on the reference's side, float32 in the product): observations incl. lidar, rewards
        ro64, rr64 = np.asarray(ro, np.float64).reshape(co.shape), np.asarray(rr, np.float64).reshape(cr.shape)
        assert (np.abs(ro64 - co) <= 1e-6 * np.maximum(1.0, np.abs(ro64))).all() and (np.abs(rr64 - cr) <= 1e-6 * np.maximum(1.0, np.abs(rr64))).all(), (W, t, 'obs / rewards')
        if args.contacts and t % args.contacts == 0:
            for e in range(N):
                (ri, rf), (ci, cf) = ref.contacts(e, 1024), core.contacts(e, 1024)
                assert len(ri) == len(ci) and np.array_equal(ri[:, :7], ci[:, :7]) and np.array_equal(rf, cf), (W, t, e, 'contact lists')
        nd += int(rd.sum())
        if rd.any(): ref.reset(mask=rd); core.reset(mask=rd)
    print("W=%d: %d free-running env-steps identical, %d episodes, %d continuous-pass events, %d capacity overflows (env restarted), %.0f s" % (W, N * T, nd, ref.stats()["toi_events"], n_ov, time.time() - t0), flush=True)
