"""Known-answer scenes for the MultiWalker dynamics, through every restatement of the physics (tests/mw_scenes.py).

PARITY WITH BOX2D STAYS UNPINNED -- none of the expected values below is Box2D output.  They are analytic (momentum in free fall,
Coulomb friction on an incline, motor speeds, the sleep timer) or follow from Box2D's published constants (b2_angularSlop,
b2_polygonRadius, b2_linearSlop, b2_timeToSleep), and they exercise what the one published trace ("Hello Box2D": a box falling on a
box) cannot: b2RevoluteJoint motors and limits, b2CollideEdgeAndPolygon with sliding friction, sleeping, the continuous pass against
an edge, and a closed-loop gait.  CPU: the independent plain-C world and the product's solver source; `-m gpu`: the HIP kernels."""
import numpy as np
import pytest

import mw_scenes as S

CPU = ["ref", "core"]
GPU = [pytest.param("hip", marks=pytest.mark.gpu)]
ALL = CPU + GPU


@pytest.mark.parametrize("kind", ALL)
def test_motors_run_at_their_speed_and_stop_at_their_limits(kind):
    """multi_walker.py:145-179 (hip limits [-0.8, 1.1], knee [-1.6, -0.1], motors on), :194-203 (motorSpeed = SPEED * sign(a),
    maxMotorTorque = 80 |a|).  Walkers in free fall with full actions: every joint turns at exactly its motor speed (4 / 6 rad/s) until
    its limit, overshoots it by less than one step of travel, is brought back to within b2_angularSlop and stays; the walker's centre
    of mass follows semi-implicit Euler under gravity to 2e-3 and its angular momentum stays where it was."""
    r = S.scene_motor_limits(S.Backend(kind, 4, 3, seed=2), steps=40)
    S.check_motor_limits(r)
    a = r["angle"][-1]
    # where Box2D's limit handling leaves a joint that its motor keeps pressing against a limit: beyond the limit by at most
    # b2_angularSlop (the position correction C = clamp(angle - limit -/+ slop, ...) stops there; seen: anywhere in that band)
    sl = S.ANGULAR_SLOP + 1e-3
    assert np.all((a[..., 0] > S.HIP_LIM[1]) & (a[..., 0] < S.HIP_LIM[1] + sl))
    assert np.all((a[..., 1] < S.KNEE_LIM[0]) & (a[..., 1] > S.KNEE_LIM[0] - sl))
    assert np.all((a[..., 2] < S.HIP_LIM[0]) & (a[..., 2] > S.HIP_LIM[0] - sl))
    # the knee created ABOVE its upper limit (straight legs, angle 0 > -0.1, :136-163) is pulled to within slop of it
    assert np.all((a[..., 3] > S.KNEE_LIM[1]) & (a[..., 3] < S.KNEE_LIM[1] + sl))


@pytest.mark.parametrize("kind", ALL)
def test_box_on_a_tilted_chain_of_edges_sticks_below_and_slides_above_the_friction_angle(kind):
    """:617-620 (edge fixtures, friction 2.5), :504-511 (package friction 0.5): b2MixFriction = sqrt(0.5 * 2.5) = 1.118, friction angle
    48.2 degrees.  At 35 degrees the package must not move; at 55 degrees it must accelerate at g (sin t - mu cos t) = 1.779 m/s^2."""
    mu = S.MU_PACKAGE_TERRAIN
    assert abs(np.degrees(np.arctan(mu)) - 48.19) < 0.01
    r = S.scene_slope(S.Backend(kind, 2, 1), np.radians(35.0), steps=60)
    assert np.abs(r["v_t"][2:]).max() < 1e-3, "below the friction angle the package stays where it is"
    assert np.abs(r["ang"]).max() < 2e-3
    th = np.radians(55.0)
    r = S.scene_slope(S.Backend(kind, 2, 1), th, steps=60)
    a = S.G * (np.sin(th) - mu * np.cos(th))
    n = np.arange(60)
    v = r["v_t"]
    # the first step settles the contact; from then on every step adds a * dt
    assert np.abs(np.diff(v[2:], axis=0) - a / S.FPS).max() < 2e-3 * a / S.FPS + 2e-5, "acceleration along the slope: g (sin t - mu cos t)"
    assert abs(v[59, 0] - v[1, 0] - a * 58 / S.FPS) < 1e-3
    # it slides ON the surface: 2 * polygonRadius - linearSlop above the edges, not rotating
    assert np.abs(r["dist"][10:] - (2 * S.POLY_RADIUS - S.LINEAR_SLOP)).max() < 5e-4 and np.abs(r["ang"]).max() < 2e-3
    assert (r["flags"][:, 0] == 1).all()   # ContactDetector: the package touched the ground -> game over (:58-62)


@pytest.mark.parametrize("kind", ALL)
def test_island_goes_to_sleep_half_a_second_after_it_came_to_rest(kind):
    """b2Island::Solve, allowSleep: the package dropped on flat ground is put to sleep in the step that brings its sleep time to
    b2_timeToSleep = 0.5 s = 25 steps below the tolerances (0.01 m/s, 2 degrees/s), with velocities exactly zero, and rests
    2 * polygonRadius - linearSlop above the edge.  (The walkers never sleep: SetMotorSpeed wakes them every step, :196-203.)"""
    S.check_sleep(S.scene_sleep(S.Backend(kind, 3, 3), steps=100))


@pytest.mark.parametrize("kind", ALL)
def test_thin_box_faster_than_its_thickness_per_step_does_not_tunnel(kind):
    """b2World::SolveTOI: the package (0.33 m thick) arrives at 40 m/s = 0.8 m per step.  Without the continuous pass it is below the
    terrain after two steps and keeps falling; with it (Box2D's default, what the reference runs) it stops on the surface."""
    y = S.scene_fast_drop(S.Backend(kind, 2, 3, continuous=True))
    assert (y > 0).all(), "never below the surface"
    assert np.all((y[3:] > 2 * S.POLY_RADIUS - S.LINEAR_SLOP - 1e-3) & (y[3:] < 2 * S.POLY_RADIUS + 2e-3)), y[:, 0]
    assert np.abs(np.diff(y[4:], axis=0)).max() < 1e-4, "and it rests there"
    y = S.scene_fast_drop(S.Backend(kind, 2, 3, continuous=False))
    assert (y[1:] < -0.5).all() and (np.diff(y, axis=0) < -0.7).all(), "b2World.continuousPhysics = False: the same box tunnels"


def test_gait_state_machine_matches_the_reference_policy_on_its_first_call():
    """the stateful gait is the reference's expressions: with fresh state it must return what heuristics/multi_walker.py returns
    (the reference's copy is in that state on EVERY call)"""
    from oracle import heuristics_oracle as ho
    obs = np.random.RandomState(0).uniform(-1, 1, (500, 32))
    obs[:, 8] = obs[:, 8] > 0; obs[:, 13] = obs[:, 13] > 0
    assert np.abs(S.StatefulGait(500)(obs) - ho.multiwalker_actions(obs)).max() < 1e-12


@pytest.mark.parametrize("kind", CPU)
def test_gait_walks_the_package_forward(kind):
    """Closed loop, 96 envs x up to 500 steps, flat terrain, the env's random initial pushes.  The gait of gym's BipedalWalker demo
    walks here: most envs carry the package many metres, a good share all the way to the end of the terrain (:420).  The reference's
    own copy of that policy (state machine restarted on every call: it stands on one leg) topples after ~84 steps in both
    restatements -- recorded, not asserted as right or wrong.  Nothing here says Box2D would give these numbers."""
    r = S.scene_gait(S.Backend(kind, 96, 3, seed=5), policy="stateful")
    assert r["reached_end"].mean() > 0.3 and r["survived"].mean() > 300 and np.median(r["travel"]) > 8.0, (r["reached_end"].mean(), r["survived"].mean())
    assert (r["outcome"] == 2).mean() < 0.1, "the package is rarely dropped"
    q = S.scene_gait(S.Backend(kind, 16, 3, seed=5), policy="reference")
    assert q["reached_end"].sum() == 0 and 60 < q["survived"].mean() < 120


def test_gait_rollout_is_identical_in_both_restatements():
    a = S.scene_gait(S.Backend("ref", 48, 3, seed=9), policy="stateful")
    b = S.scene_gait(S.Backend("core", 48, 3, seed=9), policy="stateful")
    assert np.array_equal(a["survived"], b["survived"]) and np.array_equal(a["outcome"], b["outcome"]) and np.array_equal(a["travel"], b["travel"])


@pytest.mark.gpu
def test_gait_on_the_kernels_1024_envs_equals_the_cpu_build():
    """1 024 envs in closed loop through the C ABI: outcome by outcome what the CPU build of the same source gives, and the same rates"""
    h = S.scene_gait(S.Backend("hip", 1024, 3, seed=5), policy="stateful")
    c = S.scene_gait(S.Backend("core", 1024, 3, seed=5), policy="stateful")
    assert np.array_equal(h["survived"], c["survived"]) and np.array_equal(h["outcome"], c["outcome"])
    assert h["reached_end"].mean() > 0.3 and h["survived"].mean() > 300
    print("gait, 1024 envs: reached the end %.3f, walker fell %.3f, package dropped %.3f, mean episode %.0f steps"
          % (h["reached_end"].mean(), (h["outcome"] == 3).mean(), (h["outcome"] == 2).mean(), h["survived"].mean()))
