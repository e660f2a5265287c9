"""Known-answer scenes built through the `Box2D` shim (oracle/shims_box2d): the same API calls a pybox2d script would make, answered by
the independent restatement's World (oracle/multiwalker_ref.c).  CPU only; TEST INFRASTRUCTURE testing test infrastructure.

Why: the MultiWalker dynamics are PARITY UNPINNED -- no Box2D to compare with -- so what can be checked is that the restated world behaves
like the rigid-body mechanics Box2D integrates, in situations the env itself never isolates: a joint to a STATIC body under gravity (the
env's joints hang on free bodies), a free hinge (motor torque 0), a limit holding against gravity, Coulomb friction bringing a sliding body
to rest.  The product's solver equals this World bit for bit on everything the env does (tests/test_multiwalker_cpu.py), so a wrong anchor
transform, inertia, limit or friction clamp here would be one there.  None of the expected values is Box2D output."""
import math
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle", "shims_box2d"))
import Box2D  # noqa: E402  (the shim; asserts below make sure it is)
from Box2D.b2 import edgeShape, fixtureDef, polygonShape, revoluteJointDef  # noqa: E402

assert "shim" in Box2D.__version__
G, DT = 10.0, 1.0 / 50.0
HX, HY = 4.0 / 30.0, 17.0 / 30.0      # the env's upper leg: LEG_W / 2, LEG_H / 2 (multi_walker.py:33)


def _pendulum(theta0, lower=-3.0, upper=3.0, torque=0.0, speed=0.0):
    """a leg-sized box hanging from a static body by a revolute joint at its top end, turned by theta0 about the pivot"""
    w = Box2D.b2World()
    anchor = w.CreateStaticBody(position=(0.0, 10.0), fixtures=fixtureDef(shape=polygonShape(box=(0.1, 0.1)), categoryBits=0x0008, maskBits=0x0008))
    c = (HY * math.sin(theta0), 10.0 - HY * math.cos(theta0))
    leg = w.CreateDynamicBody(position=c, angle=theta0, fixtures=fixtureDef(shape=polygonShape(box=(HX, HY)), density=1.0, categoryBits=0x0002, maskBits=0x0001))
    j = w.CreateJoint(revoluteJointDef(bodyA=anchor, bodyB=leg, localAnchorA=(0, 0), localAnchorB=(0, HY), enableMotor=True, enableLimit=True,
                                       maxMotorTorque=torque, motorSpeed=speed, lowerAngle=lower, upperAngle=upper))
    return w, leg, j


def test_free_hinge_swings_with_the_period_of_a_physical_pendulum():
    theta0 = 0.2
    w, leg, j = _pendulum(theta0)
    m = 1.0 * 4 * HX * HY
    I_pivot = m * (4 * HX * HX + 4 * HY * HY) / 12.0 + m * HY * HY
    T = 2 * math.pi * math.sqrt(I_pivot / (m * G * HY)) * (1 + theta0 ** 2 / 16)        # first amplitude correction
    ang, pivot_err = [], 0.0
    for _ in range(400):
        w.Step(DT, 6 * 30, 2 * 30)
        ang.append(j.angle)
        p, a = leg.position, leg.angle
        top = (p.x - HY * math.sin(a), p.y + HY * math.cos(a))                          # the leg's joint anchor in world coordinates (its origin is its centre)
        pivot_err = max(pivot_err, math.hypot(top[0], top[1] - 10.0))
    ang = np.array(ang)
    down = [i + ang[i] / (ang[i] - ang[i + 1]) for i in range(len(ang) - 1) if ang[i] > 0 >= ang[i + 1]]   # zero crossings, interpolated
    periods = np.diff(down) * DT
    assert len(periods) >= 4 and abs(periods.mean() - T) / T < 0.01, (periods, T)
    peaks = [abs(ang[i]) for i in range(1, len(ang) - 1) if abs(ang[i]) >= abs(ang[i - 1]) and abs(ang[i]) > abs(ang[i + 1])]
    assert 0.9 * theta0 < min(peaks) and max(peaks) < 1.02 * theta0, peaks              # semi-implicit Euler: the amplitude neither grows nor decays fast
    assert pivot_err < 0.005 + 1e-4                                                     # the point constraint holds to b2_linearSlop


def test_limit_holds_against_gravity_within_the_angular_slop():
    """released inside the limits, the box falls onto its upper limit and is held there (b2RevoluteJoint limit state, position correction)"""
    w, leg, j = _pendulum(0.5, lower=-0.2, upper=0.45 + 1e-9)    # starts 0.05 rad beyond the upper limit
    slop = 2.0 / 180.0 * math.pi
    for _ in range(100):
        w.Step(DT, 6 * 30, 2 * 30)
    assert j.angle < 0.45 + slop                                  # pushed back inside the slop and kept from swinging further out ...
    w2, leg2, j2 = _pendulum(0.4, lower=-0.2, upper=0.45)           # a free pendulum would reach -0.4
    lo, after = 0.0, []
    for t in range(200):
        w2.Step(DT, 6 * 30, 2 * 30)
        lo = min(lo, j2.angle)
        after.append(j2.angle)
    # the limit becomes active in the step after the angle has passed it (b2RevoluteJoint::InitVelocityConstraints looks at the angle the
    # step starts with): at ~2 rad/s that is an overshoot of up to one step of travel, 0.04 rad, which the position correction takes back
    assert -0.2 - 0.05 < lo < -0.2 + 1e-3, lo
    assert max(after[100:]) < 0.39                                 # the stop at the limit is inelastic: it never gets back up to where it started


def test_motor_against_gravity_holds_its_speed_until_the_torque_runs_out():
    """maxMotorTorque above m g d sin(theta): the hinge turns at motorSpeed; below it gravity wins (b2RevoluteJoint motor clamp at h * maxMotorTorque)"""
    m = 1.0 * 4 * HX * HY
    need = m * G * HY          # torque that holds the leg horizontal
    w, leg, j = _pendulum(0.0, torque=3.0 * need, speed=1.0)
    for _ in range(40):
        w.Step(DT, 6 * 30, 2 * 30)
    assert abs(j.speed - 1.0) < 1e-3 and 0.7 < j.angle < 0.85     # 40 steps at 1 rad/s
    w, leg, j = _pendulum(1.2, torque=0.3 * need, speed=1.0)
    for _ in range(20):
        w.Step(DT, 6 * 30, 2 * 30)
    assert j.speed < 0.0                                          # falls back although the motor pushes up


def test_sliding_box_stops_where_coulomb_friction_says():
    w = Box2D.b2World()
    for i in range(-2, 40):
        w.CreateStaticBody(fixtures=fixtureDef(shape=edgeShape(vertices=[(i * 14 / 30.0, 0.0), ((i + 1) * 14 / 30.0, 0.0)]), friction=2.5, categoryBits=0x0001))
    hx, hy, mu = 0.5, 0.25, math.sqrt(np.float32(0.5) * np.float32(2.5))
    box = w.CreateDynamicBody(position=(1.0, hy + 0.015), fixtures=fixtureDef(shape=polygonShape(box=(hx, hy)), density=1.0, friction=0.5, categoryBits=0x0004))
    for _ in range(60):                                           # settle
        w.Step(DT, 6 * 30, 2 * 30)
    x0 = box.position.x
    # no API to set a velocity: push for one step instead (F dt / m), then let friction work
    v0 = 6.0
    box.ApplyForceToCenter((box.mass * v0 / DT + mu * box.mass * G, 0.0), True)
    xs, vs = [], []
    for _ in range(80):
        w.Step(DT, 6 * 30, 2 * 30)
        xs.append(box.position.x); vs.append(box.linearVelocity.x)
    v1 = vs[0]
    assert abs(v1 - v0) < 0.05                                    # the push minus one step of friction
    n_stop = next(i for i, v in enumerate(vs) if abs(v) < 1e-4)
    assert abs(n_stop - v1 / (mu * G * DT)) <= 1.5                # deceleration mu g, step by step
    k = np.arange(1, n_stop)
    travel = DT * (v1 + np.maximum(v1 - k * mu * G * DT, 0.0).sum())      # x after the push step already includes v1 * dt
    assert abs((xs[n_stop] - x0) - travel) < 0.02 * travel, (xs[n_stop] - x0, travel)
    assert abs(box.angle) < 1e-3 and abs(box.position.y - (hy + 0.015)) < 2e-3   # no tipping, rests at 2 polygonRadius - linearSlop above the edges
