"""A package named `Box2D` with just the classes madrl_environments/walker/multi_walker.py uses -- NOT Box2D.

TEST INFRASTRUCTURE ONLY.  The b2World here is the World of oracle/multiwalker_ref.c (this repository's independent plain-C restatement of
the Box2D 2.3.0 subset the env needs), reached through its `mwb_*` entry points.  Putting this directory on sys.path lets the UNMODIFIED
reference module import and run in an image that has no Box2D: its own reset() builds the world call by call (bodies, fixtures, joints, the
initial pushes), its own apply_action / get_observation / ContactDetector / LidarCallback / reward and termination code run on top of the
restated dynamics.  What comes out (oracle/make_golden_multiwalker.py -> tests/golden/multiwalker_envlayer_*.npz) pins the ENV LAYER of
this repository's MultiWalker to the reference's own code.  It does not pin the dynamics: b2World::Step is restated, not run.

Like pybox2d, every number crosses this boundary as a C float (Python floats are rounded to float32 on the way in, float32 values come
back widened).  Known reductions, all documented where they are used: RayCast reports only the closest edge fixture of category bit 1 (what
LidarCallback accepts: deviation D2 of multiwalker_ref.c), a world whose bodies have all been destroyed starts over as a fresh world
(D1), only the shape / joint kinds and keyword arguments the reference passes are understood (anything else raises).
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(_HERE))))
from oracle import multiwalker_ref as _mwr  # noqa: E402

# which build of the restatement: the product's documented sin / cos polynomial (default: recordings can then be compared with the product
# free-running, bit for bit) or libm's like Box2D (MADRL_BOX2D_SHIM_LIBM=1)
_POLY = os.environ.get("MADRL_BOX2D_SHIM_LIBM", "0") != "1"
_L = _mwr.lib(poly=_POLY)
_L.mwb_create.restype = C.c_void_p
_L.mwb_create.argtypes = [C.c_float, C.c_float, C.c_int]
_L.mwb_destroy.argtypes = [C.c_void_p]
_L.mwb_create_body.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int]
_L.mwb_create_revolute.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_float] * 8 + [C.c_int, C.c_int]
_L.mwb_apply_force_to_center.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_int]
_L.mwb_set_motor_speed.argtypes = [C.c_void_p, C.c_int, C.c_float]
_L.mwb_set_max_motor_torque.argtypes = [C.c_void_p, C.c_int, C.c_float]
_L.mwb_step.argtypes = [C.c_void_p, C.c_float, C.c_int, C.c_int]
_L.mwb_events.argtypes = [C.c_void_p, C.c_void_p]
_L.mwb_body_state.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
_L.mwb_joint_state.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
_L.mwb_raycast_closest.argtypes = [C.c_void_p] + [C.c_float] * 4 + [C.c_int, C.c_void_p]

CONTINUOUS_PHYSICS = True   # b2World's default; the known-answer scenes switch it off


def _f(x):
    return float(np.float32(x))


class b2Vec2(object):
    __slots__ = ("x", "y")

    def __init__(self, x=0.0, y=0.0):
        if isinstance(x, (tuple, list, b2Vec2, np.ndarray)):
            x, y = x[0], x[1]
        self.x, self.y = _f(x), _f(y)

    def __getitem__(self, i):
        return (self.x, self.y)[i]

    def __iter__(self):
        return iter((self.x, self.y))

    def __len__(self):
        return 2

    def __repr__(self):
        return "b2Vec2(%r, %r)" % (self.x, self.y)


class _Filter(object):
    def __init__(self, category, mask):
        self.categoryBits, self.maskBits = category, mask


class polygonShape(object):
    def __init__(self, vertices=None, box=None):
        assert (vertices is None) != (box is None), "polygonShape(vertices=...) or polygonShape(box=(hx, hy))"
        self.kind = 0 if box is None else 1
        self.data = [(_f(x), _f(y)) for x, y in vertices] if box is None else [(_f(box[0]), _f(box[1]))]
        assert box is None or len(box) == 2
        self.vertices = list(self.data)


class edgeShape(object):
    def __init__(self, vertices):
        assert len(vertices) == 2, "edgeShape(vertices=[p1, p2]): b2EdgeShape::Set(v1, v2), no ghost vertices"
        self.kind, self.data = 2, [(_f(x), _f(y)) for x, y in vertices]
        self.vertices = list(self.data)


class circleShape(object):   # imported by the reference module, never constructed
    def __init__(self, *a, **k):
        raise NotImplementedError("circleShape is not part of the restated subset")


class fixtureDef(object):
    def __init__(self, shape=None, density=0.0, friction=0.2, restitution=0.0, categoryBits=0x0001, maskBits=0xFFFF, **unknown):
        assert not unknown, "fixtureDef: unsupported keyword arguments %r" % sorted(unknown)
        assert _f(restitution) == 0.0, "restitution is not part of the restated subset"
        self.shape, self.density, self.friction, self.categoryBits, self.maskBits = shape, _f(density), _f(friction), int(categoryBits), int(maskBits)


class revoluteJointDef(object):
    def __init__(self, bodyA, bodyB, localAnchorA=(0, 0), localAnchorB=(0, 0), enableMotor=False, enableLimit=False, maxMotorTorque=0.0, motorSpeed=0.0,
                 lowerAngle=0.0, upperAngle=0.0, referenceAngle=0.0, collideConnected=False, **unknown):
        assert not unknown, "revoluteJointDef: unsupported keyword arguments %r" % sorted(unknown)
        assert _f(referenceAngle) == 0.0 and not collideConnected
        self.bodyA, self.bodyB, self.localAnchorA, self.localAnchorB = bodyA, bodyB, b2Vec2(localAnchorA), b2Vec2(localAnchorB)
        self.enableMotor, self.enableLimit = bool(enableMotor), bool(enableLimit)
        self.maxMotorTorque, self.motorSpeed, self.lowerAngle, self.upperAngle = _f(maxMotorTorque), _f(motorSpeed), _f(lowerAngle), _f(upperAngle)


class contactListener(object):
    def __init__(self):
        pass

    def BeginContact(self, contact):
        pass

    def EndContact(self, contact):
        pass


class rayCastCallback(object):
    def __init__(self, **kw):
        pass

    def ReportFixture(self, fixture, point, normal, fraction):
        raise NotImplementedError


class _Fixture(object):
    def __init__(self, body, fd):
        self.body, self.shape, self.filterData = body, fd.shape, _Filter(fd.categoryBits, fd.maskBits)
        self.density, self.friction = fd.density, fd.friction


class _Contact(object):
    def __init__(self, fa, fb):
        self.fixtureA, self.fixtureB = fa, fb


class b2Body(object):
    """attribute access reads the C world; anything else (color1, ground_contact, userData ...) is a plain Python attribute like on pybox2d's"""

    def __init__(self, world, index, fd, dynamic):
        self._world, self._index, self._dynamic, self._alive = world, index, dynamic, True
        self.fixtures = [_Fixture(self, fd)]
        self.userData = None

    def _state(self):
        assert self._alive and self._world._h, "body used after DestroyBody"
        out = (C.c_float * 10)()
        _L.mwb_body_state(self._world._h, self._index, out)
        return out

    position = property(lambda self: b2Vec2(*self._state()[0:2]))
    angle = property(lambda self: float(self._state()[2]))
    linearVelocity = property(lambda self: b2Vec2(*self._state()[3:5]))
    angularVelocity = property(lambda self: float(self._state()[5]))
    worldCenter = property(lambda self: b2Vec2(*self._state()[6:8]))
    awake = property(lambda self: bool(self._state()[8]))
    mass = property(lambda self: float(self._state()[9]))

    def ApplyForceToCenter(self, force, wake):
        f = b2Vec2(force)
        self._world._log.append(("ApplyForceToCenter", self._index, f.x, f.y))
        _L.mwb_apply_force_to_center(self._world._h, self._index, f.x, f.y, int(bool(wake)))


class b2RevoluteJoint(object):
    def __init__(self, world, index, jd):
        self._world, self._index, self.bodyA, self.bodyB = world, index, jd.bodyA, jd.bodyB

    def _state(self):
        out = (C.c_float * 4)()
        _L.mwb_joint_state(self._world._h, self._index, out)
        return out

    angle = property(lambda self: float(self._state()[0]))
    speed = property(lambda self: float(self._state()[1]))

    @property
    def motorSpeed(self):
        return float(self._state()[2])

    @motorSpeed.setter
    def motorSpeed(self, v):
        _L.mwb_set_motor_speed(self._world._h, self._index, _f(v))

    @property
    def maxMotorTorque(self):
        return float(self._state()[3])

    @maxMotorTorque.setter
    def maxMotorTorque(self, v):
        _L.mwb_set_max_motor_torque(self._world._h, self._index, _f(v))


class b2World(object):
    def __init__(self, gravity=(0, -10), doSleep=True):
        assert doSleep, "allowSleep = false is not part of the restated subset"
        self._gravity = b2Vec2(gravity)
        self._h = None
        self._bodies, self._joints = [], []
        self._log = []            # every creation call of the current world, in order: what tests compare the oracle's own reset with
        self.contactListener = None
        self.n_steps = 0

    def __del__(self):
        if getattr(self, "_h", None):
            _L.mwb_destroy(self._h)
            self._h = None

    def _handle(self):
        if self._h is None:       # first creation call, or first after every body was destroyed: a fresh world (D1 of multiwalker_ref.c)
            self._h = _L.mwb_create(self._gravity.x, self._gravity.y, int(CONTINUOUS_PHYSICS))
            self._bodies, self._joints, self._log = [], [], []
        return self._h

    def _create_body(self, dynamic, position, angle, fixtures):
        assert isinstance(fixtures, fixtureDef), "one fixtureDef per body (what the reference passes)"
        h = self._handle()
        sh = fixtures.shape
        flat = [c for v in sh.data for c in v]
        arr = (C.c_float * len(flat))(*flat)
        p = b2Vec2(position)
        idx = _L.mwb_create_body(h, int(dynamic), p.x, p.y, _f(angle), sh.kind, arr, len(sh.data), fixtures.density, fixtures.friction,
                                 fixtures.categoryBits, fixtures.maskBits)
        assert idx >= 0, "the restated world is full (or a polygon has more than 8 vertices)"
        b = b2Body(self, idx, fixtures, dynamic)
        self._bodies.append(b)
        self._log.append(("CreateBody", idx, int(dynamic), p.x, p.y, _f(angle), sh.kind, tuple(flat), fixtures.density, fixtures.friction,
                          fixtures.categoryBits, fixtures.maskBits))
        return b

    def CreateDynamicBody(self, position=(0, 0), angle=0.0, fixtures=None, **unknown):
        assert not unknown, "CreateDynamicBody: unsupported keyword arguments %r" % sorted(unknown)
        return self._create_body(True, position, angle, fixtures)

    def CreateStaticBody(self, position=(0, 0), angle=0.0, fixtures=None, **unknown):
        assert not unknown, "CreateStaticBody: unsupported keyword arguments %r" % sorted(unknown)
        return self._create_body(False, position, angle, fixtures)

    def CreateJoint(self, jd):
        assert isinstance(jd, revoluteJointDef)
        idx = _L.mwb_create_revolute(self._handle(), jd.bodyA._index, jd.bodyB._index, jd.localAnchorA.x, jd.localAnchorA.y, jd.localAnchorB.x,
                                     jd.localAnchorB.y, jd.lowerAngle, jd.upperAngle, jd.maxMotorTorque, jd.motorSpeed, int(jd.enableMotor), int(jd.enableLimit))
        assert idx >= 0, "only revolute joints with motor and limit enabled are restated"
        j = b2RevoluteJoint(self, idx, jd)
        self._joints.append(j)
        self._log.append(("CreateJoint", idx, jd.bodyA._index, jd.bodyB._index, jd.localAnchorA.x, jd.localAnchorA.y, jd.localAnchorB.x, jd.localAnchorB.y,
                          jd.lowerAngle, jd.upperAngle, jd.maxMotorTorque, jd.motorSpeed))
        return j

    def DestroyBody(self, body):
        """The reference destroys every body of the world in reset() (after clearing the contact listener, so no EndContact is delivered)
        and builds a new set: the C world is dropped once the last body is gone."""
        assert body._alive and body._world is self
        body._alive = False
        if not any(b._alive for b in self._bodies):
            _L.mwb_destroy(self._h)
            self._h = None

    def Step(self, dt, velocityIterations, positionIterations):
        assert all(b._alive for b in self._bodies), "a world with some of its bodies destroyed is not part of the restated subset"
        n = _L.mwb_step(self._handle(), _f(dt), int(velocityIterations), int(positionIterations))
        assert n >= 0, "contact event log overflow"
        self.n_steps += 1
        if n and self.contactListener is not None:
            ev = (C.c_int32 * (3 * n))()
            _L.mwb_events(self._h, ev)
            for k in range(n):   # b2ContactListener calls in the order the step made them (the listener here only sets flags, so delivering
                c = _Contact(self._bodies[ev[3 * k + 1]].fixtures[0], self._bodies[ev[3 * k + 2]].fixtures[0])   # them after the step changes nothing)
                (self.contactListener.BeginContact if ev[3 * k] else self.contactListener.EndContact)(c)

    def RayCast(self, callback, point1, point2):
        """b2World::RayCast hands the callback every fixture the ray crosses, in tree order; LidarCallback ignores all but category-bit-1
        fixtures (the terrain edges) and ends the query at the first of those it is handed.  Here it is handed the CLOSEST such edge, once
        (deviation D2: equal whenever the ray crosses one edge, which is the rule on this terrain)."""
        p1, p2 = b2Vec2(point1), b2Vec2(point2)
        out = (C.c_float * 5)()
        hit = _L.mwb_raycast_closest(self._handle(), p1.x, p1.y, p2.x, p2.y, 0x0001, out)
        if hit >= 0:
            callback.ReportFixture(self._bodies[hit].fixtures[0], b2Vec2(out[1], out[2]), b2Vec2(out[3], out[4]), float(out[0]))


class _B2Namespace(object):
    pass


b2 = sys.modules.setdefault(__name__ + ".b2", type(sys)(__name__ + ".b2"))
for _n in ("circleShape", "contactListener", "edgeShape", "fixtureDef", "polygonShape", "revoluteJointDef", "rayCastCallback", "world"):
    setattr(b2, _n, b2World if _n == "world" else globals()[_n])
b2.vec2 = b2Vec2
b2ContactListener, b2RayCastCallback = contactListener, rayCastCallback
__version__ = "shim over oracle/multiwalker_ref.c (Box2D 2.3.0 subset restated; NOT Box2D)"
