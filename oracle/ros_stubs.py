"""Minimal stand-ins for the ROS python modules the reference imports.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the shipped
product; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it.

The reference core (``/root/reference/src/prkt_core_v2.py:14-35``) imports
``rospy``, ``geometry_msgs.msg``, ``nav_msgs.msg``, ``viz_feature_sim.msg`` and
(through ``utils.py:6``) ``tf.transformations``.  None of those exist in this
image, so ``install()`` places small pure-python modules with the same names
in ``sys.modules``.  They are used (a) by ``oracle/make_golden.py`` to import
the *unmodified* reference in this container and capture golden vectors, and
(b) by the unit tests that replay the reference's own assertions.

The message classes only carry the attributes the hot path reads:
``Twist.linear.x / .angular.z`` (prkt_core_v2.py:176-177),
``Odometry.pose.pose.position.{x,y}`` / ``.orientation`` / ``.header.frame_id``
(:126, :203-206, :402-404), ``Blob.bearing / .color.{r,g,b}`` (:409, :425-427),
``VizScan.observes`` (:344).

``tf.transformations`` is third-party (ROS indigo ``tf`` package, not vendored
under /root/reference and not pinned by package.xml:50,58).  The two functions
used by ``utils.py:18,28`` are restated here from their published algorithm
(Gohlke's ``transformations.py``, static-xyz axes):

* ``quaternion_from_euler(ai, aj, ak)`` for axes 'sxyz' -> (x, y, z, w)
* ``euler_from_quaternion(q)`` = ``euler_from_matrix(quaternion_matrix(q))``
  with ``quaternion_matrix`` normalising by ``sqrt(2 / q.q)`` and returning
  identity when ``q.q < 4 * eps``.
"""
from __future__ import annotations

import math
import sys
import types

import numpy as np

_EPS4 = np.finfo(float).eps * 4.0


# --------------------------------------------------------------------------- rospy
class Duration(object):
    def __init__(self, secs=0, nsecs=0):
        total = int(secs) * 10**9 + int(nsecs)
        self.secs, self.nsecs = divmod(total, 10**9)

    @classmethod
    def from_sec(cls, s):
        secs = int(math.floor(s))
        nsecs = int((s - secs) * 1e9)
        return cls(secs, nsecs)

    def to_sec(self):
        return float(self.secs) + float(self.nsecs) / 1e9

    def _ns(self):
        return self.secs * 10**9 + self.nsecs


class Time(object):
    """Controllable clock: ``Time.set_now(t)`` / ``Time.advance(dt)``."""

    _now_ns = 0

    def __init__(self, secs=0, nsecs=0):
        total = int(secs) * 10**9 + int(nsecs)
        self.secs, self.nsecs = divmod(total, 10**9)

    @classmethod
    def now(cls):
        return cls(0, cls._now_ns)

    @classmethod
    def set_now(cls, t):
        cls._now_ns = int(round(t * 1e9))

    @classmethod
    def advance(cls, dt):
        cls._now_ns += int(round(dt * 1e9))

    def to_sec(self):
        return float(self.secs) + float(self.nsecs) / 1e9

    def _ns(self):
        return self.secs * 10**9 + self.nsecs

    def __sub__(self, other):
        if isinstance(other, Time):
            return Duration(0, self._ns() - other._ns())
        return Time(0, self._ns() - other._ns())

    def __add__(self, other):
        return Time(0, self._ns() + other._ns())


class Publisher(object):
    def __init__(self, *a, **k):
        self.count = 0

    def publish(self, *a, **k):
        self.count += 1


class Subscriber(object):
    def __init__(self, *a, **k):
        pass


class Rate(object):
    def __init__(self, hz):
        self.hz = hz

    def sleep(self):
        pass


def _make_rospy():
    m = types.ModuleType("rospy")
    m.Time = Time
    m.Duration = Duration
    m.Publisher = Publisher
    m.Subscriber = Subscriber
    m.Rate = Rate
    m.is_shutdown = lambda: False
    m.loginfo = lambda *a, **k: None
    m.logwarn = lambda *a, **k: None
    m.init_node = lambda *a, **k: None
    return m


# --------------------------------------------------------------------------- msgs
class _Vec3(object):
    def __init__(self):
        self.x = 0.0
        self.y = 0.0
        self.z = 0.0


class Quaternion(object):
    def __init__(self, x=0.0, y=0.0, z=0.0, w=0.0):
        self.x, self.y, self.z, self.w = x, y, z, w


class Twist(object):
    def __init__(self):
        self.linear = _Vec3()
        self.angular = _Vec3()


class _Pose(object):
    def __init__(self):
        self.position = _Vec3()
        self.orientation = Quaternion()


class _PoseWithCov(object):
    def __init__(self):
        self.pose = _Pose()
        self.covariance = [0.0] * 36


class _TwistWithCov(object):
    def __init__(self):
        self.twist = Twist()
        self.covariance = [0.0] * 36


class _Header(object):
    def __init__(self):
        self.seq = 0
        self.stamp = Time()
        self.frame_id = ""


class Odometry(object):
    def __init__(self):
        self.header = _Header()
        self.child_frame_id = ""
        self.pose = _PoseWithCov()
        self.twist = _TwistWithCov()


class _Color(object):
    def __init__(self):
        self.r = 0
        self.g = 0
        self.b = 0
        self.a = 0


class Blob(object):
    def __init__(self, bearing=0.0, r=0, g=0, b=0):
        self.bearing = bearing
        self.size = 0
        self.color = _Color()
        self.color.r, self.color.g, self.color.b = r, g, b


class VizScan(object):
    def __init__(self, observes=None):
        self.header = _Header()
        self.observes = list(observes) if observes is not None else []


class Observation(object):
    pass


# --------------------------------------------------------------------------- tf
def quaternion_from_euler(ai, aj, ak, axes="sxyz"):
    """Static-xyz euler -> quaternion (x, y, z, w); tf.transformations algorithm."""
    if axes != "sxyz":
        raise NotImplementedError(axes)
    ai /= 2.0
    aj /= 2.0
    ak /= 2.0
    ci, si = math.cos(ai), math.sin(ai)
    cj, sj = math.cos(aj), math.sin(aj)
    ck, sk = math.cos(ak), math.sin(ak)
    cc, cs = ci * ck, ci * sk
    sc, ss = si * ck, si * sk
    q = np.empty((4,), dtype=np.float64)
    q[0] = cj * sc - sj * cs
    q[1] = cj * ss + sj * cc
    q[2] = cj * cs - sj * sc
    q[3] = cj * cc + sj * ss
    return q


def quaternion_matrix(quaternion):
    q = np.array(quaternion[:4], dtype=np.float64, copy=True)
    nq = np.dot(q, q)
    if nq < _EPS4:
        return np.identity(4)
    q *= math.sqrt(2.0 / nq)
    q = np.outer(q, q)
    return np.array(
        (
            (1.0 - q[1, 1] - q[2, 2], q[0, 1] - q[2, 3], q[0, 2] + q[1, 3], 0.0),
            (q[0, 1] + q[2, 3], 1.0 - q[0, 0] - q[2, 2], q[1, 2] - q[0, 3], 0.0),
            (q[0, 2] - q[1, 3], q[1, 2] + q[0, 3], 1.0 - q[0, 0] - q[1, 1], 0.0),
            (0.0, 0.0, 0.0, 1.0),
        ),
        dtype=np.float64,
    )


def euler_from_matrix(matrix, axes="sxyz"):
    if axes != "sxyz":
        raise NotImplementedError(axes)
    M = np.array(matrix, dtype=np.float64, copy=False)[:3, :3]
    cy = math.sqrt(M[0, 0] * M[0, 0] + M[1, 0] * M[1, 0])
    if cy > _EPS4:
        ax = math.atan2(M[2, 1], M[2, 2])
        ay = math.atan2(-M[2, 0], cy)
        az = math.atan2(M[1, 0], M[0, 0])
    else:
        ax = math.atan2(-M[1, 2], M[1, 1])
        ay = math.atan2(-M[2, 0], cy)
        az = 0.0
    return ax, ay, az


def euler_from_quaternion(quaternion, axes="sxyz"):
    return euler_from_matrix(quaternion_matrix(quaternion), axes)


# --------------------------------------------------------------------------- install
def install():
    """Place the stub modules in ``sys.modules`` (idempotent)."""
    if "rospy" in sys.modules and getattr(sys.modules["rospy"], "_pk_stub", False):
        return
    rospy = _make_rospy()
    rospy._pk_stub = True
    sys.modules["rospy"] = rospy

    def _pkg(name, **attrs):
        pkg = types.ModuleType(name)
        msg = types.ModuleType(name + ".msg")
        for k, v in attrs.items():
            setattr(msg, k, v)
        pkg.msg = msg
        sys.modules[name] = pkg
        sys.modules[name + ".msg"] = msg

    _pkg("geometry_msgs", Twist=Twist, Quaternion=Quaternion)
    _pkg("nav_msgs", Odometry=Odometry)
    _pkg("viz_feature_sim", Blob=Blob, VizScan=VizScan, Observation=Observation)

    tf = types.ModuleType("tf")
    tft = types.ModuleType("tf.transformations")
    tft.quaternion_from_euler = quaternion_from_euler
    tft.euler_from_quaternion = euler_from_quaternion
    tft.quaternion_matrix = quaternion_matrix
    tft.euler_from_matrix = euler_from_matrix
    tf.transformations = tft
    sys.modules["tf"] = tf
    sys.modules["tf.transformations"] = tft

    # py2-isms in the reference: xrange (prkt_core_v2.py:159)
    import builtins

    if not hasattr(builtins, "xrange"):
        builtins.xrange = range


REFERENCE_SRC = "/root/reference/src"


def import_reference():
    """Import the unmodified reference core (only possible in the build container).

    Returns the ``prkt_core_v2`` module, or raises ImportError when
    ``/root/reference`` is absent (e.g. on the GPU box).
    """
    import os

    if not os.path.isdir(REFERENCE_SRC):
        raise ImportError("reference tree not present: %s" % REFERENCE_SRC)
    install()
    if REFERENCE_SRC not in sys.path:
        sys.path.insert(0, REFERENCE_SRC)
    import warnings

    warnings.filterwarnings("ignore", category=DeprecationWarning)
    import prkt_core_v2  # noqa: E402

    return prkt_core_v2
