"""Message/time types for the FastSLAM facade.

The reference core takes ROS types at its boundary (prkt_core_v2.py:14-32):
``rospy.Time``/``Duration``, ``geometry_msgs.msg.Twist``, ``nav_msgs.msg.Odometry``,
``viz_feature_sim.msg.Blob``.  When ROS is importable the facade uses the real classes,
so ``prkt_ros.py`` and rviz see exactly what they saw before.  Without ROS (this image,
the GPU box) the small duck-typed stand-ins below carry the same attribute paths.
"""
from __future__ import annotations

import math
import time as _time


def _try_import(mod, name):
    try:
        m = __import__(mod, fromlist=[name])
        return getattr(m, name)
    except Exception:
        return None


def ros():
    """The rospy module if one is importable (real ROS or a test stub), else None."""
    try:
        import rospy  # noqa: F401

        return rospy
    except Exception:
        return None


# ---------------------------------------------------------------- stand-ins
class Duration(object):
    def __init__(self, secs=0.0):
        self._s = float(secs)

    @classmethod
    def from_sec(cls, s):
        return cls(s)

    def to_sec(self):
        return self._s

    @property
    def secs(self):
        return int(math.floor(self._s))


class Time(object):
    _override = None  # tests: Time.set_now(t)

    def __init__(self, secs=0.0):
        self._s = float(secs)

    @classmethod
    def now(cls):
        return cls(cls._override if cls._override is not None else _time.time())

    @classmethod
    def set_now(cls, t):
        cls._override = t

    def to_sec(self):
        return self._s

    def __sub__(self, other):
        if isinstance(other, Time):
            return Duration(self._s - other._s)
        return Time(self._s - other.to_sec())

    def __add__(self, other):
        return Time(self._s + other.to_sec())


class _Vec3(object):
    def __init__(self):
        self.x = 0.0
        self.y = 0.0
        self.z = 0.0


class _Quaternion(object):
    def __init__(self, x=0.0, y=0.0, z=0.0, w=0.0):
        self.x, self.y, self.z, self.w = x, y, z, w


class _Twist(object):
    def __init__(self):
        self.linear = _Vec3()
        self.angular = _Vec3()


class _Pose(object):
    def __init__(self):
        self.position = _Vec3()
        self.orientation = _Quaternion()


class _PoseWithCov(object):
    def __init__(self):
        self.pose = _Pose()


class _TwistWithCov(object):
    def __init__(self):
        self.twist = _Twist()


class _Header(object):
    def __init__(self):
        self.frame_id = ""
        self.stamp = None


class _Odometry(object):
    def __init__(self):
        self.header = _Header()
        self.pose = _PoseWithCov()
        self.twist = _TwistWithCov()


class _Color(object):
    def __init__(self):
        self.r = 0
        self.g = 0
        self.b = 0


class _Blob(object):
    def __init__(self, bearing=0.0, r=0, g=0, b=0):
        self.bearing = bearing
        self.size = 0
        self.color = _Color()
        self.color.r, self.color.g, self.color.b = r, g, b


Twist = _try_import("geometry_msgs.msg", "Twist") or _Twist
Quaternion = _try_import("geometry_msgs.msg", "Quaternion") or _Quaternion
Odometry = _try_import("nav_msgs.msg", "Odometry") or _Odometry
Blob = _try_import("viz_feature_sim.msg", "Blob") or _Blob


def now():
    r = ros()
    return r.Time.now() if r is not None else Time.now()


# ---------------------------------------------------------------- heading <-> quaternion
def heading_to_quaternion(heading):
    """utils.heading_to_quaternion (utils.py:21-35): yaw -> (0, 0, sin(h/2), cos(h/2))."""
    q = Quaternion()
    q.x = 0.0
    q.y = 0.0
    q.z = math.sin(heading / 2.0)
    q.w = math.cos(heading / 2.0)
    return q


def quaternion_to_heading(q):
    """utils.quaternion_to_heading (utils.py:8-19): yaw of a quaternion (x, y, z, w); the
    all-zero default quaternion of a fresh Odometry reads as heading 0, as tf does."""
    try:
        x, y, z, w = q.x, q.y, q.z, q.w
    except AttributeError:
        x, y, z, w = q
    n = x * x + y * y + z * z + w * w
    if n < 8.881784197001252e-16:  # 4 * eps: tf.transformations.quaternion_matrix -> identity
        return 0.0
    s = 2.0 / n
    return math.atan2(s * (x * y + z * w), 1.0 - s * (y * y + z * z))
