"""CPU, build container only: the oracle against the LIVE reference on fresh random inputs
(beyond the committed goldens).  Skipped wherever /root/reference is absent (the GPU box)."""
import math
import os
import random

import numpy as np
import pytest

from oracle import fastslam_oracle as O
from oracle import ros_stubs

pytestmark = pytest.mark.skipif(not os.path.isdir(ros_stubs.REFERENCE_SRC), reason="reference tree not present")


@pytest.fixture(scope="module")
def ref():
    core = ros_stubs.import_reference()
    return core


def test_random_triples_live(ref):
    from utils import heading_to_quaternion
    from viz_feature_sim.msg import Blob

    rs = np.random.RandomState(4242)
    for i in range(40):
        pose = (rs.uniform(-3, 3), rs.uniform(-3, 3), rs.uniform(-3, 3))
        mean = np.array([rs.uniform(-25, 25), rs.uniform(-25, 25), *rs.uniform(0, 255, 3)])
        a = rs.normal(size=(5, 5))
        cov = a @ a.T / 5 + 0.2 * np.identity(5)
        tb = math.atan2(mean[1] - pose[1], mean[0] - pose[0]) - pose[2]
        blob = (tb + rs.uniform(-0.6, 0.6), *(mean[2:] + rs.uniform(-12, 12, 3)))
        p = ref.FilterParticle()
        p.state.pose.pose.position.x, p.state.pose.pose.position.y = pose[0], pose[1]
        p.state.pose.pose.orientation = heading_to_quaternion(pose[2])
        f = ref.Feature(mean=mean.copy(), covar=cov.copy())
        p.feature_set[1] = f
        from utils import quaternion_to_heading

        hd = quaternion_to_heading(p.state.pose.pose.orientation)
        assert abs(hd - float(O.wrap_heading(pose[2]))) < 1e-15
        want = float(p.probability_of_match(p.state, Blob(*blob), f))
        got = float(O.probability_of_match(pose[0], pose[1], hd, blob, mean, cov))
        assert abs(got - want) <= 1e-11 * abs(want), i
        H = p.measurement_jacobian(1)
        Q = p.measurement_covariance(H, 1, 0.1 * np.identity(4))
        K = p.kalman_gain(1, H, ref.inverse(Q))
        w = float(p.importance_factor(Q, Blob(*blob), p.generate_measurement(1)))
        nm, nc, ow, aux = O.ekf_update_dense(pose[0], pose[1], mean, cov, blob, 0.1 * np.identity(4))
        assert np.allclose(aux["K"], K, rtol=1e-11, atol=1e-14)
        assert abs(ow - w) <= 1e-11 * w


def test_resample_random_live(ref):
    rs = np.random.RandomState(77)
    for trial in range(30):
        P = int(rs.randint(2, 400))
        w = np.exp(rs.normal(0, rs.uniform(0.1, 8), P))
        fs = ref.FastSLAM([])
        fs.particles = [ref.FilterParticle() for _ in range(P)]
        for i, (p, wi) in enumerate(zip(fs.particles, w)):
            p.weight = float(wi)
            p._idx = i
        random.seed(trial)
        fs.low_variance_resample()
        u = random.Random(trial).random()
        anc = np.array([p._idx for p in fs.particles])
        assert np.array_equal(O.low_variance_ancestors(w, u), anc), trial


def test_new_landmark_geometry_live(ref):
    """f4 (prkt_core_v2.py:546-746): the facade's host restatement of the reading geometry against the live
    reference on random readings.  The two sides extract the heading from the quaternion with different
    (stubbed tf vs. closed form) arithmetic, hence the 1e-12 on the crossing point; verdicts must agree."""
    from nav_msgs.msg import Odometry
    from utils import heading_to_quaternion
    from viz_feature_sim.msg import Blob

    import parakeet_slam_amd.core as C

    rs = np.random.RandomState(1)
    rp, mp = ref.FilterParticle(), C.FilterParticle()

    class IterDict(dict):  # py2 dict.iteritems, which prkt_core_v2.py:579 calls
        def iteritems(self):
            return iter(self.items())

    rp.potential_features = IterDict()

    def reading():
        st = Odometry()
        st.pose.pose.position.x, st.pose.pose.position.y = rs.uniform(-5, 5, 2)
        st.pose.pose.orientation = heading_to_quaternion(rs.uniform(-3, 3))
        b = Blob()
        b.bearing = rs.uniform(-3, 3)
        b.color.r, b.color.g, b.color.b = rs.uniform(0, 255, 3)
        return st, b

    crossing = 0
    for i in range(1500):
        a, b = reading(), reading()
        want, got = rp.reading_distance_function(a[0], a[1], b[0], b[1]), mp.reading_distance_function(a[0], a[1], b[0], b[1])
        assert want == got, i
        crossing += want < float("inf")
        assert rp.color_distance(a[1], b[1]) == mp.color_distance(a[1], b[1])
        x, y = rp.cross_readings(a, b), mp.cross_readings(a, b)
        assert np.allclose(x, y, rtol=1e-12, atol=1e-12), i
        args = (rs.uniform(-3, 3), rs.uniform(-3, 3), a[1].bearing, rs.uniform(-3, 3), rs.uniform(-3, 3), b[1].bearing)
        assert rp.ray_intersect(*args) == mp.ray_intersect(*args) == O.ray_intersect(*args)
        # the oracle's restatement (what GrowingOracle runs on), on the same readings
        from utils import quaternion_to_heading

        ha, hb = quaternion_to_heading(a[0].pose.pose.orientation), quaternion_to_heading(b[0].pose.pose.orientation)
        pa, pb = a[0].pose.pose.position, b[0].pose.pose.position
        ox = O.cross_readings(pa.x, pa.y, ha + a[1].bearing, pb.x, pb.y, hb + b[1].bearing)
        assert (ox is None) == (x is None) and (x is None or np.allclose(ox, x, rtol=1e-12, atol=1e-12))
        ca, cb = (a[1].color.r, a[1].color.g, a[1].color.b), (b[1].color.r, b[1].color.g, b[1].color.b)
        assert O.colour_distance(ca, cb) == rp.color_distance(a[1], b[1])
        inter = O.ray_intersect(pa.x, pa.y, a[1].bearing + ha, pb.x, pb.y, b[1].bearing + hb)
        assert (O.colour_distance(ca, cb) if inter else float("inf")) == want
    assert 200 < crossing < 1300
    # the bookkeeping: orphans pile up, a stored reading in potential_features is found by its (negative) key
    r1, r2, r3 = reading(), reading(), reading()
    for p in (rp, mp):
        p.add_hypothesis(*r1)
        p.add_hypothesis(*r2)
    assert sorted(rp.hypothesis_set) == sorted(mp.hypothesis_set) == [1, 2] and rp.next_id == mp.next_id == 3
    for p in (rp, mp):
        p.add_new_feature(1, *r3)
    assert list(rp.potential_features) == list(mp.potential_features) == [-3]
    assert np.array_equal(np.asarray(rp.potential_features[-3].covar), mp.potential_features[-3].covar)
    assert np.allclose(np.asarray(rp.potential_features[-3].mean, dtype=float), mp.potential_features[-3].mean, rtol=1e-12)
    # find_nearest_reading walks potential_features: the stored READINGS of the reference's own unit test (:228-274)
    for p in (rp, mp):
        p.potential_features.clear()
        p.potential_features[-1] = r1
        p.potential_features[-2] = r2
    for k in range(50):
        q = reading()
        assert rp.find_nearest_reading(*q) == mp.find_nearest_reading(*q)
