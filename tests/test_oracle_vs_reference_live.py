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
