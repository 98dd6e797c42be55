"""CPU: the reference's own known-answer assertions (src/test_prkt_ros2.py) and
the vectors quoted in SURVEY 8a, restated against the oracle."""
import math

import numpy as np

from oracle import fastslam_oracle as O


def test_prob_position_match_reference_test():
    # test_prkt_ros2.py:126-152
    cov = (0.1, 0.0, 0.1)
    r1 = float(O.prob_position_match(1.0, 0.0, cov, 0.0, 0.0, 0.0))
    assert r1 > 1.59 and abs(r1 - 1 / (2 * math.pi * 0.1)) < 1e-14
    r2 = float(O.prob_position_match(1.0, 0.0, cov, 0.0, 0.0, 1.0))
    assert 0.0 < r2 < 0.05
    r2m = float(O.prob_position_match(1.0, 0.0, cov, 0.0, 0.0, -1.0))
    assert 0.0 < r2m < 0.05
    assert float(O.prob_position_match(1.0, 0.0, cov, 0.0, 0.0, math.pi)) == 0.0  # :474


def test_closest_point_reference_test():
    # test_prkt_ros2.py:154-189
    cx, cy = O.closest_point(1.0, 0.0, 0.0, 0.0, 0.0)
    assert (float(cx), float(cy)) == (1.0, 0.0)
    cx, cy = O.closest_point(1.0, 0.0, 0.0, 0.0, math.pi / 2)
    assert cx < 1e-5 and cy < 1e-5
    for b in (math.pi, 3 * math.pi / 4, -3 * math.pi / 4):
        cx, cy = O.closest_point(1.0, 0.0, 0.0, 0.0, b)
        assert (float(cx), float(cy)) == (0.0, 0.0)


def test_prob_color_match_reference_test():
    # test_prkt_ros2.py:191-222
    cov = (5.0, 0.0, 0.0, 5.0, 0.0, 5.0)
    r1 = float(O.prob_color_match((255, 0, 0), cov, (255, 0, 0)))
    assert r1 > 0.005 and abs(r1 - 1 / math.sqrt((2 * math.pi) ** 3 * 125)) < 1e-16
    r2 = float(O.prob_color_match((255, 0, 0), cov, (250, 0, 0)))
    r3 = float(O.prob_color_match((255, 0, 0), cov, (250, 0, 5)))
    r4 = float(O.prob_color_match((255, 0, 0), cov, (200, 0, 5)))
    assert r2 < r1 and r3 < r2 and r4 < r2


def test_gates_reference_tests():
    # test_prkt_ros2.py:98-124: exact zeros
    mean = np.array([1.0, 0, 0, 0, 0])
    cov = np.identity(5)
    assert float(O.probability_of_match(0, 0, 0, (0.0, 255, 0, 0), mean, cov)) == 0.0
    assert float(O.probability_of_match(0, 0, 0, (math.pi, 0, 0, 0), mean, cov)) == 0.0


def test_generate_measurement_reference_test():
    # test_prkt_ros2.py:403-423
    _, _, _, aux = O.ekf_update_dense(-1.0, -1.0, np.array([0, 0, 73, 165, 255.0]), np.identity(5), (0, 0, 0, 0),
                                      0.1 * np.identity(4))
    assert aux["zhat"][0] == math.pi / 4
    assert list(aux["zhat"][1:]) == [73, 165, 255]


def test_survey_known_answer_vector():
    # SURVEY.md 8a "Known-answer vector captured from the reference"
    pose = (0.5, -0.25, 0.3)
    mean = np.array([3, 4, 100, 150, 200.0])
    cov = 0.25 * np.identity(5)
    blob = (0.9, 101, 149, 202)
    assert abs(float(O.probability_of_match(*pose, blob, mean, cov)) - 7.804697433262233e-07) < 1e-19
    nm, nc, w, aux = O.ekf_update_dense(pose[0], pose[1], mean, cov, blob, 0.1 * np.identity(4))
    assert abs(aux["zhat"][0] - 1.039072259536091) < 1e-15
    assert np.allclose(np.diag(aux["Q"]), [0.11028277635, .35, .35, .35], rtol=1e-10)
    assert abs(w - 8.819701295333626e-05) < 1e-18
    assert np.allclose(nm, [2.944889780603, 3.967582223884, 100.714285714286, 149.285714285714, 201.428571428571],
                       rtol=1e-12)
    assert np.allclose(nc[2:, 2:], 0.071428571429 * np.identity(3), rtol=1e-10, atol=1e-15)


def test_heading_wrap_matches_quaternion_round_trip():
    # SURVEY 8a a2 probe: wrap(3.5) = -2.78318...
    assert abs(float(O.wrap_heading(3.5)) - (3.5 - 2 * math.pi)) < 1e-15
    assert float(O.wrap_heading(0.0)) == 0.0


def test_one_full_step_summary_from_survey():
    # P=50, the 4 prkt_ros.py landmarks made mutable, seeds 7
    import random

    means = np.array([[0, 25, 161, 77, 137], [10, 25, 75, 55, 230], [0, 15, 82, 120, 68], [10, 15, 224, 37, 192.0]])
    covs = np.broadcast_to(0.25 * np.identity(5), (4, 5, 5))
    f = O.OracleFilter(50, means, covs)
    z = np.random.RandomState(7).standard_normal((50, 3))
    u = random.Random(7).random()
    blobs = np.column_stack([np.arctan2(means[:, 1], means[:, 0]), means[:, 2:]])
    f.step(0.2, 0.1, 0.1, z, blobs, u)
    assert np.allclose(f.summary(), (0.0190041676078814, 7.642232493308172e-05, 0.009759447788568397), rtol=0,
                       atol=1e-15)
