#!/usr/bin/env python3
"""Generate golden vectors by running the UNMODIFIED reference in this container.

TEST INFRASTRUCTURE ONLY.  Run from the repo root:

    python oracle/make_golden.py            # writes tests/golden/*.npz

It imports ``/root/reference/src/prkt_core_v2.py`` through ``oracle/ros_stubs``
(the reference tree never travels to the GPU box; only the ``.npz`` fixtures it
produced do).  Each fixture stores inputs and the reference's outputs:

  ka_triples.npz     per (pose, landmark, blob) known answers for every scalar
                     function on the path (a4..a11 of SURVEY 8a)
  step_small.npz     P=16, L=6 full-state trajectory incl. an unmatched blob, a
                     doubly matched landmark and an immutable landmark
  step_refscene.npz  the reference's own scene (prkt_ros.py:33-52): 4 immutable
                     landmarks, P=50
  step_config1.npz   BASELINE.json config 1: P=100, L=B=50, 3 steps (poses,
                     ancestors, summaries, strided landmark sample)
  motion.npz         motion_update sequences incl. heading wrap
  resample.npz       low_variance_resample ancestor lists for crafted weights
  step_potential.npz the potential-feature branch of cam_cb (:109-118, :366-367): reference particles whose
                     `potential_features` are populated by hand (negative ids, update_count 0..6, one immutable),
                     P=12, 5 full + 4 potential landmarks per particle, 3 steps: ids (negative for potential
                     matches), weights, per-slot means / covariances / counts / potential flags, promotions

Versions used for the committed fixtures: Python 3.10.12, NumPy 2.2.6, SciPy 1.15.3.
"""
from __future__ import annotations

import math
import os
import random as pyrandom
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import ros_stubs  # noqa: E402

core = ros_stubs.import_reference()
import rospy  # noqa: E402  (stub)
from geometry_msgs.msg import Twist  # noqa: E402
from utils import heading_to_quaternion, quaternion_to_heading  # noqa: E402
from viz_feature_sim.msg import Blob, VizScan  # noqa: E402

from oracle.fastslam_oracle import synthetic_scan, synthetic_world, truth_step  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


class IterDict(dict):
    """py2 ``dict.iteritems`` for prkt_core_v2.py:579 (reached when a blob is unmatched)."""

    def iteritems(self):
        return iter(self.items())


class View(object):
    """Stands in for the CamSlam360 node that cam_cb reads the scan from (:82)."""

    def __init__(self, blobs):
        self.last_sensor_reading = VizScan([Blob(b[0], b[1], b[2], b[3]) for b in blobs])


def build_filter(P, means, covs, immutable):
    feats = []
    for m, c, im in zip(means, covs, immutable):
        f = core.Feature(mean=np.array(m), covar=np.array(c))
        f.__immutable__ = bool(im)
        feats.append(f)
    fs = core.FastSLAM(feats)
    fs.num_particles = P
    fs.particles = [core.FilterParticle() for _ in range(P)]
    for p in fs.particles:
        p.load_feature_list(feats)
        p.potential_features = IterDict()
    return fs


def poses_of(fs):
    out = np.empty((len(fs.particles), 4))
    for i, p in enumerate(fs.particles):
        out[i, 0] = float(p.state.pose.pose.position.x)
        out[i, 1] = float(p.state.pose.pose.position.y)
        out[i, 2] = float(quaternion_to_heading(p.state.pose.pose.orientation))
        out[i, 3] = float(p.weight)
    return out


def maps_of(fs, lsel=None):
    P = len(fs.particles)
    ids = sorted(fs.particles[0].feature_set.keys())
    if lsel is not None:
        ids = [ids[i] for i in lsel]
    L = len(ids)
    mean = np.empty((P, L, 5))
    cov = np.empty((P, L, 5, 5))
    cnt = np.empty((P, L), dtype=np.int64)
    for i, p in enumerate(fs.particles):
        for j, id_ in enumerate(ids):
            f = p.feature_set[id_]
            mean[i, j] = np.asarray(f.mean, dtype=np.float64)
            cov[i, j] = np.asarray(f.covar, dtype=np.float64)
            cnt[i, j] = f.update_count
    return mean, cov, cnt


class Recorder(object):
    """Wraps the module-level RNG callables the reference imported by name."""

    def __init__(self):
        self.normals = []
        self.uniforms = []
        self._normal = core.normal
        self._random = core.random
        core.normal = self.normal
        core.random = self.random

    def normal(self, loc, scale, size):
        v = self._normal(loc, scale, size)
        self.normals.append((float(scale), float(v[0])))
        return v

    def random(self):
        u = self._random()
        self.uniforms.append(u)
        return u

    def restore(self):
        core.normal = self._normal
        core.random = self._random


def run_steps(P, means, covs, immutable, blobs_per_step, v, w, dts, seed, lsel=None, ids_capture=True):
    """Run cam_cb once per entry of blobs_per_step; capture everything per step."""
    np.random.seed(seed)
    pyrandom.seed(seed)
    zstream = np.random.RandomState(seed)
    rospy.Time.set_now(0.0)
    fs = build_filter(P, means, covs, immutable)
    tw = Twist()
    tw.linear.x = v
    tw.angular.z = w
    fs.last_control = tw
    rec = Recorder()
    S = len(blobs_per_step)
    out = dict(
        z=[], post_motion=[], ids=[], weights=[], u=[], ancestors=[], post_resample=[], summary=[],
        mean=[], cov=[], count=[],
    )
    try:
        for s in range(S):
            blobs = blobs_per_step[s]
            rospy.Time.advance(dts[s])
            view = View(blobs)
            n0 = len(rec.normals)
            # --- instrument: capture post-motion poses and ids by replaying the
            # association on a deepcopy AFTER the motion update.  cam_cb does motion
            # inside the loop at i == 0 (:75-77), so run it through a wrapper.
            captured = {}
            orig_motion_update = fs.motion_update

            def mu(tw_, _orig=orig_motion_update, _c=captured):
                _orig(tw_)
                _c["post_motion"] = poses_of(fs)
                if ids_capture:
                    _c["ids"] = np.array(
                        [[pr[0] for pr in p.match_features_to_scan(view.last_sensor_reading)] for p in fs.particles],
                        dtype=np.int32,
                    )

            fs.motion_update = mu
            orig_resample = fs.low_variance_resample

            def rs_(_orig=orig_resample, _c=captured):
                _c["weights"] = np.array([float(p.weight) for p in fs.particles])
                for i, p in enumerate(fs.particles):
                    p._golden_index = i
                _orig()
                _c["ancestors"] = np.array([p._golden_index for p in fs.particles], dtype=np.int64)

            fs.low_variance_resample = rs_
            fs.cam_cb(view)
            fs.motion_update = orig_motion_update
            fs.low_variance_resample = orig_resample
            drawn = rec.normals[n0:]
            assert len(drawn) == 3 * P, (len(drawn), P)
            z = zstream.standard_normal(3 * P).reshape(P, 3)
            scales = np.array([d[0] for d in drawn]).reshape(P, 3)
            vals = np.array([d[1] for d in drawn]).reshape(P, 3)
            assert np.array_equal(0.0 + scales * z, vals), "legacy normal != loc + scale*gauss"
            out["z"].append(z)
            out["post_motion"].append(captured["post_motion"])
            if ids_capture:
                out["ids"].append(captured["ids"])
            out["weights"].append(captured["weights"])
            out["u"].append(rec.uniforms[-1])
            out["ancestors"].append(captured["ancestors"])
            out["post_resample"].append(poses_of(fs))
            out["summary"].append(np.array(fs.summary()))
            m, c, n = maps_of(fs, lsel)
            out["mean"].append(m)
            out["cov"].append(c)
            out["count"].append(n)
    finally:
        rec.restore()
    res = {k: np.array(vv) for k, vv in out.items() if len(vv)}
    res.update(
        P=P, L=len(means), v=v, w=w, dts=np.array(dts), seed=seed,
        means0=np.array(means, dtype=np.float64), covs0=np.array(covs, dtype=np.float64),
        immutable=np.array(immutable, dtype=np.uint8), Qt=np.array(fs.Qt),
        blobs=np.array(blobs_per_step, dtype=np.float64),
    )
    if lsel is not None:
        res["lsel"] = np.array(lsel)
    return res


# ------------------------------------------------------------------------- triples
def random_spd(rs, n, scale):
    a = rs.normal(size=(n, n))
    return scale * (a @ a.T / n + 0.3 * np.identity(n))


def gen_triples():
    rs = np.random.RandomState(2024)
    rows = []
    cases = []
    # SURVEY 8a known answer first
    cases.append(((0.5, -0.25, 0.3), np.array([3, 4, 100, 150, 200.0]), 0.25 * np.identity(5), (0.9, 101, 149, 202)))
    for i in range(95):
        pose = (rs.uniform(-2, 2), rs.uniform(-2, 2), rs.uniform(-0.6, 0.6))
        mean = np.array([rs.uniform(-20, 20), rs.uniform(-20, 20), rs.uniform(0, 255), rs.uniform(0, 255), rs.uniform(0, 255)])
        cov = np.zeros((5, 5))
        kind = i % 4
        if kind == 0:
            cov = rs.uniform(0.05, 2.0) * np.identity(5)
        elif kind in (1, 2):
            cov[:2, :2] = random_spd(rs, 2, rs.uniform(0.05, 1.0))
            cov[2:, 2:] = random_spd(rs, 3, rs.uniform(0.5, 8.0))
        else:
            cov = random_spd(rs, 5, rs.uniform(0.1, 1.0))  # dense: xy-rgb cross terms
        tb = math.atan2(mean[1] - pose[1], mean[0] - pose[0]) - pose[2]
        j = i % 6
        db = rs.uniform(-0.45, 0.45) if j != 5 else rs.choice([-1, 1]) * rs.uniform(0.5001, 0.9)
        dc = rs.uniform(-9, 9, size=3) if j != 4 else rs.uniform(-30, 30, size=3)
        blob = (tb + db, mean[2] + dc[0], mean[3] + dc[1], mean[4] + dc[2])
        cases.append((pose, mean, cov, blob))
    # landmark exactly at the robot: q == 0 branch of measurement_jacobian (:788-797)
    cases.append(((1.0, 2.0, 0.1), np.array([1.0, 2.0, 10, 20, 30.0]), 0.5 * np.identity(5), (0.0, 11, 19, 31)))
    Qt = 0.1 * np.identity(4)
    for pose, mean, cov, blob in cases:
        p = core.FilterParticle()
        p.state.pose.pose.position.x = pose[0]
        p.state.pose.pose.position.y = pose[1]
        p.state.pose.pose.orientation = heading_to_quaternion(pose[2])
        f = core.Feature(mean=mean.copy(), covar=cov.copy())
        p.feature_set[1] = f
        b = Blob(*blob)
        hd = float(quaternion_to_heading(p.state.pose.pose.orientation))
        pom = float(p.probability_of_match(p.state, b, f))
        ppm = float(p.prob_position_match(f.mean, f.covar, pose[0], pose[1], blob[0]))
        cpt = p.closest_point(float(mean[0]), float(mean[1]), pose[0], pose[1], blob[0])
        pcm = float(p.prob_color_match(f.mean, f.covar, b))
        pb = p.generate_measurement(1)
        zhat = np.array([pb.bearing, pb.color.r, pb.color.g, pb.color.b], dtype=np.float64)
        H = p.measurement_jacobian(1)
        Q = p.measurement_covariance(H, 1, Qt)
        Qinv = core.inverse(Q)
        K = p.kalman_gain(1, H, Qinv)
        wgt = float(p.importance_factor(Q, b, pb))
        f.update_mean(K, b, pb)
        f.update_covar(K, H)
        rows.append(dict(
            pose=np.array([pose[0], pose[1], hd]), mean=mean, cov=cov, blob=np.array(blob, dtype=np.float64),
            pom=pom, ppm=ppm, closest=np.array(cpt, dtype=np.float64), pcm=pcm, zhat=zhat, H=np.array(H), Q=np.array(Q),
            K=np.array(K), weight=wgt, new_mean=np.array(f.mean, dtype=np.float64), new_cov=np.array(f.covar),
            count=f.update_count,
        ))
    res = {k: np.array([r[k] for r in rows]) for k in rows[0]}
    res["Qt"] = Qt
    return res


# ------------------------------------------------------------------------- scenes
def scene_small():
    means = np.array([
        [6.0, 1.0, 200, 30, 40],
        [4.0, 5.0, 20, 220, 60],
        [-3.0, 6.0, 90, 90, 240],
        [-7.0, -1.0, 250, 250, 10],
        [-2.0, -6.0, 10, 128, 128],
        [5.0, -5.0, 180, 60, 200],
    ], dtype=np.float64)
    covs = np.broadcast_to(0.25 * np.identity(5), (6, 5, 5)).copy()
    # landmark 2 gets a non-trivial block-diagonal covariance
    covs[1, :2, :2] = [[0.4, 0.1], [0.1, 0.3]]
    covs[1, 2:, 2:] = [[3.0, 0.5, -0.2], [0.5, 2.0, 0.3], [-0.2, 0.3, 4.0]]
    immutable = [0, 0, 0, 1, 0, 0]
    P, S = 16, 4
    v, w, dt = 0.2, 0.1, 0.1
    pose = (0.0, 0.0, 0.0)
    steps = []
    for s in range(S):
        pose = truth_step(pose, v, w, dt)
        scan = synthetic_scan(means, pose)
        extra_dup = scan[1].copy()  # second sighting of landmark 2 (sequential double update)
        extra_dup[0] += 0.02
        extra_dup[1:] += [1.0, -1.0, 0.5]
        stray = np.array([1.3, 5.0, 5.0, 5.0])  # matches nothing -> weight *= 0.1
        blobs = np.vstack([scan[:3], stray[None], scan[3:], extra_dup[None]])
        steps.append(blobs)
    return run_steps(P, means, covs, immutable, steps, v, w, [dt] * S, seed=11)


def scene_reference():
    # prkt_ros.py:33-52 (int64 means, Sigma0 = 0.25 I5, all immutable)
    means = [[0, 25, 161, 77, 137], [10, 25, 75, 55, 230], [0, 15, 82, 120, 68], [10, 15, 224, 37, 192]]
    covs = [0.25 * np.identity(5)] * 4
    P, S = 50, 3
    v, w, dt = 0.2, 0.1, 0.1
    pose = (0.0, 0.0, 0.0)
    steps = []
    fm = np.array(means, dtype=np.float64)
    for s in range(S):
        pose = truth_step(pose, v, w, dt)
        scan = synthetic_scan(fm, pose)
        stray = np.array([-2.0, 128.0, 128.0, 128.0])
        steps.append(np.vstack([scan, stray[None]]))
    return run_steps(P, means, covs, [1, 1, 1, 1], steps, v, w, [dt] * S, seed=7)


def scene_config1():
    L, P, S = 50, 100, 3
    means, covs = synthetic_world(L)
    v, w, dt = 0.2, 0.1, 0.1
    pose = (0.0, 0.0, 0.0)
    steps = []
    for s in range(S):
        pose = truth_step(pose, v, w, dt)
        steps.append(synthetic_scan(means, pose))
    lsel = list(range(0, L, 7))
    return run_steps(P, means, covs, [0] * L, steps, v, w, [dt] * S, seed=7, lsel=lsel)


def scene_potential():
    """cam_cb with POTENTIAL features (prkt_core_v2.py:109-118; match_one lists them behind the full features, :366-367).
    The filter itself never fills `potential_features`, so they are put there by hand: slot
    L0 + j of particle i is the reference's potential feature -(L0 + 1 + j), with its own mean per particle (the full
    features' noise-free positions displaced), update_count 0 / 2 / 6 / 4 and the last of them immutable.  A potential feature
    matched by a blob is updated, the particle's weight takes 0.1, and past update_count 5 it moves to feature_set under
    its positive id -- i.e. slot + 1.  (No potential feature is sighted twice in one scan: the reference's loop would look
    a promoted feature up under its old negative id and raise KeyError.)"""
    rs = np.random.RandomState(41)
    full = np.array([
        [6.0, 1.0, 200, 30, 40],
        [4.0, 5.0, 20, 220, 60],
        [-3.0, 6.0, 90, 90, 240],
        [-7.0, -1.0, 250, 250, 10],
        [5.0, -5.0, 180, 60, 200],
    ], dtype=np.float64)
    hidden = np.array([  # true landmarks the potential features stand for
        [-2.0, -6.0, 10, 128, 128],
        [8.0, 4.0, 120, 10, 90],
        [1.0, 9.0, 60, 180, 30],
        [-8.0, 3.0, 30, 60, 120],
    ], dtype=np.float64)
    L0, NP = len(full), len(hidden)
    fcov = np.broadcast_to(0.25 * np.identity(5), (L0, 5, 5)).copy()
    P, S = 12, 3
    v, w, dt = 0.2, 0.1, 0.1
    count0 = np.array([0, 2, 6, 4])          # update_count of the four potential features at the start
    pot_imm = np.array([0, 0, 0, 1])          # the last potential feature is immutable in every particle: updated never, promoted
                                              # never (its count stays), it weighs 0.1 at every sighting
    np.random.seed(13)
    pyrandom.seed(13)
    zstream = np.random.RandomState(13)
    rospy.Time.set_now(0.0)
    fs = build_filter(P, full, fcov, [0, 0, 0, 1, 0])
    pot_mean0 = np.empty((P, NP, 5))
    pot_cov0 = np.empty((P, NP, 5, 5))
    for i, p in enumerate(fs.particles):
        p.potential_features = IterDict()
        for j in range(NP):
            m = hidden[j].copy()
            m[:2] += rs.normal(0, 0.15, 2)      # triangulated somewhere near the truth, differently in every particle
            m[2:] += rs.normal(0, 1.0, 3)
            c = np.identity(5) * rs.uniform(0.5, 1.5)  # add_new_feature starts from the identity (:676-686)
            f = core.Feature(mean=m.copy(), covar=c.copy())
            f.update_count = int(count0[j])
            if pot_imm[j]:
                f.__immutable__ = True
            p.potential_features[-(L0 + 1 + j)] = f
            pot_mean0[i, j] = m
            pot_cov0[i, j] = c
        p.next_id = L0 + NP + 1
    tw = Twist()
    tw.linear.x = v
    tw.angular.z = w
    fs.last_control = tw

    def slots(fs):
        mean = np.empty((P, L0 + NP, 5))
        cov = np.empty((P, L0 + NP, 5, 5))
        cnt = np.empty((P, L0 + NP), dtype=np.int64)
        pot = np.zeros((P, L0 + NP), dtype=bool)
        for i, p in enumerate(fs.particles):
            for sl in range(L0 + NP):
                if (sl + 1) in p.feature_set:
                    f = p.feature_set[sl + 1]
                else:
                    f = p.potential_features[-(sl + 1)]
                    pot[i, sl] = True
                mean[i, sl] = np.asarray(f.mean, dtype=np.float64)
                cov[i, sl] = np.asarray(f.covar, dtype=np.float64)
                cnt[i, sl] = f.update_count
        return mean, cov, cnt, pot

    rec = Recorder()
    out = dict(z=[], post_motion=[], ids=[], weights=[], u=[], ancestors=[], post_resample=[], mean=[], cov=[], count=[], potential=[],
               blobs=[], pre_mean=[], pre_count=[], pre_potential=[])
    pose = (0.0, 0.0, 0.0)
    try:
        for s in range(S):
            pose = truth_step(pose, v, w, dt)
            scan = synthetic_scan(np.vstack([full, hidden]), pose)
            # (every blob matches something: with potential features present an unmatched blob sends the reference into
            # find_nearest_reading, :576-590, which takes the Feature objects for stored readings and raises TypeError)
            dup = scan[1].copy()  # a full feature sighted twice (sequential double update); potential ones only once
            dup[0] += 0.015
            dup[1:] += [0.5, -0.5, 0.25]
            blobs = np.vstack([scan[:4], scan[4:][::-1], dup[None]])
            rospy.Time.advance(dt)
            view = View(blobs)
            n0 = len(rec.normals)
            captured = {}
            orig_motion_update = fs.motion_update

            def mu(tw_, _orig=orig_motion_update, _c=captured):
                _orig(tw_)
                _c["post_motion"] = poses_of(fs)
                _c["ids"] = np.array([[pr[0] for pr in p.match_features_to_scan(view.last_sensor_reading)] for p in fs.particles],
                                     dtype=np.int32)
                _c["pre"] = slots(fs)

            fs.motion_update = mu
            orig_resample = fs.low_variance_resample

            def rs_(_orig=orig_resample, _c=captured):
                _c["weights"] = np.array([float(p.weight) for p in fs.particles])
                _c["post"] = slots(fs)  # before the resample: per particle as it was updated
                for i, p in enumerate(fs.particles):
                    p._golden_index = i
                _orig()
                _c["ancestors"] = np.array([p._golden_index for p in fs.particles], dtype=np.int64)

            fs.low_variance_resample = rs_
            fs.cam_cb(view)
            fs.motion_update = orig_motion_update
            fs.low_variance_resample = orig_resample
            drawn = rec.normals[n0:]
            assert len(drawn) == 3 * P
            z = zstream.standard_normal(3 * P).reshape(P, 3)
            scales = np.array([d[0] for d in drawn]).reshape(P, 3)
            vals = np.array([d[1] for d in drawn]).reshape(P, 3)
            assert np.array_equal(0.0 + scales * z, vals)
            out["z"].append(z)
            out["blobs"].append(blobs)
            out["post_motion"].append(captured["post_motion"])
            out["ids"].append(captured["ids"])
            out["pre_mean"].append(captured["pre"][0])
            out["pre_count"].append(captured["pre"][2])
            out["pre_potential"].append(captured["pre"][3])
            out["weights"].append(captured["weights"])
            out["mean"].append(captured["post"][0])
            out["cov"].append(captured["post"][1])
            out["count"].append(captured["post"][2])
            out["potential"].append(captured["post"][3])
            out["u"].append(rec.uniforms[-1])
            out["ancestors"].append(captured["ancestors"])
            out["post_resample"].append(poses_of(fs))
    finally:
        rec.restore()
    res = {k: np.array(vv) for k, vv in out.items()}
    ids = res["ids"]
    assert (ids < 0).any() and (ids > L0).any(), "the scene must show potential matches and promoted ones"
    res.update(P=P, L0=L0, NP=NP, v=v, w=w, dt=dt, seed=13, full_means=full, full_covs=fcov, full_immutable=np.array([0, 0, 0, 1, 0], dtype=np.uint8),
               pot_mean0=pot_mean0, pot_cov0=pot_cov0, pot_count0=count0, pot_immutable=pot_imm.astype(np.uint8),
               Qt=np.array(fs.Qt))
    return res


def gen_motion():
    """motion_update only (prkt_core_v2.py:148-208), incl. heading wrap through +-pi."""
    P = 32
    seed = 5
    np.random.seed(seed)
    zstream = np.random.RandomState(seed)
    rospy.Time.set_now(0.0)
    fs = build_filter(P, [[1, 1, 1, 1, 1.0]], [np.identity(5)], [0])
    h0 = np.linspace(-3.2, 3.2, P)
    for p, h in zip(fs.particles, h0):
        p.state.pose.pose.orientation = heading_to_quaternion(float(h))
        p.state.pose.pose.position.x = float(h) * 0.5
    start = poses_of(fs)
    controls = [(0.2, 0.1), (1.0, -2.0), (-0.5, 3.0), (0.0, 0.0), (2.0, 0.5)]
    dts = [0.1, 0.25, 0.5, 0.1, 1.0]
    zs, posts = [], []
    for (v, w), dt in zip(controls, dts):
        tw = Twist()
        tw.linear.x = v
        tw.angular.z = w
        fs.last_control = tw  # motion_update moves with last_control (:163)
        rospy.Time.advance(dt)
        fs.motion_update(tw)
        zs.append(zstream.standard_normal(3 * P).reshape(P, 3))
        posts.append(poses_of(fs))
    return dict(start=start, controls=np.array(controls), dts=np.array(dts), z=np.array(zs), post=np.array(posts))


def gen_resample():
    """low_variance_resample (prkt_core_v2.py:210-252) on crafted weight vectors."""
    rs = np.random.RandomState(99)
    cases = []
    P = 50

    def run(weights, seed):
        rospy.Time.set_now(0.0)
        fs = build_filter(len(weights), [[1, 1, 1, 1, 1.0]], [np.identity(5)], [0])
        for i, (p, wgt) in enumerate(zip(fs.particles, weights)):
            p.weight = float(wgt)
            p._golden_index = i
        pyrandom.seed(seed)
        rec = Recorder()
        try:
            fs.low_variance_resample()
        finally:
            rec.restore()
        anc = np.array([p._golden_index for p in fs.particles], dtype=np.int64)
        return rec.uniforms[-1], anc

    specs = [
        ("uniform", np.ones(P)),
        ("zeros", np.zeros(P)),
        ("onehot", np.eye(P)[17]),
        ("random", rs.uniform(0, 1, P)),
        ("lognormal_wide", np.exp(rs.normal(0, 6, P))),
        ("tiny", np.exp(rs.normal(-300, 3, P))),
        ("two_spikes", np.where(np.arange(P) % 25 == 3, 1.0, 1e-12)),
        ("random_1000", rs.uniform(0, 1, 1000) ** 4),
        ("lognormal_1000", np.exp(rs.normal(0, 3, 1000))),
    ]
    out = {}
    for k, (name, wv) in enumerate(specs):
        u, anc = run(wv, seed=100 + k)
        assert len(anc) == len(wv), (name, len(anc))
        out["w_" + name] = np.asarray(wv, dtype=np.float64)
        out["u_" + name] = np.float64(u)
        out["a_" + name] = anc
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    t0 = time.time()
    np.savez_compressed(os.path.join(OUT, "ka_triples.npz"), **gen_triples())
    print("ka_triples done", time.time() - t0)
    np.savez_compressed(os.path.join(OUT, "motion.npz"), **gen_motion())
    np.savez_compressed(os.path.join(OUT, "resample.npz"), **gen_resample())
    print("motion/resample done", time.time() - t0)
    np.savez_compressed(os.path.join(OUT, "step_small.npz"), **scene_small())
    print("step_small done", time.time() - t0)
    np.savez_compressed(os.path.join(OUT, "step_refscene.npz"), **scene_reference())
    print("step_refscene done", time.time() - t0)
    np.savez_compressed(os.path.join(OUT, "step_potential.npz"), **scene_potential())
    print("step_potential done", time.time() - t0)
    if "--skip-config1" not in sys.argv:
        np.savez_compressed(os.path.join(OUT, "step_config1.npz"), **scene_config1())
        print("step_config1 done", time.time() - t0)


if __name__ == "__main__":
    main()
