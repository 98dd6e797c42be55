"""FastSLAM / FilterParticle / Feature: the reference's class surface over the HIP library.

This mirrors ``/root/reference/src/prkt_core_v2.py`` name for name (same constructor
arguments, attributes, method names, argument meaning and error behaviour) so that its
only caller, ``prkt_ros.py`` (left untouched), can ``from parakeet_slam_amd import
FastSLAM, Feature`` instead of ``from prkt_core_v2 import ...`` -- see INTEGRATION.md.

What is different on purpose:
  * all per-particle state lives in HBM behind one ``pk_filter`` handle; ``particles`` is
    a list-like of lazily materialised ``FilterParticle`` snapshots;
  * ``num_particles`` is a constructor keyword (the reference hard-codes 50, :41);
  * one lock serialises ``cam_cb`` / ``motion_update`` / ``summary`` (the reference races
    between the rospy callback thread and the main loop);
  * the particle update itself (motion sample, data association, EKF, weights, resample,
    summary) runs ONLY on the GPU.  There is no CPU fallback: without the shared library
    or without a HIP device, construction raises.
"""
from __future__ import annotations

import copy as _copy
import inspect as _inspect
import math
import random as _pyrandom
import threading

import numpy as np

from . import _lib, msgs
from .msgs import Blob, Odometry, Twist, heading_to_quaternion, quaternion_to_heading

NO_MATCH_WEIGHT = 0.1  # prkt_core_v2.py:851-857


def _default_Qt():
    # prkt_core_v2.py:50-53
    return np.array([[.1, 0, 0, 0], [0, .1, 0, 0], [0, 0, .1, 0], [0, 0, 0, .1]])


def _blob_row(blob):
    """matrix.blob_to_matrix (matrix.py:35-39)."""
    if isinstance(blob, (np.ndarray, list, tuple)):
        return np.asarray(blob, dtype=np.float64).reshape(4)
    return np.array([blob.bearing, blob.color.r, blob.color.g, blob.color.b], dtype=np.float64)


def _blob_matrix(observes):
    """The scan as a (B, 4) array of (bearing, r, g, b) -- blob_to_matrix (matrix.py:35-39) for every blob of
    ``ros_view.last_sensor_reading.observes`` (prkt_core_v2.py:82, :344).  Message objects are read in one flat pass (2 000 blobs:
    0.6 ms of attribute reads instead of 2 ms of one small array per blob); a caller that already holds the scan as a (B, 4)
    array may hand that over as ``observes`` and pays nothing."""
    if isinstance(observes, np.ndarray) and observes.ndim == 2 and observes.shape[1] == 4:
        return np.ascontiguousarray(observes, dtype=np.float64)
    observes = list(observes)
    try:
        flat = [v for b in observes for c in (b.color,) for v in (b.bearing, c.r, c.g, c.b)]
        return np.array(flat, dtype=np.float64).reshape(-1, 4)
    except AttributeError:  # rows given as sequences, or a mix
        return np.array([_blob_row(b) for b in observes], dtype=np.float64).reshape(-1, 4)


def _state_pose(state):
    return (float(state.pose.pose.position.x), float(state.pose.pose.position.y),
            float(quaternion_to_heading(state.pose.pose.orientation)))


def _make_state(x, y, h):
    st = Odometry()
    st.pose.pose.position.x = float(x)
    st.pose.pose.position.y = float(y)
    st.pose.pose.orientation = heading_to_quaternion(float(h))
    return st


# =============================================================================
class Feature(object):
    """prkt_core_v2.py:881-930.  A landmark EKF: mean (x, y, r, g, b), covar 5x5."""

    def __init__(self, mean=None, covar=None):
        self.__immutable__ = False
        if mean is None:
            mean = np.array([0, 0, 0, 0, 0])
        if covar is None:
            covar = np.identity(5, dtype=np.int64)
        self.mean = np.array(mean)
        self.covar = np.array(covar)
        self.identity = np.identity(self.covar.shape[0])
        self.update_count = 0

    # The two methods below are the reference's object-level API (:897-930) for a caller
    # that already holds K and H as matrices.  cam_cb never calls them: inside the filter
    # the same update runs fused in the HIP kernel k_observe.  They are two NumPy products,
    # exactly as matrix.py defines them, kept so scripts that poke a lone Feature still run.
    def update_mean(self, kalman_gain, measure, expected_measure):
        if self.__immutable__:
            return None
        delz = _blob_row(measure) - _blob_row(expected_measure)
        self.mean = self.mean + np.dot(kalman_gain, delz)
        self.update_count += 1

    def update_covar(self, kalman_gain, bigH):
        if self.__immutable__:
            return None
        adjust = np.subtract(self.identity, np.dot(kalman_gain, bigH))
        self.covar = np.dot(adjust, self.covar)
        self.update_count += 1


# =============================================================================
class _FeatureSet(dict):
    """feature_set of a particle that lives on the GPU: downloaded on first touch."""

    def __init__(self, loader):
        super().__init__()
        self._loader = loader

    def _load(self):
        if self._loader is not None:
            ld, self._loader = self._loader, None
            for k, v in ld().items():
                dict.__setitem__(self, k, v)

    def __getitem__(self, k):
        self._load()
        return dict.__getitem__(self, k)

    def __iter__(self):
        self._load()
        return dict.__iter__(self)

    def __len__(self):
        self._load()
        return dict.__len__(self)

    def __contains__(self, k):
        self._load()
        return dict.__contains__(self, k)

    def keys(self):
        self._load()
        return dict.keys(self)

    def values(self):
        self._load()
        return dict.values(self)

    def items(self):
        self._load()
        return dict.items(self)

    def get(self, k, default=None):
        self._load()
        return dict.get(self, k, default)

    def __deepcopy__(self, memo):
        self._load()
        out = {}
        for k, v in dict.items(self):
            out[k] = _copy.deepcopy(v, memo)
        return out


class FilterParticle(object):
    """prkt_core_v2.py:278-877.  A robot pose hypothesis with its own landmark map.

    A standalone instance is a host object; its scalar methods evaluate the device
    functions of the HIP path on the GPU through ``pk_probe`` / a one-particle filter.
    ``FastSLAM.particles[i]`` yields snapshots of the particles that live in HBM.
    """

    Qt = _default_Qt()
    _device = 0

    def __init__(self, state=None):
        if state is None:
            state = _make_state(0.0, 0.0, 0.0)
        self.state = state
        self.feature_set = {}
        self.potential_features = {}
        self.weight = 1
        self.hypothesis_set = {}
        self.next_id = 1

    # -- bookkeeping (:294-315) ----------------------------------------------
    def load_feature_list(self, features):
        for feature in features:
            self.feature_set[self.next_id] = feature
            self.next_id += 1

    def get_feature_by_id(self, id_):
        if id_ < 0:
            return self.potential_features[int(id_)]
        return self.feature_set[id_]

    # -- device evaluation helpers ---------------------------------------------
    def _probe(self, pose, mean, cov, blob, Qt=None):
        return _lib.probe(pose, np.asarray(mean, dtype=np.float64), np.asarray(cov, dtype=np.float64), blob,
                          self.Qt if Qt is None else Qt, device=self._device)

    def _one_particle_filter(self, state):
        if len(self.potential_features):
            raise NotImplementedError(
                "potential (negative-id) features are outside the device path: the reference never "
                "creates one (SURVEY.md section 2, row 1b)")
        ids = sorted(self.feature_set.keys())
        f = _lib.DeviceFilter(1, len(ids), device=self._device)
        if ids:
            means = np.array([np.asarray(self.feature_set[i].mean, dtype=np.float64) for i in ids])
            covs = np.array([np.asarray(self.feature_set[i].covar, dtype=np.float64) for i in ids])
            imm = np.array([bool(self.feature_set[i].__immutable__) for i in ids], dtype=np.uint8)
            f.upload_map(means, covs.reshape(len(ids), 25), imm)
        x, y, h = _state_pose(state)
        f.upload_poses(np.array([[x, y, h, 1.0]]))
        return f, ids

    # -- a3: association (:317-381) ----------------------------------------------
    def match_features_to_scan(self, scan):
        blobs = list(scan.observes)
        if not blobs:
            return []
        f, ids = self._one_particle_filter(self.state)
        try:
            got = f.associate(np.array([_blob_row(b) for b in blobs]))[0]
        finally:
            f.close()
        return [((ids[g - 1] if g > 0 else 0), b) for g, b in zip(got, blobs)]

    def match_one(self, state, blob):
        f, ids = self._one_particle_filter(state)
        try:
            g = int(f.associate(_blob_row(blob)[None, :])[0, 0])
        finally:
            f.close()
        return ids[g - 1] if g > 0 else 0

    # -- a4..a6 (:383-544) ---------------------------------------------------------
    def probability_of_match(self, state, blob, feature):
        r = self._probe(_state_pose(state), feature.mean, feature.covar, _blob_row(blob))
        return float(r["probability_of_match"])

    def prob_position_match(self, f_mean, f_covar, s_x, s_y, bearing):
        cov = np.identity(5)
        cov[:2, :2] = np.asarray(f_covar, dtype=np.float64)[0:2, 0:2]
        mean = np.zeros(5)
        mean[:2] = np.asarray(f_mean, dtype=np.float64)[:2]
        r = self._probe((float(s_x), float(s_y), 0.0), mean, cov, (float(bearing), 0.0, 0.0, 0.0))
        return float(r["prob_position_match"])

    def closest_point(self, f_x, f_y, s_x, s_y, obs_bearing):
        r = self._probe((float(s_x), float(s_y), 0.0), (float(f_x), float(f_y), 0, 0, 0), np.identity(5),
                        (float(obs_bearing), 0.0, 0.0, 0.0))
        return (float(r["closest_point"][0]), float(r["closest_point"][1]))

    def prob_color_match(self, f_mean, f_covar, blob):
        cov = np.identity(5)
        cov[2:, 2:] = np.asarray(f_covar, dtype=np.float64)[2:, 2:]
        z = _blob_row(blob)
        r = self._probe((0.0, 0.0, 0.0), np.asarray(f_mean, dtype=np.float64), cov, z)
        return float(r["prob_color_match"])

    # -- a7..a11 (:748-877) ----------------------------------------------------------
    def _probe_feature(self, feature_id, blob=None, Qt=None):
        f = self.get_feature_by_id(feature_id)
        z = _blob_row(blob) if blob is not None else np.array([0.0, f.mean[2], f.mean[3], f.mean[4]], dtype=np.float64)
        return self._probe(_state_pose(self.state), f.mean, f.covar, z, Qt)

    def generate_measurement(self, featureid):
        r = self._probe_feature(featureid)
        f = self.get_feature_by_id(featureid)
        bobby = Blob()
        bobby.bearing = float(r["zhat"][0])
        bobby.color.r = f.mean[2]
        bobby.color.g = f.mean[3]
        bobby.color.b = f.mean[4]
        return bobby

    def measurement_jacobian(self, feature_id):
        r = self._probe_feature(feature_id)
        H = np.zeros((4, 5))
        H[0, 0], H[0, 1] = r["H0"]
        H[1, 2] = H[2, 3] = H[3, 4] = 1.0
        return H

    def measurement_covariance(self, bigH, feature_id, Qt):
        """Q = H Sigma H' + Qt (:804-819).  H is re-derived on the device from the particle
        state, which is what ``measurement_jacobian`` returned."""
        return self._probe_feature(feature_id, Qt=Qt)["Q"]

    def kalman_gain(self, feature_id, bigH, Qinv):
        """K = Sigma H' Q^-1 (:821-833); Qt is recovered from the caller's Q^-1."""
        Q0 = self._probe_feature(feature_id, Qt=np.zeros((4, 4)))["Q"]
        Qt = np.linalg.inv(np.asarray(Qinv, dtype=np.float64)) - Q0
        return self._probe_feature(feature_id, Qt=Qt)["K"]

    def importance_factor(self, bigQ, blob, pseudoblob):
        """(2 pi ||Q||_F)^-1/2 exp(-1/2 d' Q^-1 d) (:835-849) for caller-supplied matrices.
        Inside the filter this factor is computed by k_observe; this scalar helper exists
        for scripts that call it directly and is a few NumPy flops, as in the reference."""
        v1 = pow(2.0 * math.pi * np.linalg.norm(bigQ), -0.5)
        delz = _blob_row(blob) - _blob_row(pseudoblob)
        return v1 * math.exp(-0.5 * np.dot(np.dot(delz.T, np.linalg.inv(bigQ)), delz))

    def no_match_weight(self):
        return NO_MATCH_WEIGHT

    # -- f4: new-landmark machinery (:546-746) ------------------------------------------------
    # Host bookkeeping per unmatched blob, as in the reference (not part of the data-parallel path: inside the
    # filter the device only applies the 0.1 weight of :95).  Restated with the reference's arithmetic order so
    # that its own unit tests (test_prkt_ros2.py:228-381) hold exactly -- including the behaviour that makes the
    # machinery a dead end there: find_nearest_reading walks ``potential_features`` (never filled by the filter)
    # and returns the nearest entry's key, which is negative or 0, so add_hypothesis always files an orphan.
    def add_hypothesis(self, state, blob):
        """:546-564."""
        pair_id = self.find_nearest_reading(state, blob)
        if pair_id > 0:
            self.add_new_feature(pair_id, state, blob)
        else:
            self.add_orphaned_reading(state, blob)

    def find_nearest_reading(self, state, blob):
        """:566-590: key of the stored reading at the smallest reading distance (first one on ties), 0 if none
        is at a finite distance."""
        best_id, best = 0, float('inf')
        for id_, reading in self.potential_features.items():
            d = self.reading_distance_function(reading[0], reading[1], state, blob)
            if d < best:
                best, best_id = d, id_
        return best_id

    def reading_distance_function(self, state1, blob1, state2, blob2):
        """:592-609: colour distance of two readings whose rays cross, else inf."""
        x1, y1, h1 = _state_pose(state1)
        x2, y2, h2 = _state_pose(state2)
        if not self.ray_intersect(x1, y1, blob1.bearing + h1, x2, y2, blob2.bearing + h2):
            return float('inf')
        return self.color_distance(blob1, blob2)

    def ray_intersect(self, x1, y1, b1, x3, y3, b3):
        """:611-643: do the half-lines (x1, y1, b1) and (x3, y3, b3) meet (both ray parameters >= 0)."""
        ax, ay = math.cos(b1), math.sin(b1)
        bx, by = math.cos(b3), math.sin(b3)
        cross = ay * bx - ax * by
        if cross == 0:
            return False
        v = (ax * y3 - ay * x3 + ay * x1 - ax * y1) / cross
        if abs(ay) < abs(ax):
            u = (x3 + bx * v - x1) / ax
        else:
            u = (y3 + by * v - y1) / ay
        return u >= 0 and v >= 0

    def color_distance(self, blob1, blob2):
        """:645-654: Euclidean distance of the two colours."""
        return math.sqrt(math.pow(blob1.color.r - blob2.color.r, 2) + math.pow(blob1.color.g - blob2.color.g, 2) +
                         math.pow(blob1.color.b - blob2.color.b, 2))

    def add_new_feature(self, old_id, state, blob):
        """:656-686: a potential feature (negative id) where the stored reading old_id and this one cross, mean
        colour of the two, identity covariance."""
        old_state, old_blob = self.hypothesis_set[old_id]
        x, y = self.cross_readings((old_state, old_blob), (state, blob))
        mean = np.array([x, y, (old_blob.color.r + blob.color.r) / 2, (old_blob.color.g + blob.color.g) / 2,
                         (old_blob.color.b + blob.color.b) / 2])
        self.potential_features[-self.next_id] = Feature(mean=mean, covar=np.identity(5))
        self.next_id += 1

    def cross_readings(self, old_reading, new_reading):
        """:688-737: intersection of the two LINES through the readings (also behind the observers), None when
        they are parallel.  Line-line intersection through two points each, one unit step along the ray apart."""
        x1, y1, h1 = _state_pose(old_reading[0])
        x3, y3, h3 = _state_pose(new_reading[0])
        return self._cross_lines(x1, y1, h1 + old_reading[1].bearing, x3, y3, h3 + new_reading[1].bearing)

    @staticmethod
    def _cross_lines(x1, y1, h1, x3, y3, h3):
        """The arithmetic of cross_readings on plain numbers (world-frame bearings h1, h3)."""
        x2, y2 = x1 + math.cos(h1), y1 + math.sin(h1)
        x4, y4 = x3 + math.cos(h3), y3 + math.sin(h3)
        d12, d34 = x1 * y2 - y1 * x2, x3 * y4 - x4 * y3
        den = (x1 - x2) * (y3 - y4) - (y1 - y2) * (x3 - x4)
        if den == 0:
            return None
        return ((d12 * (x3 - x4) - (x1 - x2) * d34) / den, (d12 * (y3 - y4) - (y1 - y2) * d34) / den)

    def add_orphaned_reading(self, state, blob):
        """:739-746."""
        self.hypothesis_set[self.next_id] = ((state, blob,))
        self.next_id += 1



# ----------------------------------------------------------------------------- new-landmark bookkeeping: arrays <-> lists
def _nl_unpack(cnt, rd, sid, L0):
    """pk_grow_download's arrays of n particles as the host lists of the growing mode."""
    n = cnt.shape[0]
    return dict(
        hyp=[[(int(r[0]),) + tuple(float(v) for v in r[1:]) for r in rd[i, :cnt[i, 0]]] for i in range(n)],
        next_id=[int(v) for v in cnt[:, 2]], used=[int(v) for v in cnt[:, 1]],
        slot_id=[{L0 + k: int(sid[i, k]) for k in range(cnt[i, 1])} for i in range(n)],
        dropped=int(cnt[:, 3].sum()))


def _nl_pack(hyp, next_id, used, slot_id, L0, S, R):
    """... and back: (counters (n, 4), readings (n, R, 8), slot ids (n, S)) for pk_grow_upload."""
    n = len(hyp)
    cnt = np.zeros((n, 4), dtype=np.int32)
    rd = np.zeros((n, R, 8))
    sid = np.zeros((n, S), dtype=np.int32)
    for i in range(n):
        cnt[i, :3] = (len(hyp[i]), used[i], next_id[i])
        if hyp[i]:
            rd[i, :len(hyp[i])] = np.asarray(hyp[i], dtype=np.float64)
        for slot, fid in slot_id[i].items():
            sid[i, slot - L0] = fid
    return cnt, rd, sid


def _nl_snapshot(nl):
    """The bookkeeping as plain numeric arrays (a snapshot never needs pickle to load): every particle's orphaned readings
    (8 numbers each) back to back with per-particle offsets, the id counters, and the (particle, slot, id) triples of the
    spare slots in use."""
    offs = np.zeros(len(nl["hyp"]) + 1, dtype=np.int64)
    offs[1:] = np.cumsum([len(h) for h in nl["hyp"]])
    flat = [rd for h in nl["hyp"] for rd in h]
    return dict(
        nl_readings=np.asarray(flat, dtype=np.float64).reshape(len(flat), 8),
        nl_offsets=offs,
        nl_next_id=np.asarray(nl["next_id"], dtype=np.int64),
        nl_used=np.asarray(nl["used"], dtype=np.int64),
        nl_slot_id=np.asarray([(i, s, v) for i, d in enumerate(nl["slot_id"]) for s, v in sorted(d.items())],
                              dtype=np.int64).reshape(-1, 3))


def _nl_snapshot_arrays(cnt, rd, sid, L0):
    """_nl_snapshot straight from pk_grow_download's arrays (no per-particle Python objects: a snapshot of 10^5 particles)."""
    n, used = cnt[:, 0].astype(np.int64), cnt[:, 1].astype(np.int64)
    offs = np.zeros(cnt.shape[0] + 1, dtype=np.int64)
    offs[1:] = np.cumsum(n)
    keep = np.arange(rd.shape[1])[None, :] < n[:, None]
    in_use = np.arange(sid.shape[1])[None, :] < used[:, None]
    ii, kk = np.nonzero(in_use)
    return dict(nl_readings=np.ascontiguousarray(rd[keep], dtype=np.float64).reshape(-1, 8), nl_offsets=offs,
                nl_next_id=cnt[:, 2].astype(np.int64), nl_used=used,
                nl_slot_id=np.stack([ii, L0 + kk, sid[in_use].astype(np.int64)], axis=1).astype(np.int64).reshape(-1, 3))


def _nl_arrays_from_snapshot(d, P, L0, S, R):
    """... and back: (counters, readings, slot ids) for pk_grow_upload from a CHECKED snapshot."""
    offs = np.asarray(d["nl_offsets"], dtype=np.int64)
    n = np.diff(offs)
    cnt = np.zeros((P, 4), dtype=np.int32)
    cnt[:, 0], cnt[:, 1], cnt[:, 2] = n, d["nl_used"], d["nl_next_id"]
    rd = np.zeros((P, R, 8))
    rd[np.arange(R)[None, :] < n[:, None]] = np.asarray(d["nl_readings"], dtype=np.float64).reshape(-1, 8)
    sid = np.zeros((P, S), dtype=np.int32)
    t = np.asarray(d["nl_slot_id"], dtype=np.int64).reshape(-1, 3)
    sid[t[:, 0], t[:, 1] - L0] = t[:, 2]
    return cnt, rd, sid


def _nl_check_snapshot(d, P, L0, L, spare, ring=None):
    """Raises ValueError unless d's nl_* arrays fit a filter of P particles, landmarks [L0, L) spare, rings of `ring` readings."""
    offs, rd = d["nl_offsets"], d["nl_readings"]
    used, slot_id = d["nl_used"], d["nl_slot_id"]
    if (offs.shape != (P + 1,) or offs[0] != 0 or np.any(np.diff(offs) < 0) or rd.shape != (int(offs[-1]), 8)
            or d["nl_next_id"].shape != (P,) or used.shape != (P,) or slot_id.ndim != 2 or slot_id.shape[1] != 3):
        raise ValueError("snapshot: malformed new-landmark bookkeeping")
    if (np.any(used < 0) or np.any(used > spare) or np.any(slot_id[:, 0] < 0) or np.any(slot_id[:, 0] >= P)
            or np.any(slot_id[:, 1] < L0) or np.any(slot_id[:, 1] >= L)):
        raise ValueError("snapshot: new-landmark bookkeeping names particles / spare slots this filter does not have")
    if ring is not None and P and int(np.diff(offs).max()) > ring:
        raise ValueError("snapshot: a particle holds %d orphaned readings, this filter keeps %d (reading_capacity)"
                         % (int(np.diff(offs).max()), ring))


def _nl_restore(d, P):
    """(hyp, next_id, used, slot_id) lists from a checked snapshot."""
    offs, rd = d["nl_offsets"], d["nl_readings"]

    def reading(r):  # (id, x, y, heading, bearing, r, g, b): the id is an integer
        return (int(r[0]),) + tuple(float(v) for v in r[1:])

    slot_ids = [dict() for _ in range(P)]
    for i, slot, fid in d["nl_slot_id"]:
        slot_ids[int(i)][int(slot)] = int(fid)
    return ([[reading(r) for r in rd[offs[i]:offs[i + 1]]] for i in range(P)], [int(v) for v in d["nl_next_id"]],
            [int(v) for v in d["nl_used"]], slot_ids)


# =============================================================================
class _ParticleList(list):
    """``FastSLAM.particles``: list-like view of the particles held in HBM."""

    def __init__(self, owner):
        super().__init__()
        self._o = owner

    def __len__(self):
        return self._o.num_particles

    def __iter__(self):
        for i in range(len(self)):
            yield self._o._particle_view(i)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self._o._particle_view(j) for j in range(*i.indices(len(self)))]
        n = len(self)
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError("particle index out of range")
        return self._o._particle_view(i)

    def __setitem__(self, i, particle):
        n = len(self)
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError("particle index out of range")
        self._o._store_particle(i, particle)

    def __repr__(self):
        return "<%d particles on GPU %d>" % (len(self), self._o._device)


class FastSLAM(object):
    """prkt_core_v2.py:37-276 on an MI355X.

    FastSLAM(preset_features=[], num_particles=50, device=0, weight_domain="linear",
             rng="global", seed=0, publish_debug=None)

    rng="global": motion noise from ``numpy.random`` and the resample draw from
    ``random.random()`` -- the reference's own streams, so ``np.random.seed(s);
    random.seed(s)`` reproduces the reference bit for bit in the draws;
    rng="device": Philox noise generated on the GPU (no host->device upload).

    new_landmarks=True (with spare_landmarks=S slots per particle): the new-landmark initialisation of :546-746 and the
    potential-feature rule of :109-118, made to work (in the reference ``find_nearest_reading`` walks
    ``potential_features`` where the readings are in ``hypothesis_set``, so nothing is ever paired): an unmatched blob is
    paired with the nearest stored reading of the particle -- rays crossing, colour distance below ``pair_threshold`` --
    and becomes a potential feature at the crossing, else it is stored as an orphaned reading.  Matching, the EKF
    update, the 0.1 weight of a potential feature and its promotion past update_count 5 run in the device kernels (a
    flag in the landmark's count word).  The per-unmatched-blob bookkeeping runs on the device too (bookkeeping="device",
    the default: one kernel behind every observe, pk_grow_enable; each particle keeps up to ``reading_capacity`` orphaned
    readings in HBM, ``readings_dropped()`` counts what did not fit) -- nothing per particle crosses to the host in a step;
    bookkeeping="host" is the per-particle host loop of the reference (and the only one on the dense layout).
    ``record_ids`` (default: only with bookkeeping="host"): keep ``last_ids`` / ``last_ancestors`` of every step.
    """

    EMPTY_COLOUR = 2.0 ** 100  # colour of a spare slot that holds nothing yet: fails every colour gate (:441), exact in float32

    def __new__(cls, *args, **kwargs):
        # FastSLAM(preset_features, devices=[0, 1, ...]): the same class surface with the particles sharded over several
        # GPUs, one child process per device (multi.py; SURVEY section 5's `devices=` keyword).  The arguments are bound
        # against FastSLAM's OWN signature -- positional ones included -- and forwarded with FastSLAM's defaults (weight_domain
        # "linear", rng "global"); what the sharded facade cannot do is refused, not dropped.
        extra = {k: kwargs.pop(k) for k in ("backend", "_shard_factory") if k in kwargs}
        try:
            ba = _inspect.signature(cls.__init__).bind(None, *args, **kwargs)
        except TypeError:
            return super(FastSLAM, cls).__new__(cls)  # __init__ raises the same TypeError with its own wording
        ba.apply_defaults()
        a = ba.arguments
        devices = a.get("devices")
        if devices is not None and len(list(devices)) > 1:
            from .multi import ShardedFastSLAM

            if a["new_landmarks"] and a["bookkeeping"] != "device":
                raise ValueError("FastSLAM(devices=[...]): the new-landmark bookkeeping of several GPUs lives on the devices "
                                 "(bookkeeping='device'): host lists cannot follow a particle from one GPU to another")
            if a["device"] not in (0, list(devices)[0]):
                raise ValueError("FastSLAM: give either device= or devices=, not both")
            return ShardedFastSLAM(a["preset_features"], num_particles=a["num_particles"], devices=list(devices),
                                   weight_domain=a["weight_domain"], rng=a["rng"], seed=a["seed"],
                                   publish_debug=a["publish_debug"], new_landmarks=a["new_landmarks"],
                                   spare_landmarks=a["spare_landmarks"], pair_threshold=a["pair_threshold"],
                                   reading_capacity=a["reading_capacity"], **extra)
        if extra:
            raise TypeError("FastSLAM: %s only apply with devices=[...] naming several GPUs" % ", ".join(sorted(extra)))
        return super(FastSLAM, cls).__new__(cls)

    def __init__(self, preset_features=[], num_particles=50, device=0, weight_domain="linear", rng="global",
                 seed=0, publish_debug=None, new_landmarks=False, spare_landmarks=0, pair_threshold=30.0, devices=None,
                 bookkeeping="device", reading_capacity=64, record_ids=None):
        if devices is not None and len(list(devices)) == 1:
            device = list(devices)[0]
        self._lock = threading.RLock()
        self.last_control = Twist()
        self.last_update = msgs.now()
        self.num_particles = int(num_particles)
        self._device = int(device)
        self._features = list(preset_features)
        self._ids = list(range(1, len(self._features) + 1))  # load_feature_list :294-299
        L = len(self._features)
        self._grow = bool(new_landmarks)
        self._spare = int(spare_landmarks) if self._grow else 0
        self._pair_threshold = float(pair_threshold)
        self._L0 = L
        self._filter = _lib.DeviceFilter(self.num_particles, L + self._spare, device=self._device)
        if L + self._spare:
            means = np.zeros((L + self._spare, 5))
            covs = np.tile(np.identity(5).reshape(25), (L + self._spare, 1))
            imm = np.zeros(L + self._spare, dtype=np.uint8)
            if L:
                means[:L] = np.array([np.asarray(f.mean, dtype=np.float64).reshape(5) for f in self._features])
                covs[:L] = np.array([np.asarray(f.covar, dtype=np.float64).reshape(25) for f in self._features])
                imm[:L] = np.array([bool(f.__immutable__) for f in self._features], dtype=np.uint8)
            means[L:, 2:] = self.EMPTY_COLOUR
            self._filter.upload_map(means, covs, imm)
        P = self.num_particles
        if bookkeeping not in ("device", "host"):
            raise ValueError("bookkeeping must be 'device' or 'host'")
        # the per-particle bookkeeping of the growing mode (follows the particles through every resample): on the device
        # (pk_grow_enable; the four views below download it when asked) or in host lists
        self._nl_device = self._grow and self._spare > 0 and bookkeeping == "device"
        self._nl_cache = None
        if self._nl_device:
            try:
                self._filter.grow_enable(L, int(reading_capacity), self._pair_threshold)
            except _lib.PkError as e:
                if e.status != _lib.PK_ERR_STATE:
                    raise
                self._filter.close()
                raise ValueError("%s -- FastSLAM(..., bookkeeping='host') keeps the new-landmark bookkeeping on the host, on any layout" % e)
        else:
            self._nl_host = dict(
                hyp=[[] for _ in range(P)],          # orphaned readings: (id, x, y, heading, bearing, r, g, b)  (:739-746)
                next_id=[L + 1] * P,                 # FilterParticle.next_id (:298)
                used=[0] * P,                        # spare slots in use
                slot_id=[dict() for _ in range(P)])  # spare slot -> feature id
        self.record_ids = (not self._nl_device) if record_ids is None else bool(record_ids)
        self.Qt = _default_Qt()
        self._domain = {"linear": _lib.PK_WEIGHTS_LINEAR, "log": _lib.PK_WEIGHTS_LOG}[weight_domain]
        if rng not in ("global", "device"):
            raise ValueError("rng must be 'global' or 'device'")
        self._rng = rng
        self._seed = int(seed)
        self._draw = 0
        self._gen = 0
        self._pose_cache = None
        self._probe = None
        self.particles = _ParticleList(self)
        self.last_ids = None
        r = msgs.ros()
        self._publish = (r is not None) if publish_debug is None else bool(publish_debug)
        if r is not None:
            self.aged_particles_pub = r.Publisher('/aged_particles', Odometry, queue_size=1)
            self.resampled_particles_pub = r.Publisher('/resampled_particles', Odometry, queue_size=1)
            self.particle_track_pub = r.Publisher('/particle_track', Odometry, queue_size=1)
        else:
            self.aged_particles_pub = self.resampled_particles_pub = self.particle_track_pub = None
            self._publish = False

    # ------------------------------------------------------------------ views
    def _touch(self):
        self._gen += 1
        self._pose_cache = None
        self._nl_cache = None

    def _nl(self):
        """The new-landmark bookkeeping of every particle as host lists (device mode: one download, kept until the next change)."""
        if not self._nl_device:
            return self._nl_host
        if self._nl_cache is None:
            self._nl_cache = _nl_unpack(*self._filter.grow_download(), L0=self._L0)
        return self._nl_cache

    _hyp = property(lambda self: self._nl()["hyp"])
    _next_id = property(lambda self: self._nl()["next_id"])
    _used = property(lambda self: self._nl()["used"])
    _slot_id = property(lambda self: self._nl()["slot_id"])

    def _nl_particle(self, i):
        """(next_id, readings, spare slot -> id) of particle i."""
        if not self._nl_device:
            h = self._nl_host
            return h["next_id"][i], h["hyp"][i], dict(h["slot_id"][i])
        one = _nl_unpack(*self._filter.grow_download(i, i + 1), L0=self._L0)
        return one["next_id"][0], one["hyp"][0], one["slot_id"][0]

    def _nl_assign(self, hyp, next_id, used, slot_id):
        """Replace the whole bookkeeping (load_state)."""
        if not self._nl_device:
            self._nl_host = dict(hyp=hyp, next_id=next_id, used=used, slot_id=slot_id)
            return
        P = self.num_particles
        _, S, R = self._filter.grow_shape()
        cnt, rd, sid = _nl_pack(hyp, next_id, used, slot_id, self._L0, S, R)
        self._filter.grow_upload(0, P, cnt, rd, sid)
        self._nl_cache = None

    def readings_dropped(self):
        """Orphaned readings that found the particle's ring full (bookkeeping="device"; the reference's dict grows without
        bound): 0 means the device bookkeeping is the reference's."""
        if not self._nl_device:
            return 0
        return int(self._filter.grow_download(readings=False, slot_ids=False)[0][:, 3].sum())

    def _poses(self):
        if self._pose_cache is None:
            self._pose_cache = self._filter.download_poses()
        return self._pose_cache

    def _particle_view(self, i):
        with self._lock:
            x, y, h, w = self._poses()[i]
            p = FilterParticle(_make_state(x, y, h))
            p.weight = float(w)
            p.Qt = self.Qt
            p._device = self._device
            p.next_id, hyp_i, slot_id = self._nl_particle(i)
            filt, ids, feats, lock = self._filter, self._ids, self._features, self._lock
            if self._grow:
                for rd in hyp_i:
                    b = Blob()
                    b.bearing = rd[4]
                    b.color.r, b.color.g, b.color.b = rd[5], rd[6], rd[7]
                    p.hypothesis_set[rd[0]] = (_make_state(rd[1], rd[2], rd[3]), b)

            def load_all():
                with lock:
                    m, c, k = filt.download_landmarks(i, i + 1)
                full, potential = {}, {}
                for j, id_ in enumerate(ids):
                    f = Feature(mean=m[0, j], covar=c[0, j])
                    f.update_count = int(k[0, j]) & ~_lib.PK_LANDMARK_POTENTIAL
                    f.__immutable__ = bool(feats[j].__immutable__)
                    full[id_] = f
                for slot, id_ in slot_id.items():
                    f = Feature(mean=m[0, slot], covar=c[0, slot])
                    f.update_count = int(k[0, slot]) & ~_lib.PK_LANDMARK_POTENTIAL
                    if int(k[0, slot]) & _lib.PK_LANDMARK_POTENTIAL:
                        potential[-id_] = f  # :685
                    else:
                        full[id_] = f        # promoted (:115-116)
                return full, potential

            p.feature_set = _FeatureSet(lambda: load_all()[0])
            if slot_id:
                p.potential_features = load_all()[1]
            return p

    def _store_particle(self, i, particle):
        """``fs.particles[i] = particle`` (the reference does this at :162): write the pose,
        weight and landmark estimates of a host particle into slot i."""
        with self._lock:
            x, y, h = _state_pose(particle.state)
            self._filter.upload_pose(i, (x, y, h, float(particle.weight)))  # (the other particles' log-weights stay as they are)
            L = len(self._ids)
            if L and all(k in particle.feature_set for k in self._ids):
                fs_ = [particle.feature_set[k] for k in self._ids]
                means = np.array([np.asarray(f.mean, dtype=np.float64) for f in fs_]).reshape(1, L, 5)
                covs = np.array([np.asarray(f.covar, dtype=np.float64) for f in fs_]).reshape(1, L, 25)
                cnts = np.array([int(f.update_count) for f in fs_], dtype=np.int32).reshape(1, L)
                if self._spare:  # the preset landmarks only: the spare slots keep what the filter put there
                    m, c, k = self._filter.download_landmarks(i, i + 1)
                    m[:, :L], k[:, :L] = means, cnts
                    c = c.reshape(1, -1, 25)
                    c[:, :L] = covs
                    means, covs, cnts = m, c, k
                self._filter.upload_landmarks(i, i + 1, means, covs, cnts)
            self._touch()

    def _publish_all(self, pub, poses):
        if not self._publish or pub is None:
            return
        for x, y, h, _w in poses:
            st = _make_state(x, y, h)
            st.header.frame_id = 'odom'
            pub.publish(st)

    # ------------------------------------------------------------------ a12
    def cam_cb(self, ros_view):
        """One filter step (:59-137)."""
        with self._lock:
            self._motion_update(self.last_control)  # :75-77
            scan = ros_view.last_sensor_reading  # :82
            blobs = _blob_matrix(scan.observes)
            self._filter.set_measurement_noise(self.Qt)
            # :73 weight = 1 (the reset is fused into the observe kernels: pk_observe_fresh; the motion
            # update in between does not read the weights), :84-124 association + EKF + weights
            if self._grow and len(blobs) and not self._nl_device:
                self.last_ids = self._filter.observe(blobs, fresh=True, return_ids=True)
                self._touch()
                self._new_landmarks(blobs, self.last_ids)  # :92-95 for every particle's unmatched blobs
            elif self._nl_device and len(blobs):
                # :92-95 on the device, behind the association and the updates (k_new_landmarks)
                self.last_ids = self._filter.observe(blobs, fresh=True, return_ids=self.record_ids)
            else:
                self._filter.observe(blobs, fresh=True)
            self._touch()
            if self._publish:
                self._publish_all(self.particle_track_pub, self._poses())  # :126-127
            self.low_variance_resample()  # :137
            self._steps_done = getattr(self, "_steps_done", 0) + 1
            if self._nl_device and not getattr(self, "_warned_dropped", False) and self._steps_done % 16 == 0:
                # (the reference's hypothesis_set grows without bound; the device ring holds reading_capacity orphaned
                # readings per particle and counts what it has to drop -- say so once, not only through readings_dropped())
                n = self.readings_dropped()
                if n:
                    import warnings

                    warnings.warn("FastSLAM(new_landmarks=True): %d orphaned readings found their particle's ring full and were dropped "
                                  "(reading_capacity=%d per particle; the reference keeps every reading): later unknown landmarks may not be "
                                  "triangulated -- raise reading_capacity, or use bookkeeping='host'" % (n, self._filter.grow_shape()[2]),
                                  RuntimeWarning, stacklevel=2)
                    self._warned_dropped = True

    def _new_landmarks(self, blobs, ids):
        """add_hypothesis (:546-564) for every unmatched blob of every particle, in scan order, with the working pairing
        rule (class docstring).  New potential features are written into the particle's next spare slot."""
        poses = self._poses()
        scratch = FilterParticle()
        for i in np.nonzero((ids == 0).any(axis=1))[0]:
            i = int(i)
            x, y, h = float(poses[i, 0]), float(poses[i, 1]), float(poses[i, 2])
            fresh = []
            for b in np.nonzero(ids[i] == 0)[0]:
                z = blobs[b]
                best, best_d = None, float('inf')
                for rd in self._hyp[i]:  # find_nearest_reading :566-590 over the stored readings
                    if not scratch.ray_intersect(rd[1], rd[2], rd[4] + rd[3], x, y, float(z[0]) + h):  # :592-609
                        continue
                    d = math.sqrt(math.pow(rd[5] - z[1], 2) + math.pow(rd[6] - z[2], 2) + math.pow(rd[7] - z[3], 2))
                    if d < best_d:
                        best, best_d = rd, d
                xy = None
                if best is not None and best_d < self._pair_threshold and self._used[i] + len(fresh) < self._spare:
                    xy = scratch._cross_lines(best[1], best[2], best[3] + best[4], x, y, h + float(z[0]))  # :688-737
                if xy is not None:  # add_new_feature :656-686
                    slot = self._L0 + self._used[i] + len(fresh)
                    fresh.append((slot, (xy[0], xy[1], (best[5] + z[1]) / 2, (best[6] + z[2]) / 2, (best[7] + z[3]) / 2)))
                    self._slot_id[i][slot] = self._next_id[i]
                else:  # add_orphaned_reading :739-746
                    self._hyp[i].append((self._next_id[i], x, y, h, float(z[0]), float(z[1]), float(z[2]), float(z[3])))
                self._next_id[i] += 1
            if fresh:
                m, c, k = self._filter.download_landmarks(i, i + 1)
                for slot, mean in fresh:
                    m[0, slot] = mean
                    c[0, slot] = np.identity(5)
                    k[0, slot] = _lib.PK_LANDMARK_POTENTIAL  # update_count 0, potential
                self._filter.upload_landmarks(i, i + 1, m, c.reshape(1, -1, 25), k)
                self._used[i] += len(fresh)

    def odom_motion_update(self, odom):
        pass  # :140-146 alpha feature, empty in the reference

    # ------------------------------------------------------------------ a2
    def _noise(self, n):
        if self._rng == "global":
            # numpy.random.normal(0, s, 1) x 3 per particle, particle-major (:185-193), is the
            # same stream as one standard_normal(3 P) call scaled by s (legacy loc + scale*gauss)
            return np.random.standard_normal((n, 3))
        return None

    def _motion_update(self, new_twist):
        dt = msgs.now() - self.last_update  # :158
        v = float(self.last_control.linear.x)  # moves with the PREVIOUS control (:163)
        w = float(self.last_control.angular.z)
        self._filter.motion(v, w, dt.to_sec(), z=self._noise(self.num_particles), seed=self._seed, draw=self._draw)
        self._draw += 1
        self._touch()
        self.last_update = self.last_update + dt  # :165
        self.last_control = new_twist  # :166

    def motion_update(self, new_twist):
        with self._lock:
            self._motion_update(new_twist)

    def motion_model(self, particle, twist, dt):
        """Move ONE particle (:168-208) and return the moved copy.  Runs the same kernel on a
        one-particle filter."""
        dt = dt.to_sec()
        x, y, h = _state_pose(particle.state)
        with self._lock:
            if self._probe is None:  # one one-particle filter per FastSLAM, made on first use (a stream + buffers: ~ms to create)
                self._probe = _lib.DeviceFilter(1, 0, device=self._device)
            f = self._probe
            f.upload_poses(np.array([[x, y, h, 1.0]]))
            z = self._noise(1)
            f.motion(float(twist.linear.x), float(twist.angular.z), dt, z=z, seed=self._seed, draw=self._draw)
            self._draw += 1
            nx, ny, nh, _ = f.download_poses()[0]
        new_particle = _copy.deepcopy(particle)
        new_particle.state = _copy.deepcopy(particle.state)
        new_particle.state.pose.pose.position.x = nx
        new_particle.state.pose.pose.position.y = ny
        new_particle.state.pose.pose.orientation = heading_to_quaternion(nh)
        return new_particle

    # ------------------------------------------------------------------ a13
    def low_variance_resample(self):
        with self._lock:
            if self._publish:
                self._publish_all(self.aged_particles_pub, self._poses())  # :237
            u = _pyrandom.random()  # :226
            host_nl = self._grow and not self._nl_device
            self.last_ancestors = self._filter.resample(u, domain=self._domain,
                                                        return_ancestors=self._publish or host_nl or (self._grow and self.record_ids))
            if host_nl:  # (device mode: pk_resample gathers the bookkeeping with the particles)
                anc = [int(a) for a in self.last_ancestors]
                h = self._nl_host
                self._nl_host = dict(hyp=[list(h["hyp"][a]) for a in anc], next_id=[h["next_id"][a] for a in anc],
                                     used=[h["used"][a] for a in anc], slot_id=[dict(h["slot_id"][a]) for a in anc])
            self._touch()
            if self._publish:
                self._publish_all(self.resampled_particles_pub, self._poses())  # :242

    # ------------------------------------------------------------------ a14
    def summary(self):
        with self._lock:
            return self._filter.summary()

    # ------------------------------------------------------------------ snapshot / restore
    def save_state(self, path):
        """Whole filter state -> .npz (poses, weights, every particle's landmark means /
        covariances / update counts, Qt, controls).  The reference has no checkpointing
        (SURVEY section 5); this is the state round trip tests and long runs want."""
        with self._lock:
            poses = self._filter.download_poses()
            m, c, k = self._filter.download_landmarks()
            extra = {}
            if self._nl_device:
                extra = _nl_snapshot_arrays(*self._filter.grow_download(), L0=self._L0)
            elif self._grow:
                extra = _nl_snapshot(self._nl())
            np.savez_compressed(
                path, poses=poses, means=m, covs=c, counts=k, Qt=np.asarray(self.Qt, dtype=np.float64),
                immutable=np.array([bool(f.__immutable__) for f in self._features], dtype=np.uint8),
                last_control=np.array([float(self.last_control.linear.x), float(self.last_control.angular.z)]),
                last_update=float(self.last_update.to_sec()), draw=self._draw, **extra)

    def load_state(self, path):
        with self._lock:
            d = np.load(path, allow_pickle=False)  # numeric arrays only: loading a snapshot never runs code
            P, L = self.num_particles, self._L0 + self._spare
            if d["poses"].shape != (P, 4) or d["means"].shape != (P, L, 5):
                raise ValueError("snapshot is for %s particles x %s landmarks, this filter has %d x %d"
                                 % (d["poses"].shape[0], d["means"].shape[1], P, L))
            if self._grow and "nl_offsets" not in d.files:
                # (snapshots of before round 3 kept the bookkeeping as a pickled object array "new_landmarks": not read any
                # more -- and a growing filter whose maps are replaced while its bookkeeping stays is inconsistent)
                raise ValueError("snapshot: no new-landmark bookkeeping (nl_* arrays) for a filter with new_landmarks=True%s"
                                 % ("; it was written in the old pickled format" if "new_landmarks" in d.files else ""))
            have_nl = "nl_offsets" in d.files and self._grow
            if have_nl:  # everything is checked before anything is assigned
                _nl_check_snapshot(d, P, self._L0, L, self._spare, self._filter.grow_shape()[2] if self._nl_device else None)

            self._filter.upload_poses(d["poses"])
            if L:
                self._filter.upload_landmarks(0, P, d["means"], d["covs"].reshape(P, L, 25), d["counts"])
            self.Qt = d["Qt"].copy()
            self.last_control.linear.x = float(d["last_control"][0])
            self.last_control.angular.z = float(d["last_control"][1])
            self._draw = int(d["draw"])
            if have_nl and self._nl_device:
                _, S, R = self._filter.grow_shape()
                self._filter.grow_upload(0, P, *_nl_arrays_from_snapshot(d, P, self._L0, S, R))
            elif have_nl:
                self._nl_assign(*_nl_restore(d, P))
            self._touch()

    def close(self):
        if self._probe is not None:
            self._probe.close()
            self._probe = None
        self._filter.close()
