"""NumPy restatement of parakeet_slam's per-timestep particle update.

TEST INFRASTRUCTURE ONLY -- this is the *oracle*, not the product.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it.  The shipped path is the HIP library behind
``include/parakeet_slam.h``; it never calls into this file.

Every function cites the reference lines (``/root/reference/src/...``) it
restates.  State is dense float64 struct-of-arrays over particles so that the
same code doubles as a vectorised CPU baseline:

    x, y, h     (P,)        pose; heading wrapped to (-pi, pi] the way the
                            reference's quaternion round trip wraps it
    logw        (P,)        natural log of the particle weight (the reference
                            keeps the linear product, prkt_core_v2.py:95,124;
                            ``weights()`` returns exp(logw))
    mean        (P, L, 5)   landmark means  (x, y, r, g, b)
    cov         (P, L, 5,5) landmark covariances (dense, like the reference)
    count       (P, L)      Feature.update_count (prkt_core_v2.py:914,930)
    immutable   (L,)        Feature.__immutable__ (prkt_core_v2.py:883,909,926)

Pinned by: ``tests/test_oracle_vs_golden.py`` against golden vectors captured
from the unmodified reference by ``oracle/make_golden.py`` (NumPy 2.2.6 /
SciPy 1.15.3), and by the known-answer assertions of the reference's own
``src/test_prkt_ros2.py`` restated in ``tests/test_reference_known_answers.py``.
"""
from __future__ import annotations

import math

import numpy as np

TWO_PI = 2.0 * math.pi
NO_MATCH_WEIGHT = 0.1  # prkt_core_v2.py:851-857
BEARING_GATE = 0.5  # prkt_core_v2.py:433
COLOR_GATE = 300.0  # prkt_core_v2.py:441
_EPS4 = np.finfo(float).eps * 4.0


# ----------------------------------------------------------------------------
# heading <-> quaternion round trip (utils.py:8-35 through tf.transformations)
# ----------------------------------------------------------------------------
def wrap_heading(h):
    """Heading after ``heading_to_quaternion`` then ``quaternion_to_heading``.

    utils.py:28 builds q = (0, 0, sin(h/2), cos(h/2)); utils.py:18 reads the yaw
    back as atan2(M10, M00) of the normalised rotation matrix.  Net effect: the
    heading is wrapped to (-pi, pi].
    """
    h = np.asarray(h, dtype=np.float64)
    sk = np.sin(h / 2.0)
    ck = np.cos(h / 2.0)
    nq = sk * sk + ck * ck
    s = np.sqrt(2.0 / nq)
    qz = sk * s
    qw = ck * s
    m00 = 1.0 - qz * qz  # 1 - q[1,1] - q[2,2] with q[1,1] = 0
    m10 = qz * qw  # q[0,1] + q[2,3] with q[0,1] = 0
    return np.arctan2(m10, m00)


# ----------------------------------------------------------------------------
# scalar / broadcast pieces of the measurement model
# ----------------------------------------------------------------------------
def closest_point(fx, fy, sx, sy, bearing):
    """prkt_core_v2.py:496-522 (+ utils.unit/scale/dot_product, utils.py:37-81)."""
    fx, fy, sx, sy, bearing = np.broadcast_arrays(
        *(np.asarray(a, dtype=np.float64) for a in (fx, fy, sx, sy, bearing))
    )
    ox = fx - sx
    oy = fy - sy
    cb = np.cos(bearing)
    sb = np.sin(bearing)
    length = np.sqrt(cb * cb + sb * sb + 0.0)
    ux = cb * (1.0 / length)
    uy = sb * (1.0 / length)
    magmag = ox * ux + oy * uy + 0.0 * 0.0
    behind = magmag < 0
    nx = np.where(behind, sx, sx + ux * magmag)
    ny = np.where(behind, sy, sy + uy * magmag)
    return nx, ny


def mvn_pdf_2(dx, dy, sxx, sxy, syy):
    """scipy.stats.multivariate_normal.pdf for k=2 (prkt_core_v2.py:489), closed form."""
    det = sxx * syy - sxy * sxy
    maha = (syy * dx * dx - 2.0 * sxy * dx * dy + sxx * dy * dy) / det
    return np.exp(-0.5 * (2.0 * math.log(TWO_PI) + np.log(det) + maha))


def sym3_inv_det(a, b, c, d, e, f):
    """Inverse (as 6 unique entries) and determinant of [[a,b,c],[b,d,e],[c,e,f]]."""
    c00 = d * f - e * e
    c01 = c * e - b * f
    c02 = b * e - c * d
    c11 = a * f - c * c
    c12 = b * c - a * e
    c22 = a * d - b * b
    det = a * c00 + b * c01 + c * c02
    inv = 1.0 / det
    return (c00 * inv, c01 * inv, c02 * inv, c11 * inv, c12 * inv, c22 * inv), det


def mvn_pdf_3(d0, d1, d2, a, b, c, d, e, f):
    """scipy.stats.multivariate_normal.pdf for k=3 (prkt_core_v2.py:543), closed form."""
    (i00, i01, i02, i11, i12, i22), det = sym3_inv_det(a, b, c, d, e, f)
    maha = (
        i00 * d0 * d0
        + i11 * d1 * d1
        + i22 * d2 * d2
        + 2.0 * (i01 * d0 * d1 + i02 * d0 * d2 + i12 * d1 * d2)
    )
    return np.exp(-0.5 * (3.0 * math.log(TWO_PI) + np.log(det) + maha))


def prob_position_match(fx, fy, cov_xy, sx, sy, bearing):
    """prkt_core_v2.py:457-494.  ``cov_xy`` = (sxx, sxy, syy) broadcastable."""
    fx = np.asarray(fx, dtype=np.float64)
    fy = np.asarray(fy, dtype=np.float64)
    pse = np.arctan2(fy - sy, fx - sx)
    nx, ny = closest_point(fx, fy, sx, sy, bearing)
    with np.errstate(all="ignore"):
        pdf = mvn_pdf_2(nx - fx, ny - fy, *cov_xy)
    return np.where(np.abs(pse - bearing) > math.pi / 2, 0.0, pdf)


def prob_color_match(mean_rgb, cov_rgb6, blob_rgb):
    """prkt_core_v2.py:524-544.  cov_rgb6 = (rr, rg, rb, gg, gb, bb)."""
    d0 = blob_rgb[0] - mean_rgb[0]
    d1 = blob_rgb[1] - mean_rgb[1]
    d2 = blob_rgb[2] - mean_rgb[2]
    with np.errstate(all="ignore"):
        return mvn_pdf_3(d0, d1, d2, *cov_rgb6)


def probability_of_match(sx, sy, sh, blob, mean, cov):
    """prkt_core_v2.py:383-455, broadcast over leading dims of ``mean``/``cov``.

    blob = (bearing, r, g, b); mean (...,5); cov (...,5,5).
    """
    fx = mean[..., 0]
    fy = mean[..., 1]
    expected = np.arctan2(fy - sy, fx - sx) - sh  # :408 robot frame
    delb = blob[0] - expected  # :415, NOT wrapped (:416-423 commented out)
    cd = (
        np.power(blob[1] - mean[..., 2], 2)
        + np.power(blob[2] - mean[..., 3], 2)
        + np.power(blob[3] - mean[..., 4], 2)
    )  # :425-427
    bp = 500.0 * prob_position_match(
        fx, fy, (cov[..., 0, 0], cov[..., 0, 1], cov[..., 1, 1]), sx, sy, blob[0]
    )  # :439 -- observed robot-frame bearing used as a world direction (:478,:510)
    cp = 500.0 * prob_color_match(
        (mean[..., 2], mean[..., 3], mean[..., 4]),
        (
            cov[..., 2, 2],
            cov[..., 2, 3],
            cov[..., 2, 4],
            cov[..., 3, 3],
            cov[..., 3, 4],
            cov[..., 4, 4],
        ),
        (blob[1], blob[2], blob[3]),
    )  # :446
    p = bp * cp / 250000.0  # :455
    p = np.where(np.abs(cd) > COLOR_GATE, 0.0, p)  # :441
    p = np.where(np.abs(delb) > BEARING_GATE, 0.0, p)  # :433
    return p


def measurement_jacobian(sx, sy, fx, fy):
    """prkt_core_v2.py:748-802 -> (H00, H01); the rest of H is [0 | I3]."""
    dx = fx - sx
    dy = fy - sy
    q = np.power(dx, 2) + np.power(dy, 2)
    with np.errstate(all="ignore"):
        h0 = np.where(q == 0, 0.0, dy / q)  # :789 (reference's sign/order, not textbook)
        h1 = np.where(q == 0, 0.0, dx / q)  # :795
    return h0, h1


def ekf_update_dense(sx, sy, mean, cov, blob, Qt):
    """One observation of one landmark, batched over leading dims.

    generate_measurement :859-877, measurement_jacobian :748-802,
    measurement_covariance :804-819, inverse matrix.py:11-12, kalman_gain :821-833,
    Feature.update_mean :897-914, Feature.update_covar :916-930,
    importance_factor :835-849 (Frobenius norm, matrix.py:31-33).

    Returns (new_mean, new_cov, weight, aux) with aux = dict(zhat, H, Q, K).
    """
    mean = np.asarray(mean, dtype=np.float64)
    cov = np.asarray(cov, dtype=np.float64)
    lead = mean.shape[:-1]
    fx = mean[..., 0]
    fy = mean[..., 1]
    zhat = np.stack(
        [np.arctan2(fy - sy, fx - sx), mean[..., 2], mean[..., 3], mean[..., 4]], axis=-1
    )  # :871 world frame, no heading subtraction
    h0, h1 = measurement_jacobian(sx, sy, fx, fy)
    H = np.zeros(lead + (4, 5))
    H[..., 0, 0] = h0
    H[..., 0, 1] = h1
    H[..., 1, 2] = 1.0
    H[..., 2, 3] = 1.0
    H[..., 3, 4] = 1.0
    Ht = np.swapaxes(H, -1, -2)
    Q = H @ cov @ Ht + Qt  # :817-818
    Qinv = np.linalg.inv(Q)  # :102
    K = cov @ Ht @ Qinv  # :833
    z = np.broadcast_to(np.asarray(blob, dtype=np.float64), zhat.shape)
    delz = z - zhat  # :911 / :846
    new_mean = mean + np.einsum("...ij,...j->...i", K, delz)  # :912-913
    new_cov = (np.identity(5) - K @ H) @ cov  # :928-929
    v1 = np.power(TWO_PI * np.sqrt(np.sum(Q * Q, axis=(-1, -2))), -0.5)  # :844-845
    expo = -0.5 * np.einsum("...i,...ij,...j->...", delz, Qinv, delz)  # :848
    weight = v1 * np.exp(expo)
    logweight = np.log(v1) + expo
    return new_mean, new_cov, weight, dict(zhat=zhat, H=H, Q=Q, K=K, logweight=logweight)


# ----------------------------------------------------------------------------
# systematic resampling
# ----------------------------------------------------------------------------
def low_variance_ancestors_sequential(weights, u):
    """Literal restatement of the loop at prkt_core_v2.py:216-250 (pure python)."""
    n = len(weights)
    sum_ = 0
    for w in weights:
        sum_ += float(w)
    range_ = sum_ / float(n)
    step = u * range_
    out = []
    count = 0
    for j in range(n):
        step = step - float(weights[j])
        while step <= 0.0 and count < n:
            out.append(j)
            step += range_
            count += 1
    return np.asarray(out, dtype=np.int64)


def low_variance_ancestors(weights, u):
    """Vectorised equivalent: first j with C_j >= u*r + k*r.

    If rounding leaves fewer than P emissions the reference silently shrinks the
    particle list; here (and on the device) the tail is clamped to P-1.
    """
    w = np.asarray(weights, dtype=np.float64)
    n = w.shape[0]
    c = np.cumsum(w)
    r = c[-1] / float(n)
    t = u * r + np.arange(n, dtype=np.float64) * r
    anc = np.searchsorted(c, t, side="left")
    return np.minimum(anc, n - 1).astype(np.int64)


# ----------------------------------------------------------------------------
# the filter
# ----------------------------------------------------------------------------
class OracleFilter(object):
    """SoA float64 FastSLAM-1.0 state with the reference's step semantics."""

    def __init__(self, num_particles, means, covs, immutable=None, Qt=None):
        # FastSLAM.__init__ prkt_core_v2.py:38-57, FilterParticle.__init__ :279-292,
        # load_feature_list :294-299 (ids 1..L in list order, same map in every particle)
        P = int(num_particles)
        means = np.asarray(means, dtype=np.float64).reshape(-1, 5)
        L = means.shape[0]
        covs = np.asarray(covs, dtype=np.float64).reshape(L, 5, 5)
        self.P, self.L = P, L
        self.x = np.zeros(P)
        self.y = np.zeros(P)
        self.h = np.zeros(P)
        self.logw = np.zeros(P)
        self.mean = np.broadcast_to(means, (P, L, 5)).copy()
        self.cov = np.broadcast_to(covs, (P, L, 5, 5)).copy()
        self.count = np.zeros((P, L), dtype=np.int64)
        self.immutable = (
            np.zeros(L, dtype=bool) if immutable is None else np.asarray(immutable, dtype=bool)
        )
        self.Qt = 0.1 * np.identity(4) if Qt is None else np.asarray(Qt, dtype=np.float64)
        self.n_unmatched = np.zeros(P, dtype=np.int64)  # next_id growth, :745-746
        # potential features (negative ids in the reference, :109-118): matched and updated, weigh 0.1, promoted at count > 5
        self.potential = np.zeros((P, self.L), dtype=bool)

    # -- a2 -----------------------------------------------------------------
    @staticmethod
    def motion_sigmas(v, w):
        """prkt_core_v2.py:185,190,193."""
        sd = abs(0.05 * v) + abs(0.005 * w) + 0.0005
        sh = abs(0.025 * w) + abs(0.005 * v) + 0.0005
        return sd, sh

    def motion(self, v, w, dt, z):
        """motion_model prkt_core_v2.py:168-208; ``z`` (P,3) standard normals in the
        order the reference draws them (drive, heading-1, heading-2), so that
        ``normal(0, s, 1) == s * z`` (numpy legacy ``loc + scale*gauss``)."""
        z = np.asarray(z, dtype=np.float64).reshape(self.P, 3)
        sd, sh = self.motion_sigmas(v, w)
        dheading = w * dt
        ds = v * dt + (0.0 + sd * z[:, 0])
        h1 = self.h + dheading / 2 + (0.0 + sh * z[:, 1])
        h2 = h1 + dheading / 2 + (0.0 + sh * z[:, 2])
        self.x = self.x + ds * np.cos(h1)
        self.y = self.y + ds * np.sin(h1)
        self.h = wrap_heading(h2)

    # -- a3..a6 -------------------------------------------------------------
    def associate(self, blobs, chunk=256):
        """match_features_to_scan/match_one prkt_core_v2.py:317-381 -> ids (P,B) int32.

        argmax with strict '>' from 0.0: probability 0 never matches, ties keep
        the earliest landmark (np.argmax returns the first maximum).
        """
        blobs = np.asarray(blobs, dtype=np.float64).reshape(-1, 4)
        B = blobs.shape[0]
        ids = np.zeros((self.P, B), dtype=np.int32)
        if self.L == 0:
            return ids
        for p0 in range(0, self.P, chunk):
            p1 = min(self.P, p0 + chunk)
            sx = self.x[p0:p1, None]
            sy = self.y[p0:p1, None]
            sh = self.h[p0:p1, None]
            mean = self.mean[p0:p1]
            cov = self.cov[p0:p1]
            for b in range(B):
                pr = probability_of_match(sx, sy, sh, blobs[b], mean, cov)
                pr = np.where(np.isnan(pr), 0.0, pr)
                best = np.argmax(pr, axis=1)
                pmax = pr[np.arange(p1 - p0), best]
                ids[p0:p1, b] = np.where(pmax > 0.0, best + 1, 0)
        return ids

    # -- a7..a12 ------------------------------------------------------------
    def observe(self, blobs, ids=None):
        """The per-particle body of cam_cb, prkt_core_v2.py:73-124 (without the motion
        update): weight reset is the caller's job (``reset_weights``).

        ids: None -> ML association; (B,) shared by every particle; or (P,B).
        id 0 = unmatched (weight *= 0.1, :94-95).  Blobs are applied in scan
        order, so a landmark matched twice is updated twice, sequentially (:88).
        Returns the (P,B) id matrix used.
        """
        blobs = np.asarray(blobs, dtype=np.float64).reshape(-1, 4)
        B = blobs.shape[0]
        if ids is None:
            ids = self.associate(blobs)
        ids = np.asarray(ids, dtype=np.int64)
        if ids.ndim == 1:
            ids = np.broadcast_to(ids, (self.P, B))
        ar = np.arange(self.P)
        for b in range(B):
            idb = ids[:, b]
            m = idb > 0
            self.logw[~m] += math.log(NO_MATCH_WEIGHT)
            self.n_unmatched[~m] += 1
            if not m.any():
                continue
            pi = ar[m]
            li = idb[m] - 1
            mean = self.mean[pi, li]
            cov = self.cov[pi, li]
            nm, nc, _w, aux = ekf_update_dense(self.x[pi], self.y[pi], mean, cov, blobs[b], self.Qt)
            mut = ~self.immutable[li]
            self.mean[pi[mut], li[mut]] = nm[mut]
            self.cov[pi[mut], li[mut]] = nc[mut]
            self.count[pi[mut], li[mut]] += 2  # :914 and :930
            pot = self.potential[pi, li]
            self.logw[pi] += np.where(pot, math.log(NO_MATCH_WEIGHT), aux["logweight"])  # :111-112 / :121
            promote = pot & (self.count[pi, li] > 5)  # :113-117
            self.potential[pi[promote], li[promote]] = False
        return ids

    def reset_weights(self):
        self.logw[:] = 0.0  # prkt_core_v2.py:73

    def weights(self):
        return np.exp(self.logw)

    # -- a13 ----------------------------------------------------------------
    def resample(self, u, domain="linear"):
        """low_variance_resample prkt_core_v2.py:210-252.  Weights are NOT reset (:252).

        domain="linear": weights = exp(logw), exactly the reference's quantity
        (underflows to 0 like the reference's product does);
        domain="log": weights = exp(logw - max logw) -- same ancestors whenever the
        linear weights do not underflow, and still meaningful when they do.
        """
        if domain == "log":
            w = np.exp(self.logw - np.max(self.logw))
        else:
            w = np.exp(self.logw)
        anc = low_variance_ancestors(w, u)
        self.gather(anc)
        return anc

    def gather(self, anc):
        self.x = self.x[anc]
        self.y = self.y[anc]
        self.h = self.h[anc]
        self.logw = self.logw[anc]
        self.mean = self.mean[anc]
        self.cov = self.cov[anc]
        self.count = self.count[anc]
        self.potential = self.potential[anc]
        self.n_unmatched = self.n_unmatched[anc]

    # -- a14 ----------------------------------------------------------------
    def summary(self):
        """prkt_core_v2.py:254-276."""
        xs = float(np.sum(self.x)) / float(self.P)
        ys = float(np.sum(self.y)) / float(self.P)
        hd = math.atan2(float(np.sum(np.sin(self.h))), float(np.sum(np.cos(self.h))))
        return xs, ys, hd

    # -- a12 ----------------------------------------------------------------
    def step(self, v, w, dt, z, blobs, u, ids=None, domain="linear"):
        """One cam_cb: weight reset :73, motion :75-77, observe :82-124, resample :137."""
        self.reset_weights()
        self.motion(v, w, dt, z)
        used = self.observe(blobs, ids)
        anc = self.resample(u, domain=domain)
        return used, anc


# ----------------------------------------------------------------------------
# f4: new-landmark machinery (prkt_core_v2.py:546-746)
# ----------------------------------------------------------------------------
EMPTY_COLOUR = 2.0 ** 100  # colour of a landmark slot that holds nothing yet: fails every colour gate (:441), exact in float32


def ray_intersect(x1, y1, b1, x3, y3, b3):
    """:611-643 -- do the half-lines meet (both ray parameters >= 0)."""
    ax, ay = math.cos(b1), math.sin(b1)
    bx, by = math.cos(b3), math.sin(b3)
    den = ay * bx - ax * by
    if den == 0:
        return False
    v = (ax * y3 - ay * x3 + ay * x1 - ax * y1) / den
    if abs(ay) < abs(ax):
        u = (x3 + bx * v - x1) / ax
    else:
        u = (y3 + by * v - y1) / ay
    return u >= 0 and v >= 0


def colour_distance(c1, c2):
    """:645-654."""
    return math.sqrt(math.pow(c1[0] - c2[0], 2) + math.pow(c1[1] - c2[1], 2) + math.pow(c1[2] - c2[2], 2))


def cross_readings(x1, y1, h1, x3, y3, h3):
    """:688-737 -- intersection of the two LINES (world bearings h1, h3), None when parallel."""
    x2, y2 = x1 + math.cos(h1), y1 + math.sin(h1)
    x4, y4 = x3 + math.cos(h3), y3 + math.sin(h3)
    t0, t3 = x1 * y2 - y1 * x2, x3 * y4 - x4 * y3
    den = (x1 - x2) * (y3 - y4) - (y1 - y2) * (x3 - x4)
    if den == 0:
        return None
    return ((t0 * (x3 - x4) - (x1 - x2) * t3) / den, (t0 * (y3 - y4) - (y1 - y2) * t3) / den)


class GrowingOracle(object):
    """OracleFilter with spare landmark slots and the new-landmark bookkeeping of :546-746 per particle, with the ONE change
    that makes it work: ``find_nearest_reading`` walks the orphaned readings (``hypothesis_set``), not
    ``potential_features`` (:579), and a reading pairs with the nearest stored one -- rays crossing, colour distance
    below ``pair_threshold`` -- as the docstring at :566-575 describes.  A pair becomes a potential feature at the rays'
    crossing (:656-686: mean colour, identity covariance) in the particle's next spare slot; potential features match and
    update, weigh 0.1 and are promoted past update_count 5 (:109-118, in OracleFilter.observe).  The stored reading stays
    (the reference never removes one)."""

    def __init__(self, num_particles, means, covs, spare, pair_threshold):
        means = np.asarray(means, dtype=np.float64).reshape(-1, 5)
        L0 = means.shape[0]
        covs = np.asarray(covs, dtype=np.float64).reshape(L0, 5, 5)
        em = np.zeros((spare, 5))
        em[:, 2:] = EMPTY_COLOUR
        ec = np.broadcast_to(np.identity(5), (spare, 5, 5))
        self.f = OracleFilter(num_particles, np.vstack([means, em]), np.concatenate([covs, ec]))
        self.L0, self.spare, self.thr = L0, spare, float(pair_threshold)
        P = int(num_particles)
        self.hyp = [[] for _ in range(P)]        # per particle: (id, x, y, heading, bearing, r, g, b)
        self.next_id = [L0 + 1] * P              # :298
        self.used = [0] * P                      # spare slots in use
        self.slot_id = [dict() for _ in range(P)]  # spare slot -> feature id (potential: the reference's -id)

    def add_hypothesis(self, i, blob):
        f = self.f
        x, y, h = float(f.x[i]), float(f.y[i]), float(f.h[i])
        best, best_d = None, float("inf")
        for rd in self.hyp[i]:
            if not ray_intersect(rd[1], rd[2], rd[4] + rd[3], x, y, blob[0] + h):
                continue
            d = colour_distance(rd[5:8], blob[1:4])
            if d < best_d:
                best, best_d = rd, d
        if best is not None and best_d < self.thr and self.used[i] < self.spare:
            xy = cross_readings(best[1], best[2], best[3] + best[4], x, y, h + blob[0])
            if xy is not None:
                slot = self.L0 + self.used[i]
                self.used[i] += 1
                f.mean[i, slot] = (xy[0], xy[1], (best[5] + blob[1]) / 2, (best[6] + blob[2]) / 2, (best[7] + blob[3]) / 2)
                f.cov[i, slot] = np.identity(5)
                f.count[i, slot] = 0
                f.potential[i, slot] = True
                self.slot_id[i][slot] = self.next_id[i]
                self.next_id[i] += 1
                return
        self.hyp[i].append((self.next_id[i], x, y, h, float(blob[0]), float(blob[1]), float(blob[2]), float(blob[3])))
        self.next_id[i] += 1

    def observe(self, blobs):
        """:84-124 for every particle: association + updates, then the unmatched blobs in scan order (:92-95)."""
        blobs = np.asarray(blobs, dtype=np.float64).reshape(-1, 4)
        ids = self.f.observe(blobs)
        for i in range(self.f.P):
            for b in np.nonzero(ids[i] == 0)[0]:
                self.add_hypothesis(i, blobs[b])
        return ids

    def gather(self, anc):
        self.f.gather(anc)
        self.hyp = [list(self.hyp[a]) for a in anc]
        self.next_id = [self.next_id[a] for a in anc]
        self.used = [self.used[a] for a in anc]
        self.slot_id = [dict(self.slot_id[a]) for a in anc]


# ----------------------------------------------------------------------------
# synthetic scene
# ----------------------------------------------------------------------------
def synthetic_world(L, seed=123):
    """L landmarks on a ring, random colours; returns (means (L,5), covs (L,5,5))."""
    rs = np.random.RandomState(seed)
    phi = -math.pi + TWO_PI * np.arange(L) / float(L) + 0.01
    rho = rs.uniform(8.0, 30.0, size=L)
    col = rs.uniform(0.0, 255.0, size=(L, 3))
    means = np.empty((L, 5))
    means[:, 0] = rho * np.cos(phi)
    means[:, 1] = rho * np.sin(phi)
    means[:, 2:] = col
    covs = np.broadcast_to(0.25 * np.identity(5), (L, 5, 5)).copy()
    return means, covs


def truth_step(pose, v, w, dt):
    """Noise-free mid-point motion (the motion model with zero noise)."""
    x, y, h = pose
    h1 = h + w * dt / 2
    h2 = h1 + w * dt / 2
    return x + v * dt * math.cos(h1), y + v * dt * math.sin(h1), float(wrap_heading(h2))


def synthetic_scan(world_means, pose):
    """Noise-free 360 degree bearing+colour scan: blob_j sees landmark j."""
    x, y, h = pose
    B = world_means.shape[0]
    blobs = np.empty((B, 4))
    blobs[:, 0] = np.arctan2(world_means[:, 1] - y, world_means[:, 0] - x) - h
    blobs[:, 1:] = world_means[:, 2:]
    return blobs
