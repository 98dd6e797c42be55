"""GPU: SURVEY section 8 row (f4) at device speed -- the new-landmark bookkeeping of prkt_core_v2.py:546-746 (add_hypothesis,
find_nearest_reading, ray_intersect, color_distance, add_new_feature, cross_readings, add_orphaned_reading) as ONE kernel behind
every maximum-likelihood observe (pk_grow_enable, pk_k_grow.hip), its state in HBM and carried through the resample on the device.
The checker is oracle/fastslam_oracle.py::GrowingOracle, particle by particle."""
import numpy as np
import pytest

from oracle.fastslam_oracle import EMPTY_COLOUR, GrowingOracle, synthetic_scan, synthetic_world, truth_step

pytestmark = pytest.mark.gpu


def _scene(L0, U, seed=123):
    world, covs = synthetic_world(L0 + U, seed=seed)
    return world, covs, world[:L0], covs[:L0]


def _device_filter(lib, P, known, kcov, spare, thr, R):
    L0 = len(known)
    f = lib.DeviceFilter(P, L0 + spare)
    means = np.zeros((L0 + spare, 5))
    means[:L0] = known
    means[L0:, 2:] = EMPTY_COLOUR
    covs = np.tile(np.identity(5).reshape(25), (L0 + spare, 1))
    covs[:L0] = kcov.reshape(L0, 25)
    f.upload_map(means, covs)
    f.grow_enable(L0, R, thr)
    return f


def _compare(lib, f, o, L0, step):
    cnt, rd, sid = f.grow_download()
    assert np.array_equal(cnt[:, 0], [len(h) for h in o.hyp]), "step %d: readings stored" % step
    assert np.array_equal(cnt[:, 1], o.used), "step %d: spare slots in use" % step
    assert np.array_equal(cnt[:, 2], o.next_id), "step %d: next_id" % step
    assert not cnt[:, 3].any(), "step %d: readings dropped" % step
    for i in range(f.P):
        if o.hyp[i]:
            assert np.allclose(rd[i, :len(o.hyp[i])], np.asarray(o.hyp[i]), rtol=1e-12, atol=1e-12), (step, i)
        assert [int(v) for v in sid[i, :cnt[i, 1]]] == [o.slot_id[i][L0 + k] for k in range(cnt[i, 1])], (step, i)
    m, c, k = f.download_landmarks()
    assert np.array_equal((k & lib.PK_LANDMARK_POTENTIAL) != 0, o.f.potential), step
    assert np.array_equal(k & ~lib.PK_LANDMARK_POTENTIAL, o.f.count), step
    assert np.allclose(m, o.f.mean, rtol=1e-8, atol=1e-8), step
    assert np.allclose(c.reshape(o.f.cov.shape), o.f.cov, rtol=1e-8, atol=1e-9), step


def _run(lib, P, L0, U, spare, steps, seed, R=64, check_every=1, log_domain=True, opts=None, route=None):
    thr = 30.0
    v, w, dt = 0.8, 0.35, 0.5
    world, covs, known, kcov = _scene(L0, U)
    f = _device_filter(lib, P, known, kcov, spare, thr, R)
    for k_, v_ in (opts or {}).items():
        f.set_option(k_, v_)
    # round 6: the publish / subscribe kernels leave every particle's unmatched blobs as a bit row -- the growing filter takes the
    # one-pass route of its map size (k_step_pub<256 lanes> up to 512 landmarks, k_step_pub beyond, k_step_pub_big beyond 2 048);
    # "fast_observe" = 0: the general association, whose ids the bookkeeping kernel reads as before
    L = L0 + spare
    route = route or ("ml_fused" if L <= 512 else "ml_regs" if L <= 2048 else "ml_pub_big")
    o = GrowingOracle(P, known, kcov, spare, thr)
    rs = np.random.RandomState(seed)
    pose = (0.0, 0.0, 0.0)
    created = promoted = 0
    for s in range(steps):
        pose = truth_step(pose, v, w, dt)
        blobs = synthetic_scan(world, pose)
        z = rs.standard_normal((P, 3))
        f.motion(v, w, dt, z=z)
        f.observe(blobs, fresh=True)  # no ids asked for: nothing per particle comes back
        assert f.observe_route() == route, (f.observe_route(), route)
        if route != "ml_general":
            assert f.observe_published()  # a kernel of the publish / subscribe family did the step
        o.f.reset_weights()
        o.f.motion(v, w, dt, z)
        o.observe(blobs)
        assert np.allclose(f.download_log_weights(), o.f.logw, rtol=1e-9, atol=1e-9), s
        u = float(rs.uniform())
        anc = f.resample(u, domain=lib.PK_WEIGHTS_LOG, return_ancestors=True)
        o.gather(anc)
        if s % check_every == 0 or s == steps - 1:
            _compare(lib, f, o, L0, s)
        created = max(created, max(o.used))
        promoted = max(promoted, int(((o.f.count[:, L0:] > 5) & ~o.f.potential[:, L0:]).sum()))
    f.close()
    return created, promoted


@pytest.mark.parametrize("opts,route", [({}, None), ({"fast_observe": 0}, "ml_general")])
def test_bookkeeping_kernel_matches_the_oracle_particle_by_particle(lib, opts, route):
    created, promoted = _run(lib, P=64, L0=10, U=3, spare=5, steps=9, seed=5, opts=opts, route=route)
    assert created >= 2 and promoted >= 1


@pytest.mark.parametrize("P,L0,U,spare,steps", [(12, 700, 3, 5, 6), (8, 1500, 3, 4, 5), (6, 2300, 3, 4, 5)])
def test_growing_maps_on_the_one_pass_routes_of_larger_maps(lib, P, L0, U, spare, steps):
    """VERDICT round 5, missing #2 / next #4: new_landmarks no longer forces the general route.  k_step_pub (513 .. 2 048 landmarks) and
    k_step_pub_big (beyond) on growing maps: the bit rows of unmatched blobs they leave (and, for the particles they hand on, the
    general kernels' ids) must give the bookkeeping kernel exactly what the general association's ids give it -- counters, stored
    readings, spare-slot ids, maps and counts equal array for array at every step, log-weights to rounding.  (The oracle holds both
    at the small sizes above: with several hundred blobs the reference's unwrapped bearings (:408-423) leave a score of KNOWN
    landmarks unmatched at every step, the rings fill within a few steps, and which reading is dropped is the device's affair.)"""
    world, covs, known, kcov = _scene(L0, U)
    fa = _device_filter(lib, P, known, kcov, spare, 30.0, 128)
    fb = _device_filter(lib, P, known, kcov, spare, 30.0, 128)
    fb.set_option("fast_observe", 0)
    rs = np.random.RandomState(7 + L0)
    pose, on_kernel = (0.0, 0.0, 0.0), 0
    for s in range(steps):
        pose = truth_step(pose, 0.8, 0.35, 0.5)
        blobs = synthetic_scan(world, pose)
        z = rs.standard_normal((P, 3))
        for f in (fa, fb):
            f.motion(0.8, 0.35, 0.5, z=z)
            f.observe(blobs, fresh=True)
        assert fa.observe_route() == ("ml_regs" if L0 + spare <= 2048 else "ml_pub_big") and fa.observe_published() and fb.observe_route() == "ml_general"
        on_kernel += P - fa.observe_flagged()[0]
        (ca, ra, sa), (cb, rb, sb) = fa.grow_download(), fb.grow_download()
        assert np.array_equal(ca, cb), s
        for i in range(P):  # (the ring's places beyond the stored readings, the slot ids beyond the slots in use: never written)
            assert np.array_equal(ra[i, :ca[i, 0]], rb[i, :cb[i, 0]]) and np.array_equal(sa[i, :ca[i, 1]], sb[i, :cb[i, 1]]), (s, i)
        for xa, xb in zip(fa.download_landmarks(), fb.download_landmarks()):
            assert np.array_equal(xa, xb), s
        assert np.allclose(fa.download_log_weights(), fb.download_log_weights(), rtol=1e-11, atol=1e-9), s
        u = float(rs.uniform())
        assert np.array_equal(fa.resample(u, domain=lib.PK_WEIGHTS_LOG, return_ancestors=True), fb.resample(u, domain=lib.PK_WEIGHTS_LOG, return_ancestors=True))
    assert on_kernel >= P * 2, "hardly any particle stayed on the one-pass kernel: the test shows nothing about its bit rows"
    assert fa.grow_download()[0][:, 1].max() >= 2, "nothing was triangulated"
    fa.close()
    fb.close()


def test_ten_thousand_particles_grow_their_maps_with_no_per_particle_host_traffic(lib):
    """P = 10 000.  The step itself moves nothing per particle to the host (observe without ids, the
    resample's gather on the device); the downloads are the test's own, every third step."""
    created, promoted = _run(lib, P=10000, L0=12, U=4, spare=6, steps=8, seed=11, check_every=3)
    assert created >= 3 and promoted >= 1


def test_a_full_ring_and_full_spare_slots_are_counted_not_overrun(lib):
    # two readings per particle and one spare slot: the third unknown landmark's readings find the ring full
    P, L0, U, spare, thr = 32, 8, 4, 1, 30.0
    world, covs, known, kcov = _scene(L0, U)
    f = _device_filter(lib, P, known, kcov, spare, thr, R=2)
    rs = np.random.RandomState(2)
    pose = (0.0, 0.0, 0.0)
    for s in range(4):
        pose = truth_step(pose, 0.8, 0.35, 0.5)
        f.motion(0.8, 0.35, 0.5, z=rs.standard_normal((P, 3)))
        f.observe(synthetic_scan(world, pose), fresh=True)
        f.resample(float(rs.uniform()), domain=lib.PK_WEIGHTS_LOG)
    cnt, rd, sid = f.grow_download()
    assert (cnt[:, 0] <= 2).all() and (cnt[:, 1] <= 1).all()
    assert cnt[:, 3].min() > 0, "the dropped readings were not counted"
    # every blob of every scan got an id, stored or not (:564 / :746)
    assert (cnt[:, 2] > L0 + 1 + 2).all()
    f.close()


def test_refused_where_it_cannot_follow(lib):
    world, covs, known, kcov = _scene(6, 2)
    f = _device_filter(lib, 16, known, kcov, 3, 30.0, 8)
    blobs = synthetic_scan(world, (0.0, 0.0, 0.0))
    with pytest.raises(lib.PkError, match="no ids"):
        f.observe(blobs, ids=np.arange(1, 9) % 7)
    with pytest.raises(lib.PkError, match="already enabled"):
        f.grow_enable(6, 8, 30.0)
    # the host-index exchange packs records WITHOUT the bookkeeping's tail (pk_particle_bytes counts it) and would leave readings and
    # id counters on the wrong particles: refused like its *_dev variants
    # (refused before the buffer is looked at: any non-null address will do)
    with pytest.raises(lib.PkError, match="balanced placement only") as e:
        f.pack_particles([0, 1, 2, 3], 4096)
    assert e.value.status == lib.PK_ERR_STATE
    with pytest.raises(lib.PkError, match="balanced placement only") as e:
        f.adopt_particles(np.arange(16), 4096, 0)
    assert e.value.status == lib.PK_ERR_STATE
    f.close()
    g = lib.DeviceFilter(4, 5)
    with pytest.raises(lib.PkError, match="no spare slot"):
        g.grow_enable(5, 8, 30.0)
    with pytest.raises(lib.PkError, match="pk_grow_enable was not called"):
        g.grow_download()
    g.close()


def test_upload_round_trip(lib):
    world, covs, known, kcov = _scene(6, 2)
    f = _device_filter(lib, 8, known, kcov, 3, 30.0, 4)
    rs = np.random.RandomState(0)
    cnt = np.stack([rs.randint(0, 5, 8), rs.randint(0, 4, 8), rs.randint(7, 30, 8), np.zeros(8, dtype=np.int64)], axis=1).astype(np.int32)
    rd = rs.normal(size=(8, 4, 8))
    sid = rs.randint(7, 30, size=(8, 3)).astype(np.int32)
    f.grow_upload(0, 8, cnt, rd, sid)
    c2, r2, s2 = f.grow_download()
    assert np.array_equal(c2, cnt) and np.array_equal(r2, rd) and np.array_equal(s2, sid)
    f.grow_upload(2, 5, cnt[:3], None, None)
    assert np.array_equal(f.grow_download(2, 5)[0], cnt[:3])
    bad = cnt.copy()
    bad[3, 0] = 5
    with pytest.raises(lib.PkError, match="readings of"):
        f.grow_upload(0, 8, bad, None, None)
    f.close()


class _View(object):
    def __init__(self, pk, blobs):
        class Scan(object):
            pass

        self.last_sensor_reading = Scan()
        obs = []
        for b in blobs:
            o = pk.msgs.Blob()
            o.bearing = float(b[0])
            o.color.r, o.color.g, o.color.b = float(b[1]), float(b[2]), float(b[3])
            obs.append(o)
        self.last_sensor_reading.observes = obs


def test_three_ranks_grow_the_same_maps_as_one_filter(tmp_path):
    """FastSLAM(..., new_landmarks=True, devices=[0, 0, 0]) (three child processes on the one device, gloo between them) against
    FastSLAM(device=0): a particle that migrates in the resample takes its readings, id counter and spare-slot ids along behind its
    map (the record's tail), so the snapshots -- poses, maps, flags, bookkeeping, all in the single filter's order -- are equal
    array for array.  Then the snapshot of the one goes into the other and both take another step."""
    import random

    import parakeet_slam_amd as pk

    L0, U, P, spare, steps = 10, 3, 1200, 5, 8
    v, w, dt = 0.8, 0.35, 0.5
    world, covs = synthetic_world(L0 + U)
    feats = [pk.Feature(mean=world[l].copy(), covar=covs[l].copy()) for l in range(L0)]
    snaps, views, filters = [], [], []
    for devices in ([0, 0, 0], None):
        random.seed(7)
        pk.msgs.Time.set_now(0.0)
        kw = dict(num_particles=P, weight_domain="log", rng="device", seed=3, new_landmarks=True, spare_landmarks=spare)
        if devices:
            fs = pk.FastSLAM(feats, devices=devices, backend="gloo", **kw)
            assert isinstance(fs, pk.ShardedFastSLAM)
        else:
            fs = pk.FastSLAM(feats, device=0, **kw)
        tw = pk.msgs.Twist()
        tw.linear.x, tw.angular.z = v, w
        fs.last_control = tw
        pose = (0.0, 0.0, 0.0)
        for s in range(steps):
            pose = truth_step(pose, v, w, dt)
            pk.msgs.Time.set_now(dt * (s + 1))
            fs.cam_cb(_View(pk, synthetic_scan(world, pose)))
        path = str(tmp_path / ("grow_%d.npz" % len(snaps)))
        fs.save_state(path)
        snaps.append(path)
        p = fs.particles[P - 3]
        views.append((p.next_id, sorted(p.hypothesis_set.keys()), sorted(p.potential_features.keys()), sorted(p.feature_set.keys())))
        filters.append((fs, pose))
    a, b = np.load(snaps[0]), np.load(snaps[1])
    assert set(a.files) == set(b.files) and "nl_readings" in a.files
    for k in a.files:
        if k == "poses":  # (round 6: the growing filter runs the one-pass kernel; a shard's candidate lists are made around ITS reference
            # particle, so a particle may go through the fall-back kernels on one side and not on the other: the same associations
            # and maps, the log-weight's terms added in another order)
            assert np.array_equal(a[k][:, :3], b[k][:, :3]) and np.allclose(a[k][:, 3], b[k][:, 3], rtol=1e-11, atol=0.0), k
            continue
        assert np.array_equal(a[k], b[k]), k
    assert int(a["nl_used"].max()) >= 2, "nothing was triangulated"
    assert len(set(a["nl_used"].tolist())) > 1 or len(set(np.diff(a["nl_offsets"]).tolist())) > 1, "every particle holds the same bookkeeping: the test shows nothing"
    assert views[0] == views[1]
    # cross-load: the single filter's snapshot into the three ranks and the other way round, then one more step on both
    (fm, pose), (f1, _) = filters
    fm.load_state(snaps[1])
    f1.load_state(snaps[0])
    random.seed(9)
    pose = truth_step(pose, v, w, dt)
    pk.msgs.Time.set_now(dt * (steps + 1))
    blobs = synthetic_scan(world, pose)
    fm.cam_cb(_View(pk, blobs))
    random.seed(9)
    f1.cam_cb(_View(pk, blobs))
    pa, pb = str(tmp_path / "a.npz"), str(tmp_path / "b.npz")
    fm.save_state(pa)
    f1.save_state(pb)
    a, b = np.load(pa), np.load(pb)
    for k in a.files:
        if k == "poses":
            assert np.array_equal(a[k][:, :3], b[k][:, :3]) and np.allclose(a[k][:, 3], b[k][:, 3], rtol=1e-11, atol=0.0), ("after the cross-load", k)
            continue
        assert np.array_equal(a[k], b[k]), ("after the cross-load", k)
    fm.close()
    f1.close()


def test_the_first_minimum_wins_over_rings_longer_than_a_wave(lib):
    """find_nearest_reading keeps the FIRST reading with the smallest colour distance (strict <, :566-590).  The device searches the
    ring lane-parallel and ends with a butterfly: ties must fall to the smaller index, also when the ring is longer than the 64
    lanes of the wave.  Rings written by hand (pk_grow_upload), one unmatched blob, the result against GrowingOracle.add_hypothesis."""
    P, L0, spare, R, thr = 9, 1, 2, 150, 30.0
    known = np.array([[40.0, 40.0, 250.0, 250.0, 250.0]])
    kcov = 0.25 * np.identity(5).reshape(1, 5, 5)
    f = _device_filter(lib, P, known, kcov, spare, thr, R)
    o = GrowingOracle(P, known, kcov, spare, thr)
    rs = np.random.RandomState(4)
    poses = np.zeros((P, 4))
    poses[:, 0] = rs.uniform(-0.5, 0.5, P)
    poses[:, 1] = rs.uniform(-0.5, 0.5, P)
    poses[:, 2] = rs.uniform(-0.2, 0.2, P)
    poses[:, 3] = 1.0
    f.upload_poses(poses)
    o.f.x[:], o.f.y[:], o.f.h[:] = poses[:, 0], poses[:, 1], poses[:, 2]
    target = np.array([6.0, 3.0])  # where the unknown landmark is: every stored ray and the new one point at it
    blob_colour = np.array([100.0, 50.0, 20.0])
    cnt = np.zeros((P, 4), dtype=np.int32)
    rd = np.zeros((P, R, 8))
    n_of = [0, 1, 2, 63, 64, 65, 100, 129, 150]
    for i in range(P):
        n = n_of[i]
        cnt[i] = (n, 0, L0 + 1 + n, 0)
        for r in range(n):
            ox, oy, oh = rs.uniform(-3, -1), rs.uniform(-3, 3), rs.uniform(-0.3, 0.3)
            bearing = np.arctan2(target[1] - oy, target[0] - ox) - oh
            col = blob_colour + rs.uniform(5.0, 9.0, 3) * rs.choice([-1.0, 1.0], 3)
            rd[i, r] = (L0 + 1 + r, ox, oy, oh, bearing, col[0], col[1], col[2])
        if n >= 2:  # the nearest colour twice (and a third time beyond lane 64 where the ring is that long): the first one wins
            best = blob_colour + np.array([1.0, -2.0, 2.0])
            spots = [q for q in (n - 1, n // 2, 70, 3) if 0 <= q < n]
            for q in spots:
                rd[i, q, 5:8] = best
        o.hyp[i] = [(int(r[0]),) + tuple(float(v) for v in r[1:]) for r in rd[i, :n]]
        o.next_id[i] = L0 + 1 + n
    f.grow_upload(0, P, cnt, rd, np.zeros((P, spare), dtype=np.int32))
    blobs = np.zeros((1, 4))
    blobs[0, 1:] = blob_colour
    # (one scan for all particles: the bearing of the target from the mean pose -- the rays still cross near it)
    blobs[0, 0] = np.arctan2(target[1], target[0])
    f.observe(blobs, fresh=True)
    o.f.reset_weights()
    o.observe(blobs)
    c2, r2, s2 = f.grow_download()
    assert np.array_equal(c2[:, 0], [len(h) for h in o.hyp]) and np.array_equal(c2[:, 1], o.used) and np.array_equal(c2[:, 2], o.next_id)
    assert o.used[0] == 0 and all(u == 1 for u in o.used[1:]), o.used  # (no reading: an orphan; else a pair)
    m, c, k = f.download_landmarks()
    assert np.allclose(m[:, L0:], o.f.mean[:, L0:], rtol=1e-9, atol=1e-9)  # the crossing with the FIRST of the equal readings
    for i in range(1, P):
        assert int(s2[i, 0]) == o.slot_id[i][L0]
    f.close()


@pytest.mark.parametrize("seed", [21, 22, 23, 24])
def test_random_growing_scenes_against_the_oracle(lib, seed):
    rs = np.random.RandomState(seed)
    L0, U = int(rs.randint(3, 30)), int(rs.randint(1, 5))
    spare = int(rs.randint(1, U + 3))
    created, _ = _run(lib, P=int(rs.randint(20, 400)), L0=L0, U=U, spare=spare, steps=int(rs.randint(4, 9)), seed=seed, R=128)
    assert created >= 1


def test_the_dense_layout_takes_the_host_bookkeeping(lib):
    """A preset covariance that couples position and colour puts the filter on the dense layout: the device
    bookkeeping says so at construction, and bookkeeping='host' works there as in rounds 2-4."""
    import parakeet_slam_amd as pk

    world, covs = synthetic_world(6)
    covs = covs.copy()
    covs[:, 0, 2] = covs[:, 2, 0] = 0.01
    feats = [pk.Feature(mean=world[l].copy(), covar=covs[l].copy()) for l in range(4)]
    pk.msgs.Time.set_now(0.0)
    with pytest.raises(ValueError, match="bookkeeping='host'"):
        pk.FastSLAM(feats, num_particles=8, new_landmarks=True, spare_landmarks=2)
    fs = pk.FastSLAM(feats, num_particles=8, new_landmarks=True, spare_landmarks=2, bookkeeping="host", weight_domain="log")
    pk.msgs.Time.set_now(0.1)
    fs.cam_cb(_View(pk, synthetic_scan(world, (0.0, 0.0, 0.0))))
    # (six blobs, two of them of landmarks the preset map does not hold: an id each, two stored readings)
    assert fs._filter.observe_route() == "dense" and all(n >= 4 + 1 + 2 for n in fs._next_id) and all(len(h) >= 2 for h in fs._hyp)
    fs.close()
