"""GPU parity of wc_match / wc_update_surfel_poses against the CPU oracle
(knn_surfel_matcher.cc:3-98, lidar_odometry.cc:160-170, surfel.h:48-58).  Index work is bit-exact."""
import numpy as np
import pytest

from wildcat_slam_amd import records as R
from wildcat_slam_amd import synth

pytestmark = pytest.mark.gpu


def _surfels_from_features(feat, t=None):
    """surfels whose 6-D matcher feature is exactly `feat` (identity pose, scales undone)"""
    n = len(feat)
    s = np.zeros(n, R.SURFEL)
    s["center"] = feat[:, :3] * 1.0
    s["normal"] = feat[:, 3:] * (5.0 * np.pi / 180.0)
    s["t"] = np.arange(n) * 1.0 if t is None else t
    p = np.zeros(n, R.POSE)
    p["quat"][:, 0] = 1.0
    return s, p


def test_knn_reference_property_test(gpu, oracle):
    # src/odometry/knn_surfel_matcher_test.cc:19-43 on the GPU index: 10 000 random 6-D vectors, self is nearest
    rng = np.random.default_rng(12345)
    feat = rng.uniform(-1, 1, size=(10_000, 6))
    s, p = _surfels_from_features(feat)
    _, idx, d2 = gpu.match(s, p, s, p, True, want_knn=True)
    assert idx.shape == (10_000, 10)
    assert np.array_equal(idx[:, 0], np.arange(10_000))
    # identical neighbour lists to the oracle's exact kd-tree on the features the GPU really sees
    f = np.concatenate([s["center"] / 1.0, s["normal"] / (5.0 * np.pi / 180.0)], 1)
    ridx, rd2 = oracle.knn6(f, f, 10)
    assert np.array_equal(idx.astype(np.int64), ridx.astype(np.int64))
    assert np.array_equal(d2, rd2)


@pytest.mark.parametrize("scans,patches", [(3, 400), (6, 150), (2, 30)])
def test_sliding_window_match_bit_exact(gpu, oracle, scans, patches):
    w = synth.surfel_window(scans, patches, seed=5 + scans)
    ref = oracle.match(w["surf"], w["pose"], w["surf"], w["pose"], True)
    got = gpu.match(w["surf"], w["pose"], w["surf"], w["pose"], True)
    assert len(ref) > 0.3 * len(w["surf"])
    assert np.array_equal(got, ref)


def test_fixed_window_match_bit_exact(gpu, oracle):
    w = synth.surfel_window(3, 300, seed=9, fixed_patches=200)
    ref = oracle.match(w["surf"], w["pose"], w["fix_surf"], w["fix_pose"], False)
    got = gpu.match(w["surf"], w["pose"], w["fix_surf"], w["fix_pose"], False)
    assert len(ref) > 50 and np.array_equal(got, ref)


def test_pair_dedup_chain(gpu, oracle):
    """three copies of one plane patch close together: query order decides who pairs with whom (std::set, cc:35-38)"""
    rng = np.random.default_rng(2)
    n = 60
    base = rng.uniform(-5, 5, size=(n // 3, 3))
    nrm = rng.normal(size=(n // 3, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    s = np.zeros(n, R.SURFEL)
    for r in range(3):
        sl = slice(r * (n // 3), (r + 1) * (n // 3))
        off = 0.01 * rng.normal(size=(n // 3, 3))
        off -= nrm * np.sum(off * nrm, axis=1, keepdims=True)  # stay in the plane: passes the 0.1 m plane-distance gate
        s["center"][sl] = base + off
        s["normal"][sl] = nrm
        s["t"][sl] = r * 0.5 + np.linspace(0, 0.4, n // 3)
    p = np.zeros(n, R.POSE)
    p["quat"][:, 0] = 1.0
    ref = oracle.match(s, p, s, p, True)
    got = gpu.match(s, p, s, p, True)
    assert len(ref) >= n // 3
    assert np.array_equal(got, ref)


def test_fewer_targets_than_k(gpu, oracle):
    # Q10: fewer than k = 10 targets; FLANN leaves the tail zero => candidate index 0 is re-tested
    w = synth.surfel_window(2, 3, seed=4)
    ref = oracle.match(w["surf"], w["pose"], w["surf"], w["pose"], True)
    got = gpu.match(w["surf"], w["pose"], w["surf"], w["pose"], True)
    assert np.array_equal(got, ref)
    none = gpu.match(w["surf"], w["pose"], w["surf"][:0], w["pose"][:0], False)
    assert len(none) == 0


def test_update_surfel_poses(gpu, oracle):
    pts, _ = synth.g2_lattice(100, m=32)
    s_ref, _, _ = oracle.extract_surfels(pts)
    imu, _ = synth.imu_states(synth.T0 - 0.01, synth.T0 + 0.52)
    s_gpu = s_ref.copy()
    n = len(s_ref)
    pose_ref, inb_ref = np.zeros(n, R.POSE), np.zeros(n, np.uint8)
    assert oracle.update_surfel_poses(imu, s_ref, pose_ref, inb_ref) == 0
    d_imu, d_s = gpu.to_device(imu), gpu.to_device(s_gpu)
    d_p, d_b = gpu.alloc(n * 56), gpu.to_device(np.zeros(n, np.uint8))
    gpu.update_surfel_poses(d_imu, len(imu), d_s, d_p, d_b, n)
    g_s, g_p = d_s.download(R.SURFEL, n), d_p.download(R.POSE, n)
    assert d_b.download(np.uint8, n).all()
    for f in ("center", "normal", "cov"):
        assert np.abs(g_s[f] - s_ref[f]).max() <= 1e-12 * max(1.0, np.abs(s_ref[f]).max())
    assert np.abs(g_p["pos"] - pose_ref["pos"]).max() < 1e-12 and np.abs(g_p["quat"] - pose_ref["quat"]).max() < 1e-14
    # second update only moves the pose (surfel.h:52: is_in_body_frame)
    imu2 = imu.copy()
    imu2["pos"] += 0.5
    gpu.to_device(imu2)
    d_imu2 = gpu.to_device(imu2)
    gpu.update_surfel_poses(d_imu2, len(imu2), d_s, d_p, d_b, n)
    g_s2 = d_s.download(R.SURFEL, n)
    assert np.array_equal(g_s2["center"], g_s["center"])
    assert np.abs(d_p.download(R.POSE, n)["pos"] - (pose_ref["pos"] + 0.5)).max() < 1e-12
    # out of range -> WC_ERR_RANGE (CHECK at lidar_odometry.cc:164)
    from wildcat_slam_amd import lib

    with pytest.raises(lib.WildcatError) as e:
        gpu.update_surfel_poses(gpu.to_device(imu[:5]), 5, d_s, d_p, d_b, n)
    assert e.value.code == 2


def _random_surfels(rng, n, extent, t0=0.0):
    s = np.zeros(n, R.SURFEL)
    s["center"] = rng.uniform(-extent / 2, extent / 2, size=(n, 3))
    nrm = rng.normal(size=(n, 3))
    s["normal"] = nrm / np.linalg.norm(nrm, axis=1, keepdims=True)
    s["t"] = t0 + np.sort(rng.uniform(0, 5, size=n))
    p = np.zeros(n, R.POSE)
    p["quat"][:, 0] = 1.0
    return s, p


def _feat(s):
    return np.concatenate([s["center"] / 1.0, s["normal"] / (5.0 * np.pi / 180.0)], 1)


def test_sparse_wide_extent(gpu, oracle):
    """3 000 surfels in a 600 m cube: the 10th neighbour of a query is tens of metres away (rounds 1-3: the grid's binary-search
    fallback; now simply a tree whose boxes are large); k-NN table and pairs exact"""
    rng = np.random.default_rng(77)
    s, p = _random_surfels(rng, 3000, 600.0)
    pairs, idx, d2 = gpu.match(s, p, s, p, True, want_knn=True)
    ridx, rd2 = oracle.knn6(_feat(s), _feat(s), 10)
    assert np.array_equal(idx.astype(np.int64), ridx.astype(np.int64)) and np.array_equal(d2, rd2)
    assert np.array_equal(pairs, oracle.match(s, p, s, p, True))


def test_queries_outside_the_target_box(gpu, oracle):
    """targets in a 20 m cube, queries in a 60 m cube around it (most of them outside every box of the tree)"""
    rng = np.random.default_rng(78)
    t, tp = _random_surfels(rng, 2500, 20.0)
    q, qp = _random_surfels(rng, 800, 60.0, t0=10.0)
    pairs, idx, d2 = gpu.match(q, qp, t, tp, False, want_knn=True)
    ridx, rd2 = oracle.knn6(_feat(t), _feat(q), 10)
    assert np.array_equal(idx.astype(np.int64), ridx.astype(np.int64)) and np.array_equal(d2, rd2)
    assert np.array_equal(pairs, oracle.match(q, qp, t, tp, False))


def test_random_and_coherent_normals(gpu, oracle):
    """random normals (the normal half of the metric prunes) and coherent ones (the centre half does): neighbour lists, distances
    (bit for bit) and pairs against the oracle"""
    rng = np.random.default_rng(4711)
    t, tp = _random_surfels(rng, 6000, 12.0)
    q, qp = _random_surfels(rng, 3000, 14.0, t0=10.0)
    pairs, idx, d2 = gpu.match(q, qp, t, tp, False, want_knn=True)
    ridx, rd2 = oracle.knn6(_feat(t), _feat(q), 10)
    assert np.array_equal(idx.astype(np.int64), ridx.astype(np.int64)) and np.array_equal(d2, rd2)
    assert np.array_equal(pairs, oracle.match(q, qp, t, tp, False))
    w = synth.surfel_window(4, 300, seed=21)
    assert np.array_equal(gpu.match(w["surf"], w["pose"], w["surf"], w["pose"], True), oracle.match(w["surf"], w["pose"], w["surf"], w["pose"], True))


def test_single_precision_first_look_is_conservative(gpu, oracle):
    """the fp32 first look of k_knn_tree (boxes and points) where single precision cannot tell the candidates apart: clusters of
    surfels ~95 m from the origin whose members differ by micrometres in position and by 1e-7 in the normal (below the fp32
    resolution of the features), so that which ten are nearest is decided far below its rounding; lists, distances and pairs
    must still be the oracle's, bit for bit"""
    rng = np.random.default_rng(99)
    nc, m = 60, 40
    centres = rng.uniform(-3, 3, size=(nc, 3)) + np.array([95.0, -95.0, 95.0])
    nrm = rng.normal(size=(nc, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    s = np.zeros(nc * m, R.SURFEL)
    s["center"] = np.repeat(centres, m, 0) + rng.integers(-50, 50, size=(nc * m, 3)) * 1e-6
    n2 = np.repeat(nrm, m, 0) + rng.integers(-50, 50, size=(nc * m, 3)) * 1e-7
    s["normal"] = n2 / np.linalg.norm(n2, axis=1, keepdims=True)
    s["t"] = np.sort(rng.uniform(0, 5, size=nc * m))
    p = np.zeros(nc * m, R.POSE)
    p["quat"][:, 0] = 1.0
    pairs, idx, d2 = gpu.match(s, p, s, p, True, want_knn=True)
    ridx, rd2 = oracle.knn6(_feat(s), _feat(s), 10)
    assert np.array_equal(idx.astype(np.int64), ridx.astype(np.int64)) and np.array_equal(d2, rd2)
    assert np.array_equal(pairs, oracle.match(s, p, s, p, True))


@pytest.mark.parametrize("nt", [1, 9, 10, 11, 600, 1024, 1025, 2100, 5000, 33000, 70000])
def test_tree_shapes(gpu, oracle, nt):
    """every shape of the index (match_tree.inc): a single leaf, one bucket, one sample stage (> 1 024 targets), two (> 32 k);
    other-set queries (located, sorted by leaf) and same-set ones; k-NN tables bit-exact against the oracle's kd-tree"""
    rng = np.random.default_rng(1000 + nt)
    ext = max(2.0, (nt / 20.0) ** (1 / 3))
    t, tp = _random_surfels(rng, nt, ext)
    q, qp = _random_surfels(rng, 700, 1.2 * ext, t0=10.0)
    _, idx, d2 = gpu.match(q, qp, t, tp, False, want_knn=True)
    ridx, rd2 = oracle.knn6(_feat(t), _feat(q), 10)
    assert np.array_equal(idx.astype(np.int64), ridx.astype(np.int64)) and np.array_equal(d2, rd2)
    _, idx, d2 = gpu.match(t, tp, t, tp, True, want_knn=True)
    ridx, rd2 = oracle.knn6(_feat(t), _feat(t), 10)
    assert np.array_equal(idx.astype(np.int64), ridx.astype(np.int64)) and np.array_equal(d2, rd2)


def test_coincident_features_overflow_a_bucket(gpu, oracle):
    """5 000 surfels with the SAME centre and normal next to 3 000 ordinary ones: every split plane of the coincident ones is the
    same value, they all land in one bucket that does not fit a workgroup's LDS (k_kd_bottom's in-place path), and all their
    distances tie - the lists are decided by the target index alone (FLANN's order of equal distances is the oracle's)"""
    rng = np.random.default_rng(5)
    s, p = _random_surfels(rng, 8000, 6.0)
    s["center"][1000:6000] = s["center"][1000]
    s["normal"][1000:6000] = s["normal"][1000]
    pairs, idx, d2 = gpu.match(s, p, s, p, True, want_knn=True)
    ridx, rd2 = oracle.knn6(_feat(s), _feat(s), 10)
    assert np.array_equal(d2, rd2)
    far = np.ones(len(s), bool)
    far[1000:6000] = False  # (among 5 000 equal distances the reference's choice is its tree's visiting order: compare the others)
    assert np.array_equal(idx[far].astype(np.int64), ridx[far].astype(np.int64))
    assert (d2[~far] == 0).all() and ((idx[~far] >= 1000) & (idx[~far] < 6000)).all()
    # ties by index: the ten smallest indices of the coincident block
    assert np.array_equal(idx[~far], np.tile(np.arange(1000, 1010), (5000, 1)))


def test_tree_is_reproducible_and_three_stage(gpu):
    """4.2 M targets: three sample stages (13 top levels), 256 other-set queries against numpy brute force; a second context
    returns the same table (the build is deterministic: ties by index, canonical order inside a bucket)"""
    from wildcat_slam_amd import lib

    rng = np.random.default_rng(31)
    nt = 4_200_000
    t, tp = _random_surfels(rng, nt, 120.0)
    q, qp = _random_surfels(rng, 256, 100.0, t0=10.0)
    _, idx, d2 = gpu.match(q, qp, t, tp, False, want_knn=True)
    ft, fq = _feat(t), _feat(q)
    for i in range(0, 256, 16):
        dd = np.zeros(nt)
        for d in range(6):  # flann::L2_Simple's order
            df = fq[i, d] - ft[:, d]
            dd += df * df
        o = np.lexsort((np.arange(nt), dd))[:10]
        assert np.array_equal(idx[i].astype(np.int64), o) and np.array_equal(d2[i], dd[o])
    c2 = lib.Context(0)
    _, idx2, d22 = c2.match(q, qp, t, tp, False, want_knn=True)
    c2.close()
    assert np.array_equal(idx, idx2) and np.array_equal(d2, d22)


@pytest.mark.parametrize("k", [1, 3, 16])
def test_other_neighbour_counts(gpu, oracle, k):
    """knn_k is a parameter (the reference hard-codes 10, knn_surfel_matcher.h:17): every instantiation of the top-k kernel"""
    params = oracle.default_params()
    params.knn_k = k
    gpu.set_params(params)
    try:
        w = synth.surfel_window(3, 200, seed=21, fixed_patches=100)
        _, idx, d2 = gpu.match(w["surf"], w["pose"], w["surf"], w["pose"], True, want_knn=True)
        ref = oracle.match(w["surf"], w["pose"], w["surf"], w["pose"], True, params)
        got = gpu.match(w["surf"], w["pose"], w["surf"], w["pose"], True)
        assert np.array_equal(got, ref)
        ref = oracle.match(w["surf"], w["pose"], w["fix_surf"], w["fix_pose"], False, params)
        got = gpu.match(w["surf"], w["pose"], w["fix_surf"], w["fix_pose"], False)
        assert np.array_equal(got, ref)
        assert idx.shape == (len(w["surf"]), k) and (np.diff(d2, axis=1) >= 0).all()
    finally:
        gpu.set_params(oracle.default_params())


def test_query_sharded_match_equals_unsharded(gpu):
    """the queries of KnnSurfelMatcher::Match are independent (knn_surfel_matcher.cc:22-48); in
    wc_match_sharded every rank searches a contiguous share of the queries, ONE all-gather of the gated neighbour lists
    gives every rank the whole table and the order-dependent pair de-duplication runs replicated.  Two ranks on one GPU
    (dist.ThreadComm stands in for RCCL): both must return exactly the unsharded call's pairs, for both matchers.  wc_match
    itself is NOT a collective with a communicator installed (no implicit collectives): rank 0 alone calls it."""
    import threading

    from wildcat_slam_amd import dist as wdist
    from wildcat_slam_amd import lib

    # (48 000 / 40 000 targets: trees with TWO sample stages - the later stage samples out of the scattered index list, and every
    # rank has to come to the same tree whatever order its atomics left that list in)
    w = synth.surfel_window(4, 12000, seed=11, fixed_patches=40000)
    ref_b = gpu.match(w["surf"], w["pose"], w["surf"], w["pose"], True)
    ref_u = gpu.match(w["surf"], w["pose"], w["fix_surf"], w["fix_pose"], False)
    assert len(ref_b) > 20000 and len(ref_u) > 10000
    world = 2
    ctxs = [lib.Context(0) for _ in range(world)]
    shared = wdist.ThreadComm.shared(world)
    out, errors = [None] * world, []

    def run(r):
        try:
            ctxs[r].set_comm(wdist.ThreadComm(shared, r, ctxs[r]))
            if r == 0:  # the plain call on ONE rank only: it must not wait for the others
                alone = ctxs[r].match(w["surf"], w["pose"], w["surf"], w["pose"], True)
                assert alone.tobytes() == ref_b.tobytes() and shared["calls"][0] == 0
            b = ctxs[r].match(w["surf"], w["pose"], w["surf"], w["pose"], True, sharded=True)
            u = ctxs[r].match(w["surf"], w["pose"], w["fix_surf"], w["fix_pose"], False, sharded=True)
            out[r] = (b, u)
        except Exception as e:  # pragma: no cover
            errors.append(e)
            shared["bar"].abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert not errors, errors
    assert shared["calls"][0] == shared["calls"][1] == 2  # one all-gather per matcher call
    for r in range(world):
        assert out[r][0].tobytes() == ref_b.tobytes() and out[r][1].tobytes() == ref_u.tobytes()
    for c in ctxs:
        c.close()


@pytest.mark.parametrize("serial", [False, True])
def test_match_pair_equals_the_two_searches_and_the_oracle(gpu, oracle, serial):
    """wc_match_pair (both KnnSurfelMatcher objects of lidar_odometry.cc:530-538 side by side, the fixed-window search on a helper
    context and host thread) against two wc_match calls and against the oracle; several calls in a row reuse the helper."""
    gpu.set_dev_option("match_pair_serial", 1 if serial else 0)  # (both searches on the ctx, one after the other)
    w = synth.surfel_window(4, 2000, seed=31, fixed_patches=1500)
    ns, nf = len(w["surf"]), len(w["fix_surf"])
    d_s, d_p = gpu.to_device(w["surf"]), gpu.to_device(w["pose"])
    d_fs, d_fp = gpu.to_device(w["fix_surf"]), gpu.to_device(w["fix_pose"])
    d_b, d_u = gpu.alloc(8 * ns), gpu.alloc(8 * ns)
    ref_b = oracle.match(w["surf"], w["pose"], w["surf"], w["pose"], True)
    ref_u = oracle.match(w["surf"], w["pose"], w["fix_surf"], w["fix_pose"], False)
    one_b = gpu.match(w["surf"], w["pose"], w["surf"], w["pose"], True)
    one_u = gpu.match(w["surf"], w["pose"], w["fix_surf"], w["fix_pose"], False)
    for _ in range(3):
        nb, nu = gpu.match_pair_device(d_s, d_p, ns, d_fs, d_fp, nf, d_b, ns, d_u, ns)
        got_b, got_u = d_b.download(R.PAIR, nb), d_u.download(R.PAIR, nu)
        assert np.array_equal(got_b, one_b) and np.array_equal(got_u, one_u)
        assert np.array_equal(got_b, ref_b) and np.array_equal(got_u, ref_u)
    gpu.set_dev_option("match_pair_serial", 0)
    assert len(ref_b) > 1000 and len(ref_u) > 500


@pytest.mark.parametrize("group", ["0", "1"], ids=["lane-per-query", "eight-lanes-per-query"])
def test_both_walks_every_k(gpu, group):
    """the matcher picks its walk by the call's sizes (below 750 k queries: eight lanes per query); the development option knn_group
    (wc_ctx_set_dev_option) pins it - tests/_match_walk_worker.py runs in a process of its own under each setting: every
    instantiated k, both kinds of search, trees of one leaf ... two sample stages, against the oracle"""
    import os
    import subprocess
    import sys

    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_match_walk_worker.py")
    r = subprocess.run([sys.executable, worker, group], env=dict(os.environ), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0 and ("walk %s ok" % group) in r.stdout.decode(), r.stdout.decode()[-3000:]


@pytest.mark.parametrize("group", [0, 1], ids=["lane-per-query", "eight-lanes-per-query"])
def test_early_bound_of_the_walks_changes_no_pair(gpu, oracle, group):
    """Round 6: a walk is bounded by the nearest candidate that passes the gates as well as by the k-th distance (the reference takes the FIRST
    gated neighbour, knn_surfel_matcher.cc:24-46; one set: only candidates with a larger index end the scan, cc:35-38).  Gates that reject the
    nearest neighbours (a long time gate, a narrow angle, a thin plane gate), gates that reject nothing, k of 1 ... 16, both sets, both walks,
    the development option knn_early off (0), on (1) and for two sets only (2): every pair list is the oracle's."""
    rng = np.random.default_rng(77)
    w = synth.surfel_window(5, 220, seed=31, fixed_patches=150)
    # jitter the normals and centres a little so that the gates cut THROUGH the neighbour lists
    w["surf"]["normal"] += rng.normal(scale=0.03, size=w["surf"]["normal"].shape)
    w["surf"]["normal"] /= np.linalg.norm(w["surf"]["normal"], axis=1, keepdims=True)
    w["surf"]["center"] += rng.normal(scale=0.02, size=w["surf"]["center"].shape)
    dt = float(np.ptp(w["surf"]["t"]))
    settings = [dict(), dict(time_diff_min=0.45 * dt), dict(time_diff_min=2.0 * dt), dict(surfel_dist_max=0.004), dict(surfel_dist_max=1e3, time_diff_min=0.0),
                dict(knn_k=1), dict(knn_k=3, time_diff_min=0.3 * dt), dict(knn_k=16, surfel_dist_max=0.01)]
    seen_cut = 0
    try:
        gpu.set_dev_option("knn_group", group)
        for kw in settings:
            params = oracle.default_params()
            for k_, v in kw.items():
                setattr(params, k_, v)
            gpu.set_params(params)
            ref_s = oracle.match(w["surf"], w["pose"], w["surf"], w["pose"], True, params)
            ref_f = oracle.match(w["surf"], w["pose"], w["fix_surf"], w["fix_pose"], False, params)
            seen_cut += int(0 < len(ref_s) < len(w["surf"]))
            for early in (0, 1, 2):
                gpu.set_dev_option("knn_early", early)
                assert np.array_equal(gpu.match(w["surf"], w["pose"], w["surf"], w["pose"], True), ref_s), (kw, early)
                assert np.array_equal(gpu.match(w["surf"], w["pose"], w["fix_surf"], w["fix_pose"], False), ref_f), (kw, early)
        assert seen_cut >= 3
    finally:
        gpu.set_dev_option("knn_early", 1)
        gpu.set_dev_option("knn_group", -1)
        gpu.set_params(oracle.default_params())
