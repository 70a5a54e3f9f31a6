"""GPU parity of the window problem (factors, J^T J / J^T r reduction, LM) against the CPU oracle.
Reference: cost_functor.h (all factors), lidar_odometry.cc:254-363 (problem construction), :551-561 (solve).
Tolerance (north_star): pose increments within 1e-6 relative; H, g, cost far tighter (pure fp64 re-association)."""
import numpy as np
import pytest

from wildcat_slam_amd import records as R
from wildcat_slam_amd import synth

pytestmark = pytest.mark.gpu


def _mode2_pairs(w, count, rng):
    """hand-made correspondences whose two surfels fall into the same sample interval (Mode 2, cost_functor.h:98)"""
    t, st = w["surf"]["t"], w["sample_times"]
    blk = np.searchsorted(st, t, side="right") - 1
    out = []
    for b in np.unique(blk):
        idx = np.nonzero(blk == b)[0]
        if len(idx) >= 2:
            for _ in range(3):
                i, j = rng.choice(idx, 2, replace=False)
                if t[i] != t[j]:
                    out.append((i, j) if t[i] < t[j] else (j, i))
        if len(out) >= count:
            break
    p = np.zeros(len(out), R.PAIR)
    p["first"], p["second"] = [a for a, _ in out], [b for _, b in out]
    return p


def _rotate_inv(quat, v):
    """R(quat)^T v for arrays of unit quaternions (w, x, y, z)"""
    w_, u = quat[:, 0:1], -quat[:, 1:4]
    t = 2.0 * np.cross(u, v)
    return v + w_ * t + np.cross(u, t)


def _setup(gpu, oracle, n_scans=3, patches=300, fixed=120, quirks=1, fix_first=True, with_imu=True, extra_mode2=True, seed=11,
           one_plane=False, sample_dt=0.08):
    w = synth.surfel_window(n_scans, patches, seed=seed, fixed_patches=fixed, sample_dt=sample_dt)
    if one_plane:  # degenerate geometry: every surfel normal is the world z axis (two translations and yaw unobservable by lidar)
        for k_s, k_p in (("surf", "pose"), ("fix_surf", "fix_pose")):
            if len(w[k_s]):
                zw = np.tile(np.array([[0.0, 0.0, 1.0]]), (len(w[k_s]), 1))
                w[k_s]["normal"] = _rotate_inv(w[k_p]["quat"], zw)
    params = oracle.default_params()
    params.reference_quirks = quirks
    pairs = oracle.match(w["surf"], w["pose"], w["surf"], w["pose"], True, params)
    if extra_mode2:
        m2 = _mode2_pairs(w, 25, np.random.default_rng(seed))
        pairs = np.concatenate([pairs, m2])
    pf = oracle.match(w["surf"], w["pose"], w["fix_surf"], w["fix_pose"], False, params) if fixed else np.zeros(0, R.PAIR)
    W = oracle.Window(w["sample_times"], w["grav"], fix_first, params)
    W.add_binary(w["surf"], w["pose"], pairs)
    if fixed:
        W.add_unary(w["fix_surf"], w["fix_pose"], w["surf"], w["pose"], pf)
    if with_imu:
        W.add_imu(w["imu"])
    gpu.set_params(params)
    d_surf, d_pose = gpu.to_device(w["surf"]), gpu.to_device(w["pose"])
    d_pairs = gpu.to_device(pairs)
    d_fs = gpu.to_device(w["fix_surf"]) if fixed else None
    d_fp = gpu.to_device(w["fix_pose"]) if fixed else None
    d_pf = gpu.to_device(pf) if fixed else None
    gpu.window_build(d_surf, d_pose, d_pairs, len(pairs), w["imu"] if with_imu else None, w["sample_times"], w["grav"], fix_first,
                     d_fs, d_fp, d_pf, len(pf))
    keep = (d_surf, d_pose, d_pairs, d_fs, d_fp, d_pf)
    return w, W, keep


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("quirks", [1, 0])
def test_evaluate_and_linearize_match_oracle(gpu, oracle, quirks):
    w, W, keep = _setup(gpu, oracle, quirks=quirks)
    c = W.counts()
    assert c[0] > 0 and c[1] > 0 and c[2] > 0 and c[3] > 0 and c[4] > 0 and c[5] > 0  # every factor mode present
    nb, nu, ni, _ = gpu.window_counts()
    assert nb == c[0] + c[1] + c[2] and nu == c[3] and ni == c[4] + c[5]
    rng = np.random.default_rng(0)
    for x in (np.zeros(12 * W.ns), 1e-3 * rng.normal(size=12 * W.ns)):
        cost_ref, res_ref = W.evaluate(x, want_residuals=True)
        cost, res = gpu.window_evaluate(x, want_residuals=True)
        assert abs(cost - cost_ref) <= 1e-11 * cost_ref
        assert np.abs(res - res_ref).max() <= 1e-9 * np.abs(res_ref).max()
        H_ref, g_ref, cl_ref = W.linearize(x)
        H, g, cl = gpu.window_linearize(x)
        assert abs(cl - cl_ref) <= 1e-11 * cl_ref
        assert _rel(H, H_ref) <= 1e-10 and _rel(g, g_ref) <= 1e-10
        assert np.array_equal(H, H.T)
        if True:  # gauge: columns 3..5 of block 0 do not exist (SubsetParameterization, cc:556-560)
            assert not H[3:6].any() and not g[3:6].any()


@pytest.mark.parametrize("cfg", [dict(), dict(quirks=0), dict(fix_first=False), dict(with_imu=False, fix_first=True), dict(fixed=0),
                                 dict(one_plane=True), dict(one_plane=True, with_imu=False),
                                 # 64 sample states = 768 unknowns = 25 Cholesky panels: several chunks of the back substitution,
                                 # 64 x 64 tiles off the diagonal, the lead workgroup next to real tiles
                                 dict(n_scans=10, patches=80, fixed=40)])
def test_lm_solve_matches_oracle(gpu, oracle, cfg):
    w, W, keep = _setup(gpu, oracle, **cfg)
    x0 = np.zeros(12 * W.ns)
    x_ref, s_ref, first_ref = W.solve(x0)
    x, s, first = gpu.window_solve(x0)
    assert s.termination == s_ref.termination and s.iterations == s_ref.iterations
    assert s.successful_steps == s_ref.successful_steps
    assert abs(s.initial_cost - s_ref.initial_cost) <= 1e-11 * s_ref.initial_cost
    assert abs(s.final_cost - s_ref.final_cost) <= 1e-8 * s_ref.final_cost
    # north_star: pose increments within 1e-6 relative
    assert _rel(first, first_ref) <= 1e-6, _rel(first, first_ref)
    assert _rel(x, x_ref) <= 1e-6, _rel(x, x_ref)
    if not cfg.get("one_plane"):
        assert s_ref.final_cost < 0.7 * s_ref.initial_cost  # the solve really removed the injected pose error


@pytest.mark.parametrize("sample_dt,ns", [(0.34, 4), (0.25, 6), (0.2, 7), (0.15, 8)])
def test_tiny_windows_match_oracle(gpu, oracle, sample_dt, ns):
    """four to eight sample states: the pose half of the reduced system is ONE 32 x 32 block (ns = 4, 5: no panel step at all, the back
    product k_back_mul takes its only block from k_schur_form's factor) or two (the appended identity rows get one step); round 6"""
    w, W, keep = _setup(gpu, oracle, n_scans=2, patches=150, fixed=60, extra_mode2=False, sample_dt=sample_dt)
    assert W.ns == ns
    x0 = np.zeros(12 * W.ns)
    x_ref, s_ref, first_ref = W.solve(x0)
    x, s, first = gpu.window_solve(x0)
    assert s.termination == s_ref.termination and s.iterations == s_ref.iterations and s.successful_steps == s_ref.successful_steps
    assert abs(s.final_cost - s_ref.final_cost) <= 1e-8 * s_ref.final_cost
    assert _rel(first, first_ref) <= 1e-6 and _rel(x, x_ref) <= 1e-6, (_rel(first, first_ref), _rel(x, x_ref))


def test_far_pairs_stay_zero_across_builds_and_timed_linearisation(gpu, oracle):
    """k_gather writes only the 6 x 6 pose corner of a block pair more than two sample blocks apart; the rest of such a block is
    zero from the build (round 5).  Windows of different sizes built back to back on ONE context - the dense H moves in memory with
    its leading dimension - must each match the oracle entry by entry, also after the back-to-back timed linearisations."""
    rng = np.random.default_rng(3)
    for cfg in (dict(n_scans=10, patches=80, fixed=40), dict(n_scans=3, patches=300, fixed=120), dict(n_scans=7, patches=120, fixed=0, fix_first=False),
                dict(n_scans=10, patches=80, fixed=40, with_imu=False)):
        w, W, keep = _setup(gpu, oracle, **cfg)
        x = 1e-3 * rng.normal(size=12 * W.ns)
        H_ref, g_ref, cl_ref = W.linearize(x)
        ms = gpu.window_linearize_timed(x, 5)
        assert 0.0 < ms < 50.0
        H, g, cl = gpu.window_linearize(x)
        assert abs(cl - cl_ref) <= 1e-11 * cl_ref
        assert _rel(H, H_ref) <= 1e-10 and _rel(g, g_ref) <= 1e-10
        assert np.array_equal(H, H.T)
        far_zero = np.ones_like(H, bool)  # entries the far pairs never write
        ns = W.ns
        for i in range(ns):
            for j in range(ns):
                if abs(i - j) <= 2:
                    far_zero[12 * i:12 * i + 12, 12 * j:12 * j + 12] = False
                else:
                    far_zero[12 * i:12 * i + 6, 12 * j:12 * j + 6] = False
        assert not H[far_zero].any() and not H_ref[far_zero].any()


def test_window_errors(gpu, oracle):
    from wildcat_slam_amd import lib

    w = synth.surfel_window(2, 50, seed=3)
    d_surf, d_pose = gpu.to_device(w["surf"]), gpu.to_device(w["pose"])
    bad = np.zeros(1, R.PAIR)
    bad["first"], bad["second"] = 5, 5  # not (older, newer): CHECK_LT at lidar_odometry.cc:256
    with pytest.raises(lib.WildcatError) as e:
        gpu.window_build(d_surf, d_pose, gpu.to_device(bad), 1, None, w["sample_times"], w["grav"], True)
    assert e.value.code == 3
    ok = np.zeros(1, R.PAIR)
    ok["first"], ok["second"] = 0, len(w["surf"]) - 1
    with pytest.raises(lib.WildcatError) as e:  # sample states do not bracket the surfels: CHECKs at cc:259-265
        gpu.window_build(d_surf, d_pose, gpu.to_device(ok), 1, None, w["sample_times"][:3], w["grav"], True)
    assert e.value.code == 2


def test_builds_back_to_back_and_after_a_failed_build(gpu, oracle):
    """wc_window_build returns with its last copy still in flight and keeps two arrays zero at rest (status words, segment heads by
    key): a second build right behind the first (another window, no solve in between), a build behind a FAILED build (whose flags and
    heads were left on the device) and a build with more sample states than the one before must each give the problem a fresh
    context gives.  Reference: lidar_odometry.cc:541-545 (the problem is constructed anew for every solve)."""
    from wildcat_slam_amd import lib

    def reference(cfg):
        fresh = lib.Context(0)
        try:
            _, W, keep = _setup(fresh, oracle, **cfg)
            H, g, c = fresh.window_linearize(np.zeros(12 * W.ns))
            return H, g, c, W
        finally:
            fresh.close()

    small, large = dict(n_scans=3, patches=200, fixed=80, seed=5), dict(n_scans=6, patches=150, fixed=60, seed=7)
    for first, second in ((small, large), (large, small)):
        _setup(gpu, oracle, **first)           # built, never solved
        _, W, keep = _setup(gpu, oracle, **second)  # right behind it
        H, g, c = gpu.window_linearize(np.zeros(12 * W.ns))
        H_ref, g_ref, c_ref, W_ref = reference(second)
        assert np.array_equal(H, H_ref) and np.array_equal(g, g_ref) and c == c_ref
        H_o, g_o, c_o = W.linearize(np.zeros(12 * W.ns))
        assert _rel(H, H_o) <= 1e-10 and _rel(g, g_o) <= 1e-10 and abs(c - c_o) <= 1e-11 * c_o
    # a failed build (a correspondence that is not (older, newer)) leaves flags and heads behind
    w = synth.surfel_window(2, 50, seed=3)
    d_surf, d_pose = gpu.to_device(w["surf"]), gpu.to_device(w["pose"])
    bad = np.zeros(4, R.PAIR)
    bad["first"], bad["second"] = [5, 1, 2, 3], [5, 40, 41, 42]
    with pytest.raises(lib.WildcatError):
        gpu.window_build(d_surf, d_pose, gpu.to_device(bad), len(bad), None, w["sample_times"], w["grav"], True)
    _, W, keep = _setup(gpu, oracle, **small)
    H, g, c = gpu.window_linearize(np.zeros(12 * W.ns))
    H_ref, g_ref, c_ref, _ = reference(small)
    assert np.array_equal(H, H_ref) and np.array_equal(g, g_ref) and c == c_ref


def test_two_rank_sharded_solve_on_one_gpu(gpu, oracle):
    """The multi-GPU scheme of SURVEY 8(e) with both 'ranks' on one device: two contexts, each with a contiguous half of
    the correspondences (IMU factors on rank 0 only, unknowns replicated), the all-reduce callback of the C-ABI summing the
    packed {H, g, cost} buffers through host memory in lock step (threads + barrier stand in for RCCL).  Both ranks must
    end at the same point, take the same number of iterations and agree with the unsharded solve."""
    import threading

    from wildcat_slam_amd import dist as wdist
    from wildcat_slam_amd import lib

    w = synth.surfel_window(4, 400, seed=23, fixed_patches=200)
    params = oracle.default_params()
    pairs = oracle.match(w["surf"], w["pose"], w["surf"], w["pose"], True, params)
    pf = oracle.match(w["surf"], w["pose"], w["fix_surf"], w["fix_pose"], False, params)
    ns = len(w["sample_times"])
    x0 = np.zeros(12 * ns)

    def build(ctx, lo_b, cnt_b, lo_u, cnt_u, with_imu):
        ctx.set_params(params)
        keep = [ctx.to_device(w["surf"]), ctx.to_device(w["pose"]), ctx.to_device(pairs[lo_b:lo_b + cnt_b]),
                ctx.to_device(w["fix_surf"]), ctx.to_device(w["fix_pose"]), ctx.to_device(pf[lo_u:lo_u + cnt_u])]
        ctx.window_build(keep[0], keep[1], keep[2], cnt_b, w["imu"] if with_imu else None, w["sample_times"], w["grav"], True,
                         keep[3], keep[4], keep[5], cnt_u)
        return keep

    # reference: everything on one context
    keep_all = build(gpu, 0, len(pairs), 0, len(pf), True)
    x_ref, s_ref, _ = gpu.window_solve(x0)

    world = 2
    ctxs = [lib.Context(0) for _ in range(world)]
    bar = threading.Barrier(world)
    stage = [None] * world
    calls = [0] * world

    def make_cb(r):
        def cb(ptr, count):  # device pointer of this rank's buffer
            host = ctxs[r].download_raw(ptr, count * 8).view(np.float64).copy()
            stage[r] = host
            bar.wait()
            total = stage[0] + stage[1]  # fixed order: bitwise the same on both ranks
            bar.wait()
            ctxs[r].upload_raw(ptr, total)
            calls[r] += 1

        return cb

    results = [None] * world
    errors = []

    def run(r):
        try:
            lo_b, cnt_b = wdist.shard_range(len(pairs), r, world)
            lo_u, cnt_u = wdist.shard_range(len(pf), r, world)
            keep = build(ctxs[r], lo_b, cnt_b, lo_u, cnt_u, r == 0)
            ctxs[r].window_set_allreduce(make_cb(r))
            results[r] = ctxs[r].window_solve(x0) + (keep,)
        except Exception as e:  # pragma: no cover
            errors.append(e)
            bar.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert not errors, errors
    (x_a, s_a, _, _), (x_b, s_b, _, _) = results
    assert calls[0] == calls[1] and calls[0] > 0
    assert np.array_equal(x_a, x_b), "ranks diverged"
    assert s_a.iterations == s_b.iterations == s_ref.iterations
    assert _rel(x_a, x_ref) < 1e-6
    for c in ctxs:
        c.close()
    del keep_all


@pytest.mark.parametrize("option", [("lm_side_stream", 2), ("lm_side_stream", 0), ("lm_one_collective", 1)])
def test_sharded_solve_forms(gpu, oracle, option):
    """the forms of a sharded window's linearisation (round 6) on three thread-ranks: the two-collective form with the large
    collective on the side stream (its event choreography, forced for a communicator of callbacks) and on the ctx stream, and rounds
    3 - 5's one all-reduce: ranks bitwise equal, iterations of the one-rank solve, corrections 1e-6, and the payload each form quotes"""
    import threading

    from wildcat_slam_amd import dist as wdist
    from wildcat_slam_amd import lib

    w = synth.surfel_window(4, 260, seed=31, fixed_patches=130)
    params = oracle.default_params()
    pairs = oracle.match(w["surf"], w["pose"], w["surf"], w["pose"], True, params)
    pf = oracle.match(w["surf"], w["pose"], w["fix_surf"], w["fix_pose"], False, params)
    ns = len(w["sample_times"])
    x0 = np.zeros(12 * ns)
    keep = [gpu.to_device(a) for a in (w["surf"], w["pose"], pairs, w["fix_surf"], w["fix_pose"], pf)]
    gpu.window_build(keep[0], keep[1], keep[2], len(pairs), w["imu"], w["sample_times"], w["grav"], True, keep[3], keep[4], keep[5], len(pf))
    x_ref, s_ref, _ = gpu.window_solve(x0)
    world = 3
    ctxs = [lib.Context(0) for _ in range(world)]
    shared = wdist.ThreadComm.shared(world)
    res, errors = [None] * world, []

    def run(r):
        try:
            c = ctxs[r]
            c.set_dev_option(*option)
            c.set_comm(wdist.ThreadComm(shared, r, c))
            k = [c.to_device(a) for a in (w["surf"], w["pose"], pairs, w["fix_surf"], w["fix_pose"], pf)]
            c.window_build(k[0], k[1], k[2], len(pairs), w["imu"], w["sample_times"], w["grav"], True, k[3], k[4], k[5], len(pf), sharded=True)
            want = wdist.packed_count(ns) if option[0] == "lm_one_collective" else wdist.corner_count(ns)
            assert c.window_reduce_bytes() == 8 * want
            res[r] = c.window_solve(x0) + (k,)
        except Exception as e:  # pragma: no cover
            errors.append(e)
            shared["bar"].abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert not errors, errors
    for r in range(1, world):
        assert np.array_equal(res[0][0], res[r][0]), "ranks diverged"
    assert res[0][1].iterations == s_ref.iterations and res[0][1].successful_steps == s_ref.successful_steps
    assert _rel(res[0][0], x_ref) < 1e-6
    for c in ctxs:
        c.close()


def test_sharded_solve_through_the_ctx_communicator(gpu, oracle):
    """the same lock-step solve through wc_window_build_sharded: every rank passes the SAME replicated arguments, the library
    takes the rank's share of the correspondences and of the IMU triples and the all-reduce goes through the ctx's wc_comm (the
    slot the in-library RCCL binding of csrc/comm.hip fills with ncclAllReduce on the ctx stream).  A problem built with the
    plain wc_window_build is not a collective although the communicator is installed; ranks that pass different lists are
    caught by the build's share check."""
    import threading

    from wildcat_slam_amd import dist as wdist
    from wildcat_slam_amd import lib

    w = synth.surfel_window(3, 300, seed=29, fixed_patches=150)
    params = oracle.default_params()
    pairs = oracle.match(w["surf"], w["pose"], w["surf"], w["pose"], True, params)
    ns = len(w["sample_times"])
    x0 = np.zeros(12 * ns)
    keep = [gpu.to_device(w["surf"]), gpu.to_device(w["pose"]), gpu.to_device(pairs)]
    gpu.window_build(keep[0], keep[1], keep[2], len(pairs), w["imu"], w["sample_times"], w["grav"], True)
    x_ref, s_ref, _ = gpu.window_solve(x0)
    world = 2
    ctxs = [lib.Context(0) for _ in range(world)]
    shared = wdist.ThreadComm.shared(world)
    res, errors = [None] * world, []

    def run(r):
        try:
            c = ctxs[r]
            c.set_comm(wdist.ThreadComm(shared, r, c))
            k = [c.to_device(w["surf"]), c.to_device(w["pose"]), c.to_device(pairs)]
            if r == 0:  # plain build + solve on ONE rank with the communicator installed: no collective, the whole problem
                c.window_build(k[0], k[1], k[2], len(pairs), w["imu"], w["sample_times"], w["grav"], True)
                xa, sa, _ = c.window_solve(x0)
                assert shared["calls"][0] == 0 and np.array_equal(xa, x_ref) and c.window_reduce_bytes() == 0
            # ranks that disagree about the problem: the share check of the build fails on every rank
            with pytest.raises(lib.WildcatError):
                c.window_build(k[0], k[1], k[2], len(pairs) - (7 if r == 1 else 0), w["imu"], w["sample_times"], w["grav"], True, sharded=True)
            c.window_build(k[0], k[1], k[2], len(pairs), w["imu"], w["sample_times"], w["grav"], True, sharded=True)
            nb, _, ni, _ = c.window_counts()
            assert nb == wdist.shard_range(len(pairs), r, world)[1] and 0 < ni < len(w["imu"]) - 2
            assert c.window_reduce_bytes() == 8 * wdist.corner_count(ns)
            res[r] = c.window_solve(x0) + (k,)
        except Exception as e:  # pragma: no cover
            errors.append(e)
            shared["bar"].abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert not errors, errors
    assert shared["calls"][0] == shared["calls"][1] > 0
    assert np.array_equal(res[0][0], res[1][0]), "ranks diverged"
    assert res[0][1].iterations == res[1][1].iterations == s_ref.iterations
    assert _rel(res[0][0], x_ref) < 1e-6
    for c in ctxs:
        c.close()


def test_c3_window_full_size(gpu, oracle):
    """BASELINE config C3 (5-scan window, 200 k surfels) at full size.  Correspondences from the GPU matcher; the normal equations
    are symmetric with a positive diagonal, bitwise reproducible, the gradient of the gauge-fixed position block is zero, the LM
    run decreases the cost and ends at a point where the gradient is much smaller than at the start; every pair is (older,
    newer); and BY VALUE against the oracle's linearisation of the same 200 k + 20 k factors (a second or two of one core): H
    block by block, g, cost, every loss-corrected residual, at x = 0 and at a random point."""
    w = synth.surfel_window(5, 40_000, seed=31, fixed_patches=20_000)
    n_s = len(w["surf"])
    d_surf, d_pose = gpu.to_device(w["surf"]), gpu.to_device(w["pose"])
    d_fs, d_fp = gpu.to_device(w["fix_surf"]), gpu.to_device(w["fix_pose"])
    d_pairs, d_pf = gpu.alloc(8 * n_s), gpu.alloc(8 * n_s)
    n_b = gpu.match_device(d_surf, d_pose, n_s, d_surf, d_pose, n_s, True, d_pairs, n_s)
    n_u = gpu.match_device(d_surf, d_pose, n_s, d_fs, d_fp, len(w["fix_surf"]), False, d_pf, n_s)
    assert n_b > n_s // 2 and n_u > 0
    pairs = d_pairs.download(R.PAIR, n_b)
    t = w["surf"]["t"]
    assert np.all(t[pairs["first"]] < t[pairs["second"]])  # (older, newer), knn_surfel_matcher.cc:40-44
    assert len(np.unique(pairs.view(np.uint64))) == n_b   # no duplicate pair
    gpu.window_build(d_surf, d_pose, d_pairs, n_b, w["imu"], w["sample_times"], w["grav"], True, d_fs, d_fp, d_pf, n_u)
    ns = len(w["sample_times"])
    x0 = np.zeros(12 * ns)
    H, g, c0 = gpu.window_linearize(x0)
    H2, g2, c02 = gpu.window_linearize(x0)
    assert np.array_equal(H, H2) and np.array_equal(g, g2) and c0 == c02  # fixed reduction order
    assert np.array_equal(H, H.T)
    d = np.diag(H)
    assert np.all(d[6:] > 0)                 # every free unknown is constrained
    assert np.all(g[3:6] == 0) and np.all(H[3:6, :] == 0)  # SubsetParameterization(12, {3,4,5}) on the first sample (cc:556-560)
    x, s, _ = gpu.window_solve(x0)
    assert s.iterations >= 1 and s.final_cost < s.initial_cost
    assert abs(s.initial_cost - c0) <= 1e-9 * c0
    _, g_end, c_end = gpu.window_linearize(x)
    assert abs(c_end - s.final_cost) <= 1e-9 * c_end
    assert np.abs(g_end).max() < 1e-2 * np.abs(g).max()
    # by value (lidar_odometry.cc:254-317 -> cost_functor.h through oracle/window.cc)
    pf = d_pf.download(R.PAIR, n_u)
    Wref = oracle.Window(w["sample_times"], w["grav"], True)
    Wref.add_binary(w["surf"], w["pose"], pairs)
    Wref.add_unary(w["fix_surf"], w["fix_pose"], w["surf"], w["pose"], pf)
    Wref.add_imu(w["imu"])
    x1 = 2e-3 * np.random.default_rng(23).normal(size=12 * ns)
    for xv, (Hg, gg, cg) in ((x0, (H, g, c0)), (x1, gpu.window_linearize(x1))):
        Hr, gr, cr = Wref.linearize(xv)
        assert abs(cg - cr) <= 1e-10 * cr
        assert np.abs(Hg - Hr).max() <= 1e-9 * np.abs(Hr).max() and np.abs(gg - gr).max() <= 1e-9 * np.abs(gr).max()
        Hb = np.abs(Hg - Hr).reshape(ns, 12, ns, 12).max(axis=(1, 3))
        Hs = np.abs(Hr).reshape(ns, 12, ns, 12).max(axis=(1, 3))
        assert np.all(Hb <= 1e-8 * np.maximum(Hs, 1e-300) + 1e-12 * np.abs(Hr).max())
        assert np.array_equal(Hs == 0, np.abs(Hg).reshape(ns, 12, ns, 12).max(axis=(1, 3)) == 0)  # same block sparsity
    cr1, res_ref = Wref.evaluate(x1, want_residuals=True)
    c1, res1 = gpu.window_evaluate(x1, want_residuals=True)
    assert abs(c1 - cr1) <= 1e-10 * cr1 and np.abs(res1 - res_ref).max() <= 1e-9 * np.abs(res_ref).max()


def test_c4_window_full_size_properties(gpu, oracle):
    """BASELINE config C4 at FULL size (20 sweeps x 50 000 patches = 1 M sliding-window surfels, 50 000 fixed-window surfels,
    IMU factors, ~2 M factors, 127 sample states) - the window bench.py times.  The oracle would need minutes, so: properties
    that hold at any size.  wc_window_counts equals the matcher's counts; the normal equations are symmetric, bitwise
    reproducible and positive on the diagonal; the gauge is free (first sample state long gone, Q12) so no row is zeroed; the
    residual vector has one entry per surfel factor and 12 per IMU factor and reproduces the cost; LM decreases the cost."""
    w = synth.surfel_window(20, 50_000, seed=synth.SEED + 7, fixed_patches=50_000)
    n_s, n_f = len(w["surf"]), len(w["fix_surf"])
    assert n_s == 1_000_000
    d_surf, d_pose = gpu.to_device(w["surf"]), gpu.to_device(w["pose"])
    d_fs, d_fp = gpu.to_device(w["fix_surf"]), gpu.to_device(w["fix_pose"])
    d_pairs, d_pf = gpu.alloc(8 * n_s), gpu.alloc(8 * n_s)
    n_b = gpu.match_device(d_surf, d_pose, n_s, d_surf, d_pose, n_s, True, d_pairs, n_s)
    n_u = gpu.match_device(d_surf, d_pose, n_s, d_fs, d_fp, n_f, False, d_pf, n_s)
    assert n_b > 0.9 * n_s and n_u > 0.9 * n_s
    pairs, pf = d_pairs.download(R.PAIR, n_b), d_pf.download(R.PAIR, n_u)
    t = w["surf"]["t"]
    assert np.all(t[pairs["first"]] < t[pairs["second"]])  # (older, newer)
    assert len(np.unique(pairs.view(np.uint64))) == n_b    # no duplicate pair (the std::set of cc:21,35-39)
    assert np.all(pf["first"] >= 0) and np.all(pf["first"] < n_f) and len(np.unique(pf["second"])) == n_u  # <= 1 per query
    # re-observations of one patch are what gets paired
    pi = w["patch_index"]
    assert np.mean(pi[pairs["first"]] == pi[pairs["second"]]) > 0.99
    gpu.window_build(d_surf, d_pose, d_pairs, n_b, w["imu"], w["sample_times"], w["grav"], False, d_fs, d_fp, d_pf, n_u)
    nb_, nu_, ni_, pieces = gpu.window_counts()
    assert (nb_, nu_) == (n_b, n_u) and ni_ >= len(w["imu"]) - 12 and pieces > 8000
    ns = len(w["sample_times"])
    assert ns >= 120
    x0 = np.zeros(12 * ns)
    H, g, c0 = gpu.window_linearize(x0)
    H2, g2, c02 = gpu.window_linearize(x0)
    assert np.array_equal(H, H2) and np.array_equal(g, g2) and c0 == c02  # fixed reduction order at 12 k pieces too
    assert np.array_equal(H, H.T)
    assert np.all(np.diag(H) > 0) and np.all(np.isfinite(H)) and np.all(np.isfinite(g))
    # block sparsity: a surfel factor couples sample blocks at most a window apart, bias blocks only couple through IMU factors
    cost, res = gpu.window_evaluate(x0, want_residuals=True)
    assert len(res) == n_b + n_u + 12 * ni_ and abs(cost - c0) <= 1e-9 * c0
    assert np.all(np.isfinite(res))
    # IMU factors carry a TrivialLoss (lidar_odometry.cc:342,356): their residuals enter the cost as plain squares; surfel
    # residuals are Cauchy-corrected (sqrt(rho') r), so 1/2 r_c^2 <= 1/2 rho(r^2) for each of them
    assert 0.5 * float(np.dot(res, res)) <= cost * (1 + 1e-12)
    # BY VALUE at full size: the oracle (oracle/window.cc, one factor at a time, dense accumulation) linearises the same 1 M + 1 M
    # surfel factors + IMU factors - seconds on one core - at x = 0 and at a random point; this is the regime of 12 k pieces and
    # gather lists of hundreds of sources that the smaller comparisons never reach (lidar_odometry.cc:254-317 -> cost_functor.h)
    Wref = oracle.Window(w["sample_times"], w["grav"], False)
    Wref.add_binary(w["surf"], w["pose"], pairs)
    Wref.add_unary(w["fix_surf"], w["fix_pose"], w["surf"], w["pose"], pf)
    Wref.add_imu(w["imu"])
    assert Wref.num_residuals() == n_b + n_u + 12 * ni_
    x1 = 2e-3 * np.random.default_rng(17).normal(size=12 * ns)
    for xv, (Hg, gg, cg) in ((x0, (H, g, c0)), (x1, gpu.window_linearize(x1))):
        Hr, gr, cr = Wref.linearize(xv)
        assert abs(cg - cr) <= 1e-10 * cr
        assert np.abs(Hg - Hr).max() <= 1e-9 * np.abs(Hr).max() and np.abs(gg - gr).max() <= 1e-9 * np.abs(gr).max()
        # block by block: every 12 x 12 block pair relative to ITS OWN size (small far-apart blocks are not hidden by the diagonal)
        Hb = np.abs(Hg - Hr).reshape(ns, 12, ns, 12).max(axis=(1, 3))
        Hs = np.abs(Hr).reshape(ns, 12, ns, 12).max(axis=(1, 3))
        assert np.all(Hb <= 1e-8 * np.maximum(Hs, 1e-300) + 1e-12 * np.abs(Hr).max())
        assert np.array_equal(Hs == 0, np.abs(Hg).reshape(ns, 12, ns, 12).max(axis=(1, 3)) == 0)  # same block sparsity
    cr1, res_ref = Wref.evaluate(x1, want_residuals=True)
    c1, res1 = gpu.window_evaluate(x1, want_residuals=True)
    assert abs(c1 - cr1) <= 1e-10 * cr1 and np.abs(res1 - res_ref).max() <= 1e-9 * np.abs(res_ref).max()
    x, s, _ = gpu.window_solve(x0)
    assert s.iterations >= 1 and s.successful_steps >= 1 and s.final_cost < s.initial_cost
    assert abs(s.initial_cost - c0) <= 1e-9 * c0
    _, g_end, c_end = gpu.window_linearize(x)
    assert abs(c_end - s.final_cost) <= 1e-9 * c_end
    assert np.abs(g_end).max() < 1e-2 * np.abs(g).max()


def test_c4_geometry_full_state_count_matches_oracle(gpu, oracle):
    """the C4 window geometry (20 sweeps, 127 sample states = 1 524 unknowns = 48 Cholesky panels, IMU factors) with 1/20 of
    the surfels, so that the CPU oracle finishes in seconds: same iterations, same steps, increments within 1e-6"""
    w, W, keep = _setup(gpu, oracle, n_scans=20, patches=2500, fixed=2500, fix_first=False, extra_mode2=False, seed=7)
    assert W.ns >= 120
    x0 = np.zeros(12 * W.ns)
    x_ref, s_ref, first_ref = W.solve(x0)
    x, s, first = gpu.window_solve(x0)
    assert s.termination == s_ref.termination and s.iterations == s_ref.iterations and s.successful_steps == s_ref.successful_steps
    assert abs(s.final_cost - s_ref.final_cost) <= 1e-8 * s_ref.final_cost
    assert _rel(first, first_ref) <= 1e-6 and _rel(x, x_ref) <= 1e-6
