"""End-to-end run of the host facade (C++ LidarOdometry with the reference's public interface, lidar_odometry.h:11-25)
on a synthetic raw sensor stream: lidar messages in the lidar frame + 200 Hz IMU with a gyro bias.  Checks that the
window bookkeeping around the hot path holds together over several sweeps and that the solve pulls the trajectory
towards the truth (IMU dead reckoning alone drifts with the bias)."""
import numpy as np
import pytest

from wildcat_slam_amd import synth

pytestmark = pytest.mark.gpu


def _drive(odo, msgs, imu):
    k = 0
    for m in msgs:
        if len(m) == 0:
            continue
        t_end = m["time"][-1]
        while k < len(imu["t"]) and imu["t"][k] <= t_end + 0.02:
            odo.add_imu(imu["t"][k], imu["acc"][k], imu["gyr"][k])
            k += 1
        odo.add_scan(m)


def test_facade_multi_sweep(gpu):
    from wildcat_slam_amd import lib

    msgs, imu, truth = synth.raw_stream(3.2, pts_per_s=500_000, gyro_bias=(0.0, 0.0, 0.02), t_start=1000.0)
    odo = lib.Odometry(0)
    _drive(odo, msgs, imu)
    assert odo.sweeps() >= 5
    st = odo.stats()
    assert st["sld_surfels"] > 2000 and st["binary"] > 500 and st["lm_iters"] >= 1
    assert st["cost1"] < st["cost0"]
    s = odo.samples()
    p_true, R_true = truth(s[:, 0])
    err = np.linalg.norm(s[:, 1:4] - p_true, axis=1)
    # yaw error of the last sample state vs pure gyro integration of the biased gyro (0.02 rad/s * t)
    q = s[-1, 4:8]
    yaw_est = np.arctan2(2 * (q[0] * q[3] + q[1] * q[2]), 1 - 2 * (q[2] ** 2 + q[3] ** 2))
    yaw_true = np.arctan2(R_true[-1][1, 0], R_true[-1][0, 0])
    t_rel = s[-1, 0] - 1000.0
    print("pos err max", err.max(), "yaw err", yaw_est - yaw_true, "dead-reckoning yaw err would be", 0.02 * t_rel)
    assert abs(yaw_est - yaw_true) < 0.5 * 0.02 * t_rel
    assert err.max() < 0.25
    odo.close()


def test_facade_residual_log(gpu):
    """the reference's residual histograms (PrintSurfelResiduals / PrintImuResiduals, lidar_odometry.cc:56-94) before and after
    every solve, off by default: one line block per factor family, counts = the problem's factor counts"""
    from wildcat_slam_amd import lib

    msgs, imu, _ = synth.raw_stream(2.2, pts_per_s=300_000, gyro_bias=(0.0, 0.0, 0.02), t_start=1000.0)
    odo = lib.Odometry(0)
    _drive(odo, msgs[:8], imu)
    assert odo.residual_log() == ""
    odo.set_residual_log(True)
    k = 0
    while k < len(imu["t"]) and imu["t"][k] <= msgs[7]["time"][-1] + 0.02:
        k += 1
    for m in msgs[8:]:
        while k < len(imu["t"]) and imu["t"][k] <= m["time"][-1] + 0.02:
            odo.add_imu(imu["t"][k], imu["acc"][k], imu["gyr"][k])
            k += 1
        odo.add_scan(m)
    st, log = odo.stats(), odo.residual_log()
    assert odo.sweeps() >= 3 and st["binary"] > 100
    for when in ("[before solve] ", "[after update] "):
        head = when + "Sliding Window Surfel residuals, cost: "
        assert head in log
        blk = log[log.index(head):]
        assert "dist: Count: %d  Min: " % int(st["binary"]) in blk.split("\n")[0]
        assert float(blk[len(head):].split(",")[0]) > 0
        for kind in ("gyro", "acc", "gyro_bias", "acc_bias"):
            assert when + "Imu residuals with type " + kind + ", cost: " in log
    assert "Fixed Window" not in log  # (no fixed window yet: PrintSurfelResiduals returns on an empty block list, :57-59)
    import re

    assert len(re.findall(r"\n\[-?[0-9.]+, -?[0-9.]+[)\]]\t", log)) == 2 * 5 * 10  # ten buckets per histogram, five histograms, twice
    odo.close()


def test_facade_outputs_are_what_the_reference_publishes(gpu):
    """config().fill_outputs: after a sweep LidarOdometry::last_outputs() holds what the reference hands to ROS at
    lidar_odometry.cc:582-602 - one marker per sliding-window surfel (PubSurfels), the sweep undistorted with the FINAL poses as
    a PointCloud2 payload stamped with its first point, and the world -> imu_link transform of the last sample state"""
    from wildcat_slam_amd import lib

    msgs, imu, truth = synth.raw_stream(1.7, pts_per_s=300_000, gyro_bias=(0.0, 0.0, 0.02), t_start=1000.0)
    odo = lib.Odometry(0)
    odo.set_fill_outputs(True)
    _drive(odo, msgs, imu)
    assert odo.sweeps() >= 2
    o, st, s = odo.outputs(), odo.stats(), odo.samples()
    assert len(o["markers"]) == int(st["sld_surfels"]) > 500
    assert np.allclose(np.linalg.norm(o["markers"][:, 3:7], axis=1), 1.0, atol=1e-12) and np.all(o["markers"][:, 13] == 1.0)
    sc = o["markers"][:, 7:10]
    assert np.all(sc[np.isfinite(sc)] >= 0) and np.nanmin(sc[:, 0]) < 0.15  # 3 sigma of a plane patch along its normal: centimetres
    # the sweep in the world frame: points of the room the synthetic scanner stands in (walls at +-20 m)
    pts = o["scan"]
    assert len(pts) > 100_000 and o["stamp"] == pts["time"][0] and np.all(np.diff(pts["time"]) >= 0)
    assert np.abs(pts["x"]).max() < 21.5 and np.abs(pts["y"]).max() < 21.5
    # tf = the last sample state, quaternion in tf's (x, y, z, w) order
    assert o["tf"][0] == s[-1, 0] and np.array_equal(o["tf"][1:4], s[-1, 1:4])
    assert np.array_equal(o["tf"][4:8], s[-1, [5, 6, 7, 4]])
    p_true, _ = truth(np.array([s[-1, 0]]))
    assert np.linalg.norm(o["tf"][1:4] - p_true[0]) < 0.25
    odo.close()


def _feed(odo, ref, msgs, imu, on_sweep):
    """the same raw stream into the facade and the orchestrated oracle; on_sweep(sweep number) after every completed sweep"""
    k = 0
    for m in msgs:
        if len(m) == 0:
            continue
        t_end = m["time"][-1]
        while k < len(imu["t"]) and imu["t"][k] <= t_end + 0.02:
            odo.add_imu(imu["t"][k], imu["acc"][k], imu["gyr"][k])
            ref.add_imu(imu["t"][k], imu["acc"][k], imu["gyr"][k])
            k += 1
        before = ref.sweeps()
        odo.add_scan(m)
        ref.add_scan(m)
        assert odo.sweeps() == ref.sweeps()
        if ref.sweeps() != before:
            on_sweep(ref.sweeps())


def _pair_set_difference(a, b, tol=1e-5):
    """pairs of one run that have no partner in the other, matched on their two surfel TIMESTAMPS within tol (surfel indices mean
    nothing across arithmetics: the default path's stamps are the correctly rounded means, <= 2e-6 s from the reference's running
    sums, and two surfels closer than that swap places).  a, b: float64[n, 2] -> (only in a, only in b)"""
    a, b = a[np.lexsort((a[:, 1], a[:, 0]))], b[np.lexsort((b[:, 1], b[:, 0]))]
    if len(a) == len(b) and (len(a) == 0 or np.abs(a - b).max() <= tol):
        return 0, 0  # the rule: same pairs in the same order
    # the exception (two first surfels within 2e-6 s of each other sort differently, or a pair is missing): match by buckets
    import collections

    w = max(4 * tol, 1e-9)
    left = collections.defaultdict(list)
    for t0, t1 in b:
        left[(int(t0 // w), int(t1 // w))].append((t0, t1))
    only_a = 0
    for t0, t1 in a:
        k0, k1 = int(t0 // w), int(t1 // w)
        hit = False
        for d0 in (-1, 0, 1):
            for d1 in (-1, 0, 1):
                lst = left.get((k0 + d0, k1 + d1))
                if lst:
                    for i, (u0, u1) in enumerate(lst):
                        if abs(u0 - t0) <= tol and abs(u1 - t1) <= tol:
                            lst.pop(i)
                            hit = True
                            break
                if hit:
                    break
            if hit:
                break
        only_a += 0 if hit else 1
    return only_a, sum(len(v) for v in left.values())


def _state_diff(a, b):
    return max(np.abs(a[:, 1:4] - b[:, 1:4]).max(), np.abs(a[:, 4:8] - b[:, 4:8]).max(), np.abs(a[:, 8:14] - b[:, 8:14]).max())


@pytest.mark.parametrize("exact_sums", [0, 1], ids=["default-arithmetic", "exact-sums"])
@pytest.mark.parametrize("quirks", [1, 0], ids=["quirks", "no-quirks"])
def test_facade_each_sweep_against_the_oracle_resynchronised(gpu, oracle, exact_sums, quirks):
    """the gate of the drop-in class (AddLidarScan, lidar_odometry.cc:487-605, hot block :523-566): the host facade against the
    orchestrated oracle, EVERY sweep held to north_star's 1e-6 on its own.  After each comparison the oracle's sample / IMU
    states are copied into the facade, so a solve never inherits the last-bit drift of the solves before it (each ends at Ceres'
    function tolerance, and a free-running chain amplifies that - see the drift report below).  Runs in the facade's DEFAULT
    extraction arithmetic (integer moments, the kernels bench.py measures) and in the reference's summation order, with the
    reference's quirks (Q1/Q3 Jacobians, Q11 fixed window) and without.  8.2 s > the 6 s sliding window: ShrinkToFit moves
    surfels into the fixed window and unary factors appear."""
    from wildcat_slam_amd import lib

    msgs, imu, _ = synth.raw_stream(8.2, pts_per_s=150_000, gyro_bias=(0.0, 0.0, 0.02), t_start=1000.0)
    odo, ref = lib.Odometry(0), oracle.Odometry()
    odo.set_exact_sums(exact_sums)
    odo.set_quirks(quirks)
    ref.set_quirks(quirks)
    odo.set_keep_pair_stamps(True)
    per_sweep, last, pair_diff, pair_sets = [], {}, [], []

    def on_sweep(k):
        a, b = odo.samples(), ref.samples()
        sa, sb = odo.stats(), ref.stats()
        assert a.shape == b.shape and np.array_equal(a[:, 0], b[:, 0])  # same sample states, same timestamps
        for key in ("sld_surfels", "fix_surfels", "lm_iters", "termination"):
            assert sa[key] == sb[key], (k, key, sa[key], sb[key])
        for key in ("binary", "unary"):
            # exact sums: the oracle's surfels in the oracle's order -> the same correspondences.  Default arithmetic: surfel stamps
            # are the correctly rounded means (<= 2e-6 s from the reference's running sums), so two surfels closer in time than that
            # may swap places - which cannot change a pair: a surfel and any of its candidates are >= 0.06 s apart (the time gate,
            # knn_surfel_matcher.cc:26), so the order of a surfel and its candidates, all the de-duplication (cc:35-38) looks at, is
            # the same in both runs.  What can differ is a gate within 2e-6 s / 3e-10 of its threshold: measured 0 in all 17 sweeps
            # with and without the quirks; two are allowed.
            assert sa[key] == sb[key] if exact_sums else abs(sa[key] - sb[key]) <= 2, (k, key, sa[key], sb[key])
        # ... and the pair SETS themselves (equal counts do not show equal pairs): every correspondence of the
        # facade has its partner in the oracle's list - the same two surfels, identified by their timestamps - and vice versa, up
        # to the two threshold cases allowed above
        for which in (0, 1):
            only_f, only_o = _pair_set_difference(odo.pair_stamps(which), ref.pair_stamps(which), 0.0 if exact_sums else 1e-5)
            pair_sets.append((k, which, only_f, only_o))
            assert only_f + only_o <= (0 if exact_sums else 4), (k, which, only_f, only_o)
        d = _state_diff(a, b)
        per_sweep.append((k, float("%.2g" % d)))
        pair_diff.append((int(sa["binary"] - sb["binary"]), int(sa["unary"] - sb["unary"])))
        assert d <= 1e-6, per_sweep
        assert abs(sa["cost1"] - sb["cost1"]) <= 1e-5 * max(1.0, abs(sb["cost1"]))
        ft, rt = odo.fixed_times(), ref.window_times(True)
        assert len(ft) == len(rt) and (np.array_equal(ft, rt) if exact_sums else np.abs(ft - rt).max() <= 1e-5 if len(ft) else True)
        if len(ft) > 1:
            assert np.all(np.diff(ft) <= (0 if exact_sums else 1e-5))  # newest first (Q11)
        odo.import_state(*ref.export_state())  # re-synchronise: the next sweep starts from the oracle's states
        last.update(sb)

    _feed(odo, ref, msgs, imu, on_sweep)
    fast, exact = odo.extract_paths()
    print("arithmetic", "exact" if exact_sums else "default", "quirks", quirks, "sweeps on the fast / exact path", fast, exact,
          "per-sweep worst sample-state difference", per_sweep, "correspondence-count differences (binary, unary) per sweep", pair_diff,
          "correspondences without a partner in the other run (sweep, family, facade only, oracle only), non-zero entries", [p for p in pair_sets if p[2] or p[3]])
    # the arithmetic asked for is the one that ran (a default-arithmetic sweep may fall back when a gate lies in the noise band)
    assert fast + exact == len(per_sweep) and (fast == 0 if exact_sums else fast >= 0.8 * len(per_sweep))
    assert len(per_sweep) >= 15 and last["fix_surfels"] > 1000 and last["unary"] > 1000
    odo.close()
    ref.close()


@pytest.mark.parametrize("exact_sums", [0, 1], ids=["default-arithmetic", "exact-sums"])
def test_facade_free_running_drift_report(gpu, oracle, exact_sums):
    """the same comparison WITHOUT re-synchronisation: a drift report, not a parity gate.  Each solve ends at Ceres' function
    tolerance (1e-6), so two runs that differ in the last bits drift apart by about that much per sweep - and once the 1e-8 of
    pose difference has put one point on the other side of a voxel face (one surfel more or less: seen at sweep 14 of the
    default arithmetic), by the weight of a surfel: 1e-4.  The band only catches a facade that walks away from the oracle
    (wrong window bookkeeping); the 1e-6 gate is the re-synchronised test above."""
    from wildcat_slam_amd import lib

    msgs, imu, _ = synth.raw_stream(8.2, pts_per_s=150_000, gyro_bias=(0.0, 0.0, 0.02), t_start=1000.0)
    odo, ref = lib.Odometry(0), oracle.Odometry()
    odo.set_exact_sums(exact_sums)
    per_sweep = []

    def on_sweep(k):
        a, b = odo.samples(), ref.samples()
        sa, sb = odo.stats(), ref.stats()
        assert a.shape == b.shape and np.array_equal(a[:, 0], b[:, 0])
        # (a free-running pair of runs may come to differ by a surfel: poses that differ by 1e-8 put a point within that of a voxel
        # face, or a gate within it of its threshold, on the other side - reported, and bounded, not a parity failure)
        d_surf = abs(sa["sld_surfels"] - sb["sld_surfels"]) + abs(sa["fix_surfels"] - sb["fix_surfels"])
        per_sweep.append((k, float("%.2g" % _state_diff(a, b)), int(d_surf)))
        if k <= 7:
            # the parity gate of the free-running pair: through sweep 7 the chain carries nothing but rounding - same
            # surfels, same LM iterations, same termination, states within 1e-7 (measured: <= 9e-9); in sweep 8 one of the sweep's
            # solves ends on the other side of Ceres' function tolerance in the two runs (3.4e-5)
            assert d_surf == 0 and sa["lm_iters"] == sb["lm_iters"] and sa["termination"] == sb["termination"] and per_sweep[-1][1] <= 1e-7, per_sweep
        assert per_sweep[-1][1] <= (1e-6 * 2 ** min(k, 8) if d_surf == 0 and all(p[2] == 0 for p in per_sweep) else 1e-3) and d_surf <= 8, per_sweep

    _feed(odo, ref, msgs, imu, on_sweep)
    print("arithmetic", "exact" if exact_sums else "default", "free-running drift per sweep", per_sweep)
    assert len(per_sweep) >= 15
    odo.close()
    ref.close()
