"""GPU parity of the sweep-preparation kernels against the oracle restatement of
lidar_odometry.cc:489-496 (pre-filter) and :143-158 (UndistortSweep).  Outputs are float32 coordinates."""
import numpy as np
import pytest

from wildcat_slam_amd import records as R
from wildcat_slam_amd import synth

pytestmark = pytest.mark.gpu

EXT_Q = None


def _ext_quat():
    return synth.mat_to_quat(synth.EXT_R[None])[0]


def test_prefilter_matches_oracle(gpu, oracle):
    msgs, imu, _ = synth.raw_stream(0.3, pts_per_s=400_000, t_start=1000.0)
    pts = synth.concat_points(*msgs)
    # add points that must be dropped: inside the blind box, too close, too far
    extra = pts[:300].copy()
    extra["x"][:100], extra["y"][:100], extra["z"][:100] = 0.01, 0.2, -0.1   # -> blind box after the extrinsic
    extra["x"][100:200] *= 0.001
    extra["y"][100:200] *= 0.001
    extra["z"][100:200] *= 0.001
    extra["x"][200:] = 500.0
    pts = synth.concat_points(pts, extra)
    args = (_ext_quat(), synth.EXT_T, 0.3, 120.0, np.array([-0.8, -0.5, -0.4]), np.array([0.3, 0.5, 0.4]))
    ref = oracle.prefilter_points(pts, *args)
    got = gpu.prefilter_points(pts, *args)
    assert 0 < len(ref) < len(pts)
    assert len(got) == len(ref)
    assert np.array_equal(got["time"], ref["time"]) and np.array_equal(got["ring"], ref["ring"])  # same survivors, same order
    for f in ("x", "y", "z"):
        assert np.array_equal(got[f], ref[f])  # the extrinsic is plain fp64 mul/add: bit-exact


def test_undistort_matches_oracle(gpu, oracle):
    pts = synth.g1_room(200_000, t_start=1000.0)
    imu, _ = synth.imu_states(1000.0 - 0.0031, 1000.51, t_origin=1000.0)
    rc, ref = oracle.undistort_sweep(pts, imu)
    assert rc == 0
    got = gpu.undistort_sweep(pts, imu)
    assert np.array_equal(got["time"], ref["time"])
    for f in ("x", "y", "z"):
        # fp64 slerp (sin/acos) may differ in the last ulp between host and device libm -> at most 1 float ulp after the cast
        d = np.abs(got[f].astype(np.float64) - ref[f].astype(np.float64))
        assert d.max() <= np.spacing(np.abs(ref[f]).max().astype(np.float32)) * 1.0
        assert (d > 0).mean() < 1e-3
    from wildcat_slam_amd import lib

    with pytest.raises(lib.WildcatError) as e:  # CHECK(idx >= 1 && idx < size), lidar_odometry.cc:149
        gpu.undistort_sweep(pts, imu[:10])
    assert e.value.code == 2


def test_undistort_packed_is_the_record_version_and_feeds_extraction(gpu, oracle):
    """wc_undistort_sweep_packed leaves the sweep as 3 floats | 1 double per point (what BuildSurfels reads,
    surfel_extraction.cc:317-324): same numbers as the 48-byte version bit for bit, and wc_extract_surfels on the packed arrays
    gives the surfels of the records byte for byte"""
    pts = synth.g1_room(200_000, t_start=1000.0)
    imu, _ = synth.imu_states(1000.0 - 0.0031, 1000.51, t_origin=1000.0)
    rec = gpu.undistort_sweep(pts, imu)
    xyz, t = gpu.undistort_sweep_packed(pts, imu)
    assert np.array_equal(t, rec["time"])
    assert np.array_equal(xyz[:, 0], rec["x"]) and np.array_equal(xyz[:, 1], rec["y"]) and np.array_equal(xyz[:, 2], rec["z"])
    d_xyz, d_t = gpu.undistort_sweep_packed(pts, imu, keep_on_device=True)
    n, cap = len(pts), (3 * len(pts)) // 20 + 1
    d_out, d_ids = gpu.alloc(cap * 144), gpu.alloc(cap * 16)
    gpu.extract_enqueue(R.Points(d_xyz.ptr, d_t.ptr, 12, 8, n), d_out, d_ids, cap, float(t[0]), float(t[-1]))
    m = gpu.extract_finish()
    s_soa, id_soa = d_out.download(R.SURFEL, m), d_ids.download(R.SURFEL_ID, m)
    s_rec, id_rec = gpu.extract_surfels(rec)
    assert m == len(s_rec) > 500 and s_soa.tobytes() == s_rec.tobytes() and id_soa.tobytes() == id_rec.tobytes()
    from wildcat_slam_amd import lib

    with pytest.raises(lib.WildcatError) as e:
        gpu.undistort_sweep_packed(pts, imu[:10])
    assert e.value.code == 2


def test_prefilter_checked_reproduces_the_monotonic_time_check(gpu, oracle):
    """CHECK(points_buff_.empty() || pt.time >= points_buff_.back().time) (lidar_odometry.cc:491) compares EVERY incoming point
    with the last point BUFFERED so far.  A filtered-out point does not move that reference: a point older than a dropped
    predecessor but not older than the last kept one passes (round 2's host loop, which matched kept points by stamp, could be
    stricter here)."""
    msgs, _, _ = synth.raw_stream(0.1, pts_per_s=200_000, t_start=1000.0)
    pts = msgs[0].copy()
    args = (_ext_quat(), synth.EXT_T, 0.3, 120.0, np.array([-0.8, -0.5, -0.4]), np.array([0.3, 0.5, 0.4]))
    got, times, mono = gpu.prefilter_points_checked(pts, *args)
    ref = oracle.prefilter_points(pts, *args)
    assert mono and np.array_equal(got["time"], ref["time"]) and np.array_equal(times, ref["time"])
    # the last buffered point before the message is NEWER than the message's first point: the CHECK fires
    assert not gpu.prefilter_points_checked(pts, *args, prev_time=float(pts["time"][0]) + 1e-3)[2]
    assert gpu.prefilter_points_checked(pts, *args, prev_time=float(pts["time"][0]))[2]
    # point 500 is dropped (far beyond max_range) and carries a stamp in the future; point 501 is older than it but not older
    # than the last KEPT point: the reference does not abort
    a = pts.copy()
    a["x"][500] = 500.0
    a["time"][500] = a["time"][520]
    _, ta, mono_a = gpu.prefilter_points_checked(a, *args)
    assert mono_a and len(ta) == len(oracle.prefilter_points(a, *args)) and not np.all(np.diff(a["time"]) >= 0)
    # a KEPT point newer than its successor: the reference aborts
    b = pts.copy()
    k = 600
    b["time"][k] = b["time"][k + 20]
    keep_b, _, mono_b = gpu.prefilter_points_checked(b, *args)
    assert (b["time"][k] in keep_b["time"]) and not mono_b
