"""ctypes binding of oracle/libwc_oracle.so — TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG = os.path.join(os.path.dirname(_HERE), "wildcat-slam_amd", "python")
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)
from wildcat_slam_amd import records as R  # noqa: E402

_LIB = None


def build():
    subprocess.run(["make", "-C", _HERE, "libwc_oracle.so"], check=True, stdout=subprocess.DEVNULL)


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libwc_oracle.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        _LIB.wco_window_create.restype = C.c_void_p
        _LIB.wco_window_num_residuals.restype = C.c_uint64
    return _LIB


def default_params():
    p = R.Params()
    lib().wco_params_default(C.byref(p))
    return p


class ExtractStats(C.Structure):
    _fields_ = [
        ("root_voxels", C.c_uint64),
        ("nodes_tested", C.c_uint64 * 4),
        ("nodes_plane", C.c_uint64 * 4),
        ("clusters_total", C.c_uint64),
        ("clusters_rejected", C.c_uint64),
        ("surfels", C.c_uint64),
        ("min_gate_margin", C.c_double),
    ]


def _vec(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def so3_exp(w):
    out = np.zeros(4)
    lib().wco_so3_exp(R.ptr(_vec(w)), R.ptr(out))
    return out


def so3_log(q):
    out = np.zeros(3)
    lib().wco_so3_log(R.ptr(_vec(q)), R.ptr(out))
    return out


def _m3(fn, v):
    out = np.zeros(9)
    getattr(lib(), fn)(R.ptr(_vec(v)), R.ptr(out))
    return out.reshape(3, 3)


def so3_jl(v):
    return _m3("wco_so3_jl", v)


def so3_jl_inv(v):
    return _m3("wco_so3_jl_inv", v)


def so3_jr(v):
    return _m3("wco_so3_jr", v)


def so3_jr_inv(v):
    return _m3("wco_so3_jr_inv", v)


def eig3(a):
    a = _vec(a).reshape(9)
    ev, V = np.zeros(3), np.zeros(9)
    lib().wco_eig3(R.ptr(a), R.ptr(ev), R.ptr(V))
    return ev, V.reshape(3, 3)


def voxel_keys(points, params=None):
    params = params or default_params()
    d = R.points_from_aos(points)
    keys = np.zeros((len(points), 3), np.int32)
    lib().wco_voxel_keys(C.byref(d), C.byref(params), R.ptr(keys))
    return keys


def extract_surfels(points, params=None, cap=None):
    """-> (surfels[SURFEL], ids[SURFEL_ID], stats)"""
    params = params or default_params()
    d = R.points_from_aos(points)
    cap = cap or max(1024, len(points) // 8)
    out = np.zeros(cap, R.SURFEL)
    ids = np.zeros(cap, R.SURFEL_ID)
    n = C.c_uint64(0)
    st = ExtractStats()
    rc = lib().wco_extract_surfels(C.byref(d), C.byref(params), R.ptr(out), R.ptr(ids), C.c_uint64(cap), C.byref(n), C.byref(st))
    if rc == 1:
        return extract_surfels(points, params, cap=int(n.value))
    assert rc == 0, rc
    return out[: n.value].copy(), ids[: n.value].copy(), st


def update_surfel_poses(imu, surf, pose, in_body):
    rc = lib().wco_update_surfel_poses(R.ptr(imu), C.c_uint64(len(imu)), R.ptr(surf), R.ptr(pose), R.ptr(in_body), C.c_uint64(len(surf)))
    return rc


def prefilter_points(points, ext_quat, ext_t, min_range, max_range, blind_min, blind_max):
    out = np.zeros(len(points), R.POINT)
    n = C.c_uint64(0)
    lib().wco_prefilter_points(R.ptr(points), C.c_uint64(len(points)), R.ptr(_vec(ext_quat)), R.ptr(_vec(ext_t)), C.c_double(min_range),
                               C.c_double(max_range), R.ptr(_vec(blind_min)), R.ptr(_vec(blind_max)), R.ptr(out), C.byref(n))
    return out[: n.value].copy()


def undistort_sweep(points, imu):
    out = np.zeros(len(points), R.POINT)
    rc = lib().wco_undistort_sweep(R.ptr(points), C.c_uint64(len(points)), R.ptr(imu), C.c_uint64(len(imu)), R.ptr(out))
    return rc, out


def knn6(cloud, query, k=10):
    cloud = np.ascontiguousarray(cloud, np.float64)
    query = np.ascontiguousarray(query, np.float64)
    idx = np.zeros((len(query), k), np.int32)
    d2 = np.zeros((len(query), k))
    lib().wco_knn6(R.ptr(cloud), C.c_uint64(len(cloud)), R.ptr(query), C.c_uint64(len(query)), C.c_int(k), R.ptr(idx), R.ptr(d2))
    return idx, d2


def match(q_surf, q_pose, t_surf, t_pose, same_set, params=None):
    params = params or default_params()
    cap = len(q_surf)
    pairs = np.zeros(max(cap, 1), R.PAIR)
    n = C.c_uint64(0)
    rc = lib().wco_match(C.byref(params), R.ptr(q_surf), R.ptr(q_pose), C.c_uint64(len(q_surf)), R.ptr(t_surf), R.ptr(t_pose),
                         C.c_uint64(len(t_surf)), C.c_int(1 if same_set else 0), R.ptr(pairs), C.c_uint64(cap), C.byref(n))
    assert rc == 0, rc
    return pairs[: n.value].copy()


class Window:
    """wco_window wrapper (the ceres::Problem of lidar_odometry.cc:541-545)."""

    def __init__(self, sample_times, grav, fix_first_pos, params=None):
        self.params = params or default_params()
        self.times = _vec(sample_times)
        self.ns = len(self.times)
        self._h = C.c_void_p(lib().wco_window_create(C.byref(self.params), R.ptr(self.times), C.c_uint64(self.ns), R.ptr(_vec(grav)), C.c_int(int(fix_first_pos))))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().wco_window_destroy(self._h)
            self._h = None

    def add_binary(self, surf, pose, pairs):
        rc = lib().wco_window_add_binary(self._h, R.ptr(surf), R.ptr(pose), R.ptr(pairs), C.c_uint64(len(pairs)))
        assert rc == 0, rc

    def add_unary(self, fix_surf, fix_pose, sld_surf, sld_pose, pairs):
        rc = lib().wco_window_add_unary(self._h, R.ptr(fix_surf), R.ptr(fix_pose), R.ptr(sld_surf), R.ptr(sld_pose), R.ptr(pairs), C.c_uint64(len(pairs)))
        assert rc == 0, rc

    def add_imu(self, imu):
        rc = lib().wco_window_add_imu(self._h, R.ptr(imu), C.c_uint64(len(imu)))
        assert rc == 0, rc

    def counts(self):
        c = (C.c_uint64 * 6)()
        lib().wco_window_counts(self._h, c)
        return list(c)

    def num_residuals(self):
        return int(lib().wco_window_num_residuals(self._h))

    def evaluate(self, x, want_residuals=False):
        x = _vec(x)
        cost = C.c_double(0)
        res = np.zeros(self.num_residuals()) if want_residuals else None
        lib().wco_window_evaluate(self._h, R.ptr(x), C.byref(cost), R.ptr(res) if want_residuals else None)
        return (cost.value, res) if want_residuals else cost.value

    def linearize(self, x):
        x = _vec(x)
        n = 12 * self.ns
        H, g = np.zeros((n, n)), np.zeros(n)
        cost = C.c_double(0)
        lib().wco_window_linearize(self._h, R.ptr(x), R.ptr(H), R.ptr(g), C.byref(cost))
        return H, g, cost.value

    def solve(self, x):
        x = _vec(x).copy()
        s = R.SolveSummary()
        first = np.zeros(12 * self.ns)
        lib().wco_window_solve(self._h, R.ptr(x), C.byref(s), R.ptr(first))
        return x, s, first


def odometry_step(ws, params=None):
    """The hot block of LidarOdometry::AddLidarScan (lidar_odometry.cc:523-566) on a scan sequence `ws` (synth.g2_scan_sequence): the
    sweeps before the newest are the window (extracted, posed; the two oldest form the fixed window - outside the timed part), the
    step takes the newest sweep: BuildSurfels -> UpdateSurfelPoses -> 2 x KnnSurfelMatcher -> problem construction -> solve ->
    UpdateSurfelPoses.  ONE implementation for bench.py's cpu_baseline leg of `odometry_step` and for the parity tests of
    wildcat_slam_amd/step.py (tests/test_step_gpu.py).  -> dict: surfels / ids per sweep, the window after the step, both pair lists,
    the solve's summary and corrections, seconds of the step itself"""
    import time

    prm = params or default_params()
    surf, ids = [], []
    for s in ws["scans"][:-1]:
        a, b, _ = extract_surfels(s, prm)
        surf.append(a)
        ids.append(b)
    n_fx = len(surf[0]) + len(surf[1])
    S = np.concatenate(surf)
    P = np.zeros(len(S), R.POSE)
    B = np.zeros(len(S), np.uint8)
    update_surfel_poses(ws["imu"], S, P, B)
    t0 = time.perf_counter()
    new, new_ids, _ = extract_surfels(ws["scans"][-1], prm)
    S2 = np.concatenate([S, new])
    P2 = np.concatenate([P, np.zeros(len(new), R.POSE)])
    B2 = np.concatenate([B, np.zeros(len(new), np.uint8)])
    sl_s, sl_p, sl_b = np.ascontiguousarray(S2[n_fx:]), np.ascontiguousarray(P2[n_fx:]), np.ascontiguousarray(B2[n_fx:])
    update_surfel_poses(ws["imu"], sl_s, sl_p, sl_b)
    fx_s, fx_p = np.ascontiguousarray(S2[:n_fx]), np.ascontiguousarray(P2[:n_fx])
    pb = match(sl_s, sl_p, sl_s, sl_p, True, prm)
    pu = match(sl_s, sl_p, fx_s, fx_p, False, prm)
    Wc = Window(ws["sample_times"], ws["grav"], False, prm)
    Wc.add_binary(sl_s, sl_p, pb)
    Wc.add_unary(fx_s, fx_p, sl_s, sl_p, pu)
    Wc.add_imu(ws["imu"])
    x, summ, _ = Wc.solve(np.zeros(12 * len(ws["sample_times"])))
    update_surfel_poses(ws["imu"], sl_s, sl_p, sl_b)
    t_step = time.perf_counter() - t0
    return dict(ids=ids + [new_ids], n_fix=n_fx, new=len(new), sld_surf=sl_s, sld_pose=sl_p, fix_surf=fx_s, fix_pose=fx_p, pairs_sld=pb, pairs_fix=pu, x=x,
                summary=summ, seconds=t_step, points=len(ws["scans"][-1]))


def bspline_fit_eval(timestamps, points, query_t):
    timestamps, points, query_t = _vec(timestamps), _vec(points), _vec(query_t)
    out = np.zeros((len(query_t), 3))
    valid = np.zeros(len(query_t), np.uint8)
    lib().wco_bspline_fit_eval(R.ptr(timestamps), R.ptr(points), C.c_uint64(len(timestamps)), R.ptr(query_t), C.c_uint64(len(query_t)), R.ptr(out), R.ptr(valid))
    return out, valid.astype(bool)


def update_imu_poses(sample_times, x, ba, bg, grav, imu):
    rc = lib().wco_update_imu_poses(R.ptr(_vec(sample_times)), R.ptr(_vec(x)), C.c_uint64(len(sample_times)), R.ptr(_vec(ba)), R.ptr(_vec(bg)), R.ptr(_vec(grav)), R.ptr(imu), C.c_uint64(len(imu)))
    return rc


class Odometry:
    """the orchestrated oracle (oracle/odometry.cc): AddImuData / AddLidarScan of lidar_odometry.cc:487-611 on the oracle stages"""

    def __init__(self):
        l = lib()
        l.wco_odom_create.restype = C.c_void_p
        l.wco_odom_num_samples.restype = C.c_uint64
        l.wco_odom_window_times.restype = C.c_uint64
        self._h = C.c_void_p(l.wco_odom_create())

    def close(self):
        if self._h:
            lib().wco_odom_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def add_imu(self, t, acc, gyr):
        a, g = (C.c_double * 3)(*acc), (C.c_double * 3)(*gyr)
        lib().wco_odom_add_imu(self._h, C.c_double(t), a, g)

    def add_scan(self, points):
        assert points.dtype == R.POINT
        points = np.ascontiguousarray(points)
        lib().wco_odom_add_scan(self._h, R.ptr(points), C.c_uint64(len(points)))
        err = lib().wco_odom_error(self._h)
        assert err == 0, f"a reference CHECK would have fired (oracle/odometry.cc:{err})"

    def set_quirks(self, on):
        lib().wco_odom_set_quirks(self._h, C.c_int(1 if on else 0))

    def export_state(self):
        """(samples (ns, 23): timestamp, cor[12], grav[3], quat[4], pos[3]; imu states as R.IMU_STATE records)"""
        cnt = (C.c_uint64 * 2)()
        lib().wco_odom_export_state(self._h, None, C.c_uint64(0), None, C.c_uint64(0), cnt)
        s, imu = np.zeros((max(cnt[0], 1), 23)), np.zeros(max(cnt[1], 1), dtype=R.IMU_STATE)
        lib().wco_odom_export_state(self._h, R.ptr(s), C.c_uint64(cnt[0]), R.ptr(imu), C.c_uint64(cnt[1]), cnt)
        return s[: cnt[0]], imu[: cnt[1]]

    def sweeps(self):
        return int(lib().wco_odom_sweeps(self._h))

    def samples(self):
        n = int(lib().wco_odom_num_samples(self._h))
        out = np.zeros((n, 15))
        for i in range(n):
            lib().wco_odom_sample(self._h, C.c_uint64(i), R.ptr(out[i]))
        return out

    def stats(self):
        s = np.zeros(10)
        lib().wco_odom_stats(self._h, R.ptr(s))
        return dict(zip(("sld_surfels", "fix_surfels", "binary", "unary", "lm_iters", "cost0", "cost1", "termination", "new_surfels", "imu_states"), s.tolist()))

    def pair_stamps(self, which):
        """(first, second) surfel timestamps of the last sweep's correspondences (0: sliding, 1: fixed window) -> float64[n, 2]"""
        lib().wco_odom_pair_stamps.restype = C.c_uint64
        n = int(lib().wco_odom_pair_stamps(self._h, C.c_int(which), None, C.c_uint64(0)))
        out = np.zeros(max(n, 2))
        lib().wco_odom_pair_stamps(self._h, C.c_int(which), R.ptr(out), C.c_uint64(n))
        return out[:n].reshape(-1, 2)

    def window_times(self, fixed):
        n = int(lib().wco_odom_window_times(self._h, C.c_int(1 if fixed else 0), None, C.c_uint64(0)))
        out = np.zeros(max(n, 1))
        lib().wco_odom_window_times(self._h, C.c_int(1 if fixed else 0), R.ptr(out), C.c_uint64(n))
        return out[:n]
