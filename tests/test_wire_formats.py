"""The ROS-free half of the reference's wire / visualisation formats (host/wire_formats.h): what
pcl::fromROSMsg / pcl::toROSMsg do for the registered point type (src/common/common.h:12-28, used at wildcat_slam_node.cc:46-52 and
lidar_odometry.cc:584-595), makeRightHanded and the RViz marker of a surfel (surfel_extraction.cc:340-434).  No GPU needed."""
import ctypes as C
import os

import numpy as np
import pytest

from wildcat_slam_amd import records as R
from wildcat_slam_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
FLOAT32, FLOAT64, UINT16, UINT32 = 7, 8, 4, 6


def _quat_to_mat(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


@pytest.fixture(scope="module")
def host():
    so = os.path.join(HERE, "..", "wildcat-slam_amd", "host", "libwildcat_odometry.so")
    return C.CDLL(os.path.abspath(so))


def _to_points(host, fields, width, height, point_step, data, row_step=0, big=False):
    names = b"".join(f[0].encode() + b"\0" for f in fields)
    offs = (C.c_uint32 * len(fields))(*[f[1] for f in fields])
    dts = (C.c_uint8 * len(fields))(*[f[2] for f in fields])
    cnt = (C.c_uint32 * len(fields))(*[f[3] for f in fields])
    out = np.zeros(width * height, R.POINT)
    raw = np.frombuffer(bytes(data), np.uint8)
    m = host.wc_host_cloud2_to_points(names, offs, dts, cnt, C.c_int(len(fields)), C.c_uint32(width), C.c_uint32(height), C.c_uint32(point_step),
                                      C.c_uint32(row_step), C.c_int(1 if big else 0), raw.ctypes.data_as(C.c_void_p), C.c_uint64(len(raw)),
                                      out.ctypes.data_as(C.c_void_p))
    return m, out


def test_pointcloud2_of_a_hesai_driver_layout(host):
    """a driver's own layout (other offsets, an extra field, point_step 34): every registered field is found by NAME and type"""
    n = 100
    rng = np.random.default_rng(3)
    drv = np.dtype({"names": ["x", "y", "z", "intensity", "timestamp", "ring", "azimuth"], "formats": ["<f4", "<f4", "<f4", "<f4", "<f8", "<u2", "<f4"],
                    "offsets": [0, 4, 8, 12, 16, 24, 26], "itemsize": 34})
    a = np.zeros(n, drv)
    for f in ("x", "y", "z", "intensity", "azimuth"):
        a[f] = rng.normal(size=n).astype(np.float32)
    a["timestamp"] = 1.6e9 + np.arange(n) * 1e-5
    a["ring"] = np.arange(n) % 32
    fields = [("x", 0, FLOAT32, 1), ("y", 4, FLOAT32, 1), ("z", 8, FLOAT32, 1), ("intensity", 12, FLOAT32, 1), ("timestamp", 16, FLOAT64, 1),
              ("ring", 24, UINT16, 1), ("azimuth", 26, FLOAT32, 1)]
    m, pts = _to_points(host, fields, n, 1, 34, a.tobytes())
    assert m == 6
    for f in ("x", "y", "z", "intensity"):
        assert np.array_equal(pts[f], a[f])
    assert np.array_equal(pts["time"], a["timestamp"]) and np.array_equal(pts["ring"], a["ring"])
    assert np.all(pts.view(np.uint8).reshape(n, 48)[:, 12:16].view(np.float32) == 0.0)  # value-initialised points: the padding word stays 0
    # organised cloud with padded rows: row_step > width * point_step
    rows = np.zeros((2, 50 * 34 + 6), np.uint8)
    rows[0, : 50 * 34] = np.frombuffer(a[:50].tobytes(), np.uint8)
    rows[1, : 50 * 34] = np.frombuffer(a[50:].tobytes(), np.uint8)
    m2, pts2 = _to_points(host, fields, 50, 2, 34, rows.tobytes(), row_step=50 * 34 + 6)
    assert m2 == 6 and pts2.tobytes() == pts.tobytes()


def test_pointcloud2_fields_pcl_would_not_match(host):
    """pcl::FieldMatches wants the same name AND datatype: a float32 `timestamp` or a uint32 `ring` is NOT taken (PCL warns and
    leaves the member as constructed), a message without `intensity` still converts; a big-endian payload is refused"""
    n = 8
    d = np.zeros(n, np.dtype({"names": ["x", "y", "z", "timestamp", "ring"], "formats": ["<f4", "<f4", "<f4", "<f4", "<u4"], "offsets": [0, 4, 8, 12, 16],
                              "itemsize": 20}))
    d["x"], d["timestamp"], d["ring"] = np.arange(n), 5.0, 7
    fields = [("x", 0, FLOAT32, 1), ("y", 4, FLOAT32, 1), ("z", 8, FLOAT32, 1), ("timestamp", 12, FLOAT32, 1), ("ring", 16, UINT32, 1)]
    m, pts = _to_points(host, fields, n, 1, 20, d.tobytes())
    assert m == 3 and np.array_equal(pts["x"], np.arange(n, dtype=np.float32))
    assert not pts["time"].any() and not pts["ring"].any() and not pts["intensity"].any()
    assert _to_points(host, fields, n, 1, 20, d.tobytes(), big=True)[0] == -1
    assert _to_points(host, fields, n, 1, 20, d.tobytes()[:-4])[0] == -1  # payload shorter than its description
    # two fields of one name: PCL's mapping takes the first
    d["y"] = 3.0
    twice = [("x", 0, FLOAT32, 1), ("x", 4, FLOAT32, 1), ("z", 8, FLOAT32, 1)]
    m, pts = _to_points(host, twice, n, 1, 20, d.tobytes())
    assert m == 2 and np.array_equal(pts["x"], np.arange(n, dtype=np.float32))


def test_points_to_pointcloud2_layout_and_round_trip(host):
    """pcl::toROSMsg: the struct bytes as they lie, fields in registration order with the struct's offsets"""
    table = (C.c_uint32 * 18)()
    names = C.create_string_buffer(64)
    step = host.wc_host_points_to_cloud2_layout(table, names, C.c_uint64(64))
    assert step == 48
    assert names.raw.split(b"\0")[:6] == [b"x", b"y", b"z", b"intensity", b"timestamp", b"ring"]
    assert list(table) == [0, FLOAT32, 1, 4, FLOAT32, 1, 8, FLOAT32, 1, 16, FLOAT32, 1, 24, FLOAT64, 1, 32, UINT16, 1]
    pts = synth.g1_room(500)
    fields = [(nm.decode(), table[3 * i], table[3 * i + 1], table[3 * i + 2]) for i, nm in enumerate(names.raw.split(b"\0")[:6])]
    m, back = _to_points(host, fields, len(pts), 1, 48, pts.tobytes())
    assert m == 6
    for f in ("x", "y", "z", "intensity", "time", "ring"):
        assert np.array_equal(back[f], pts[f])


def test_make_right_handed(host):
    """surfel_extraction.cc:340-358: columns normalised; a left-handed frame swaps its first two columns AND eigenvalues"""
    V = np.array([[0.0, 2.0, 0.0], [3.0, 0.0, 0.0], [0.0, 0.0, 0.5]])  # columns: 3 e_y, 2 e_x, 0.5 e_z -> (e_y x e_x) . e_z = -1
    ev = np.array([1.0, 2.0, 3.0])
    v, e = V.copy().reshape(-1), ev.copy()
    host.wc_host_make_right_handed(v.ctypes.data_as(C.c_void_p), e.ctypes.data_as(C.c_void_p))
    assert np.array_equal(v.reshape(3, 3), np.array([[1.0, 0, 0], [0, 1.0, 0], [0, 0, 1.0]])) and np.array_equal(e, [2.0, 1.0, 3.0])
    v2, e2 = np.eye(3).reshape(-1) * 4.0, ev.copy()
    host.wc_host_make_right_handed(v2.ctypes.data_as(C.c_void_p), e2.ctypes.data_as(C.c_void_p))
    assert np.array_equal(v2.reshape(3, 3), np.eye(3)) and np.array_equal(e2, ev)


def test_surfel_marker(host):
    """PubSurfels (surfel_extraction.cc:360-408): position = world centre, orientation = right-handed eigen-frame of the WORLD
    covariance, scale = 3 sqrt(eigenvalue), colour = (world normal + 1) / 2, alpha 1"""
    rng = np.random.default_rng(11)
    for _ in range(20):
        A = rng.normal(size=(3, 3)) * np.array([0.3, 0.1, 0.01])
        cov = A @ A.T
        w, U = np.linalg.eigh(cov)
        s = np.zeros(1, R.SURFEL)
        s["center"], s["cov"], s["normal"] = rng.normal(size=3), cov.reshape(-1), U[:, 0]
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        p = np.zeros(1, R.POSE)
        p["quat"], p["pos"] = q, rng.normal(size=3) * 5
        out = np.zeros(14)
        host.wc_host_marker(s.ctypes.data_as(C.c_void_p), p.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        Rw = _quat_to_mat(q)
        assert np.allclose(out[0:3], Rw @ s["center"][0] + p["pos"][0], atol=1e-12)
        Rm = _quat_to_mat(out[3:7])
        assert abs(np.linalg.det(Rm) - 1) < 1e-12
        cov_w = Rw @ cov @ Rw.T
        assert np.allclose(Rm @ np.diag((out[7:10] / 3) ** 2) @ Rm.T, cov_w, atol=1e-12 * np.abs(cov_w).max() + 1e-18)  # the same ellipsoid
        assert sorted((out[7:10] / 3) ** 2) == pytest.approx(sorted(w), rel=1e-9, abs=1e-18)
        assert np.allclose(out[10:13], ((Rw @ U[:, 0] + 1) / 2).astype(np.float32), atol=1e-7) and out[13] == 1.0
