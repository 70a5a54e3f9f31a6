"""numpy / ctypes mirrors of the C-ABI records in include/wc_types.h.

Every dtype here is byte-compatible with the C struct of the same name, so a numpy array can be handed to
the C-ABI (or copied to HBM) without conversion.
"""
import ctypes as C

import numpy as np

# reference hilti_ros::Point (src/common/common.h:12-28): 48-byte AoS record
POINT = np.dtype(
    {
        "names": ["x", "y", "z", "intensity", "time", "ring"],
        "formats": ["f4", "f4", "f4", "f4", "f8", "u2"],
        "offsets": [0, 4, 8, 16, 24, 32],
        "itemsize": 48,
    }
)
SURFEL = np.dtype(
    [("t", "f8"), ("center", "f8", 3), ("cov", "f8", 9), ("normal", "f8", 3), ("resolution", "f8"), ("sigma", "f8")]
)
SURFEL_ID = np.dtype([("kx", "i4"), ("ky", "i4"), ("kz", "i4"), ("node", "u4")])
POSE = np.dtype([("pos", "f8", 3), ("quat", "f8", 4)])
IMU_STATE = np.dtype([("t", "f8"), ("pos", "f8", 3), ("quat", "f8", 4), ("acc", "f8", 3), ("gyr", "f8", 3)])
PAIR = np.dtype([("first", "i4"), ("second", "i4")])

ROUTE_POINT = np.dtype([("x", "f4"), ("y", "f4"), ("z", "f4"), ("src", "u4"), ("t", "f8")])  # wc_route_point
assert ROUTE_POINT.itemsize == 24

assert SURFEL.itemsize == 144 and POSE.itemsize == 56 and IMU_STATE.itemsize == 112 and PAIR.itemsize == 8
assert SURFEL_ID.itemsize == 16 and POINT.itemsize == 48


class Points(C.Structure):
    """wc_points descriptor."""

    _fields_ = [
        ("xyz", C.c_void_p),
        ("time", C.c_void_p),
        ("xyz_stride", C.c_uint32),
        ("time_stride", C.c_uint32),
        ("n", C.c_uint64),
    ]


class Params(C.Structure):
    """wc_params."""

    _fields_ = [
        ("voxel_size", C.c_float),
        ("max_layer", C.c_int32),
        ("min_points", C.c_int32),
        ("planer_threshold", C.c_float),
        ("min_plane_likeness", C.c_double),
        ("view_point", C.c_double * 3),
        ("cluster_gap", C.c_double),
        ("cluster_min_points", C.c_int32),
        ("center_scale", C.c_double),
        ("angular_scale", C.c_double),
        ("surfel_dist_max", C.c_double),
        ("knn_k", C.c_int32),
        ("time_diff_min", C.c_double),
        ("surfel_sigma0", C.c_double),
        ("cauchy_a", C.c_double),
        ("w_gyr", C.c_double),
        ("w_acc", C.c_double),
        ("w_bg", C.c_double),
        ("w_ba", C.c_double),
        ("imu_dt", C.c_double),
        ("max_iterations", C.c_int32),
        ("reference_quirks", C.c_int32),
        ("exact_sums", C.c_int32),
    ]


COMM_ALLREDUCE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64)
COMM_ALLTOALLV = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(C.c_uint64))
COMM_ALLGATHERV = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint64))


class Comm(C.Structure):
    """wc_comm."""

    _fields_ = [
        ("user", C.c_void_p),
        ("rank", C.c_int32),
        ("world", C.c_int32),
        ("allreduce_f64", COMM_ALLREDUCE),
        ("alltoallv", COMM_ALLTOALLV),
        ("allgatherv", COMM_ALLGATHERV),
        ("stream_ordered", C.c_int32),
    ]


class SolveSummary(C.Structure):
    """wc_solve_summary."""

    _fields_ = [
        ("initial_cost", C.c_double),
        ("final_cost", C.c_double),
        ("iterations", C.c_int32),
        ("successful_steps", C.c_int32),
        ("unsuccessful_steps", C.c_int32),
        ("termination", C.c_int32),
        ("n_linearizations", C.c_int32),
        ("n_cost_evaluations", C.c_int32),
        ("first_step", C.c_double * 16),
    ]


def points_from_aos(arr, base_ptr=None):
    """Descriptor for a POINT-dtype array (host array unless base_ptr, a device address, is given)."""
    assert arr.dtype == POINT
    base = arr.ctypes.data if base_ptr is None else base_ptr
    return Points(base, base + 24, 48, 48, len(arr))


class SweepJob(C.Structure):
    """wc_sweep_job (include/wc_types.h): one sweep of wc_extract_surfels_batch_*"""

    _fields_ = [("pts", Points), ("t_lo", C.c_double), ("t_hi", C.c_double), ("d_out", C.c_void_p), ("d_ids", C.c_void_p), ("cap", C.c_uint64)]


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)
