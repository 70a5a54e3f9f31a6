"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU, exports every symbol that
include/wildcat_hip.h declares, agrees with the oracle on the reference's hard-coded parameters, and refuses to
compute without a device (no silent CPU fallback)."""
import ctypes as C

import pytest

from wildcat_slam_amd import lib
from wildcat_slam_amd import records as R


def test_library_exports_every_declared_symbol():
    l = lib.load()
    syms = lib.declared_symbols()
    assert len(syms) >= 25
    missing = [s for s in syms if not hasattr(l, s)]
    assert not missing, missing
    assert b"gfx950" in l.wc_version()


def test_default_params_match_reference_values(oracle):
    a, b = lib.default_params(), oracle.default_params()
    for name, _ in R.Params._fields_:
        va, vb = getattr(a, name), getattr(b, name)
        if hasattr(va, "__len__"):
            assert list(va) == list(vb), name
        else:
            assert va == vb, name
    # SURVEY.md §2.1 spot checks (surfel_extraction.cc:327, knn_surfel_matcher.h:37-41, lio_config.h:32)
    assert a.voxel_size == C.c_float(0.8).value and a.max_layer == 2 and a.min_points == 20 and a.knn_k == 10
    assert a.planer_threshold == C.c_float(0.01).value and a.cluster_gap == 0.05 and a.imu_dt == 0.005


def test_record_layouts_match_header():
    assert R.SURFEL.itemsize == 144 and R.POSE.itemsize == 56 and R.IMU_STATE.itemsize == 112
    assert R.POINT.itemsize == 48 and R.POINT.fields["time"][1] == 24  # hilti_ros::Point, common.h:12-28
    assert C.sizeof(R.Points) == 32


def test_no_gpu_means_loud_failure():
    l = lib.load()
    if l.wc_device_count() > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(lib.WildcatError) as e:
        lib.Context(0)
    assert e.value.code == 12  # WC_ERR_NOGPU: there is no CPU fallback behind the C-ABI
