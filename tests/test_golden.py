"""Committed golden vectors (tests/golden/*.npz), produced by the INDEPENDENT numpy/scipy restatement
oracle/np_check.py (LAPACK eigh, cKDTree, finite-difference Jacobians — `python oracle/np_check.py --write`).
The CPU oracle is checked against them here; the HIP path is checked against the same files in the gpu-marked tests."""
import os

import numpy as np
import pytest

import helpers
from wildcat_slam_amd import records as R
from wildcat_slam_amd import synth

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load_extract():
    z = np.load(os.path.join(G, "extract_small.npz"))
    pts = np.ascontiguousarray(z["points"]).view(R.POINT).reshape(-1)
    return pts, z["surfels"].view(R.SURFEL), z["ids"].view(R.SURFEL_ID)


def _window_case():
    z = np.load(os.path.join(G, "window_small.npz"))
    w = synth.surfel_window(3, 120, seed=31, fixed_patches=60)
    return w, z


def _check_extract(s, ids, s_ref, id_ref):
    # ids / counts exact; geometry to 1e-6 relative (eigh vs Jacobi, pairwise vs sequential sums)
    info = helpers.check_surfels(s, ids, s_ref, id_ref, tol=1e-6, t_tol=1e-5)
    assert info["n"] == len(s_ref) > 1000
    assert len(np.unique(s_ref["resolution"])) >= 2  # multi-resolution output (Q4) is covered


def test_oracle_extraction_vs_numpy_golden(oracle):
    pts, s_ref, id_ref = _load_extract()
    s, ids, st = oracle.extract_surfels(pts)
    _check_extract(s, ids, s_ref, id_ref)
    assert st.clusters_rejected > 0 or st.nodes_tested[1] > st.nodes_plane[1]


def test_oracle_match_vs_numpy_golden(oracle):
    w, z = _window_case()
    assert np.array_equal(oracle.match(w["surf"], w["pose"], w["surf"], w["pose"], True), z["pairs"].view(R.PAIR))
    assert np.array_equal(oracle.match(w["surf"], w["pose"], w["fix_surf"], w["fix_pose"], False), z["pairs_fix"].view(R.PAIR))


@pytest.mark.parametrize("quirks", [0, 1])
def test_oracle_factors_vs_finite_differences(oracle, quirks):
    """analytic residuals / Jacobians (cost_functor.h) against finite differences of independently written residual
    functions; quirks = 1 checks the Jacobian-overwrite rule (Q1) on surfel factors (IMU factors are in the quirks = 0
    case: Q3 makes the reference's analytic IMU Jacobian differ from the true derivative by construction)."""
    w, z = _window_case()
    params = oracle.default_params()
    params.reference_quirks = quirks
    W = oracle.Window(w["sample_times"], w["grav"], True, params)
    W.add_binary(w["surf"], w["pose"], z["pairs"].view(R.PAIR))
    W.add_unary(w["fix_surf"], w["fix_pose"], w["surf"], w["pose"], z["pairs_fix"].view(R.PAIR))
    if quirks == 0:
        W.add_imu(w["imu"])
    x = z["x"]
    cost, res = W.evaluate(x, want_residuals=True)
    H, g, _ = W.linearize(x)
    assert abs(cost - z[f"cost_q{quirks}"]) <= 1e-9 * cost
    nl = len(z["pairs"]) + len(z["pairs_fix"])  # surfel residuals: the sign of the eigenvector normal is arbitrary (Q9)
    ref = z[f"res_q{quirks}"]
    assert np.abs(np.abs(res[:nl]) - np.abs(ref[:nl])).max() <= 1e-7 * np.abs(res).max()
    if len(res) > nl:
        assert np.abs(res[nl:] - ref[nl:]).max() <= 1e-7 * np.abs(res).max()
    assert np.abs(H - z[f"H_q{quirks}"]).max() <= 2e-6 * np.abs(H).max()
    assert np.abs(g - z[f"g_q{quirks}"]).max() <= 2e-6 * np.abs(g).max()


@pytest.mark.gpu
def test_gpu_extraction_vs_numpy_golden(gpu):
    pts, s_ref, id_ref = _load_extract()
    s, ids = gpu.extract_surfels(pts)
    _check_extract(s, ids, s_ref, id_ref)


@pytest.mark.gpu
def test_gpu_match_and_window_vs_numpy_golden(gpu, oracle):
    w, z = _window_case()
    pairs, pf = z["pairs"].view(R.PAIR), z["pairs_fix"].view(R.PAIR)
    assert np.array_equal(gpu.match(w["surf"], w["pose"], w["surf"], w["pose"], True), pairs)
    assert np.array_equal(gpu.match(w["surf"], w["pose"], w["fix_surf"], w["fix_pose"], False), pf)
    for quirks in (0, 1):
        params = oracle.default_params()
        params.reference_quirks = quirks
        gpu.set_params(params)
        d_s, d_p = gpu.to_device(w["surf"]), gpu.to_device(w["pose"])
        d_fs, d_fp = gpu.to_device(w["fix_surf"]), gpu.to_device(w["fix_pose"])
        d_pairs, d_pf = gpu.to_device(pairs), gpu.to_device(pf)
        gpu.window_build(d_s, d_p, d_pairs, len(pairs), w["imu"] if quirks == 0 else None, w["sample_times"], w["grav"], True, d_fs, d_fp,
                         d_pf, len(pf))
        H, g, cost = gpu.window_linearize(z["x"])
        assert abs(cost - z[f"cost_q{quirks}"]) <= 1e-9 * cost
        assert np.abs(H - z[f"H_q{quirks}"]).max() <= 2e-6 * np.abs(H).max()
        assert np.abs(g - z[f"g_q{quirks}"]).max() <= 2e-6 * np.abs(g).max()
    gpu.set_params(oracle.default_params())
