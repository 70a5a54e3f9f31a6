"""Shared comparison helpers for the parity tests."""
import numpy as np


def id_tuples(ids):
    return [tuple(r) for r in np.stack([ids["kx"], ids["ky"], ids["kz"], ids["node"].astype(np.int64)], -1).tolist()]


def match_by_id(ids_a, ids_b):
    """permutation p such that b[p[i]] has the id of a[i]; asserts the id multisets are identical"""
    ta, tb = id_tuples(ids_a), id_tuples(ids_b)
    assert len(ta) == len(tb), (len(ta), len(tb))
    pos = {t: i for i, t in enumerate(tb)}
    assert len(pos) == len(tb), "duplicate surfel ids"
    assert set(ta) == set(tb), "surfel id sets differ"
    return np.array([pos[t] for t in ta], np.int64)


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    denom = np.maximum(np.abs(b).max(), 1e-300)
    return float(np.abs(a - b).max() / denom)


def check_surfels(s_gpu, id_gpu, s_ref, id_ref, tol=1e-6, t_tol=None, strict_sigma=False):
    """counts and ids bit-exact; geometry within `tol` relative (north_star: 1e-6).
    sigma = sqrt(lambda_0) of a cluster whose points lie EXACTLY in a plane (every stretch of one scan line on a surface: line direction x
    ray direction) is the reference's own rounding noise - NaN or ~1e-7 by the luck of its un-centred sums.  strict_sigma (the exact
    arithmetic, which reproduces those sums): the NaN pattern and every value must agree.  Otherwise (the default arithmetic, exact
    integer sums): where the reference's sigma is NaN or below the noise floor sqrt(16 sqrt(n) |c|^2 eps) ~ 3e-5 x |c|, the other side
    must be NaN or below twice that floor, and nothing more is compared."""
    p = match_by_id(id_ref, id_gpu)
    g = s_gpu[p]
    n = len(s_ref)
    if n == 0:
        return dict(n=0)
    # normals: unit vectors, compare absolutely
    dn = np.abs(g["normal"] - s_ref["normal"]).max()
    scale_c = np.maximum(np.abs(s_ref["center"]).max(), 1.0)
    dc = np.abs(g["center"] - s_ref["center"]).max() / scale_c
    # degenerate clusters (all points identical after rounding): zero covariance, sigma = sqrt(lambda_min) with lambda_min a
    # rounding-sized negative number, i.e. NaN in the reference too - the NaN pattern must agree, the rest is compared
    cov_scale = np.maximum(np.abs(s_ref["cov"]).max(axis=1, keepdims=True), 1e-300)
    dcov = (np.abs(g["cov"] - s_ref["cov"]) / cov_scale).max()
    nan_ref = np.isnan(s_ref["sigma"])
    if strict_sigma:
        assert np.array_equal(np.isnan(g["sigma"]), nan_ref)
        ok = ~nan_ref
    else:
        floor = 3e-5 * np.sqrt((s_ref["center"] ** 2).sum(axis=1) + 1.0)
        noise = nan_ref | (s_ref["sigma"] <= floor)
        assert np.all(np.isnan(g["sigma"][noise]) | (g["sigma"][noise] <= 2 * floor[noise]))
        assert not np.isnan(g["sigma"][~noise]).any()
        ok = ~noise
    dsig = (np.abs(g["sigma"][ok] - s_ref["sigma"][ok]).max() / max(np.abs(s_ref["sigma"][ok]).max(), 1e-300)) if ok.any() else 0.0
    assert np.array_equal(g["resolution"], s_ref["resolution"])
    assert dn <= tol and dc <= tol and dcov <= tol and dsig <= tol, (dn, dc, dcov, dsig)
    dt = np.abs(g["t"] - s_ref["t"]).max()
    if t_tol is not None:
        assert dt <= t_tol, dt
    # the GPU output must itself be sorted by timestamp (surfel_extraction.cc:334)
    assert np.all(np.diff(s_gpu["t"]) >= 0)
    bit_exact = all(np.array_equal(g[f], s_ref[f], equal_nan=True) for f in ("t", "center", "cov", "normal", "sigma"))
    return dict(n=n, dn=dn, dc=dc, dcov=dcov, dsig=dsig, dt=dt, bit_exact=bit_exact)


def check_fast_and_exact(gpu, oracle, pts, params=None, expect_fast=None, **kw):
    """wc_extract_surfels in both arithmetic modes against the oracle:
    exact  - every sum in the reference's order: surfels AND ids byte for byte the oracle's, in the oracle's order;
    fast   - (default) integer moments: counts and ids identical as sets, geometry within 1e-6 (north_star), output sorted by
             its own (correctly rounded) time stamps.  expect_fast: True = the sweep must have been completed by the fast path
             itself (no silent fall-back to the exact path), None = either."""
    s_ref, id_ref, st = oracle.extract_surfels(pts, params) if params is not None else oracle.extract_surfels(pts)
    out = {}
    for exact in (True, False):
        gpu.set_exact_sums(exact)
        try:
            s_gpu, id_gpu = gpu.extract_surfels(pts, **kw)
            path = gpu.extract_path_info()
        finally:
            gpu.set_exact_sums(False)
        assert len(s_gpu) == len(s_ref) == st.surfels, (exact, len(s_gpu), len(s_ref))
        if exact:
            assert not path["fast"]
            assert id_gpu.tobytes() == id_ref.tobytes()  # the ORDER is the oracle's too: stamp, then voxel index and node id (Q7)
            info = check_surfels(s_gpu, id_gpu, s_ref, id_ref, tol=1e-6, t_tol=1e-5, strict_sigma=True) if len(s_ref) else dict(n=0)
        else:
            if expect_fast:
                assert path["fast"], path
            info = check_surfels(s_gpu, id_gpu, s_ref, id_ref, tol=1e-6, t_tol=1e-5) if len(s_ref) else dict(n=0)
            info["fast_path"] = path["fast"]
        out["exact" if exact else "fast"] = info
    return out, st
