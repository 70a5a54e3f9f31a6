"""Randomised PARAMETER + cloud stress of the default (integer-moment) extraction against the CPU oracle, several repetitions per
configuration (a race shows as a repetition that differs) in both forms of the node stage.  Round 5 wrote it after a parameter test
found a displaced-root race of the layer-2 pass.  python profiles/stress_params.py [seconds] [seed0] [exact]  (exact: also two runs in the exact arithmetic, byte for byte)"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R + "/wildcat-slam_amd/python", R + "/oracle", R + "/tests"]
import numpy as np
import pyoracle, helpers
from wildcat_slam_amd import lib, synth
from test_fuzz_gpu import _cloud

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ctx = lib.Context(0)
t_end = time.time() + budget
n_cfg = n_bad = n_fast = n_runs = n_exact = 0
exact_too = len(sys.argv) > 3 and sys.argv[3] == "exact"
seed = seed0
while time.time() < t_end:
    seed += 1
    rng = np.random.default_rng(77_000 + seed)
    kind = seed % 4
    if kind == 0:
        pts = _cloud(5000 + seed)
    elif kind == 1:
        pts = synth.g2_lattice(int(rng.integers(20, 400)), m=int(rng.integers(21, 64)), seed=int(rng.integers(1, 1 << 30)))[0]
    elif kind == 2:
        pts = synth.g1_room(int(rng.integers(20_000, 250_000)), seed=int(rng.integers(1, 1 << 30)))
    else:  # a lattice whose patches straddle the voxels of an odd grid, shifted
        pts = synth.g2_lattice(int(rng.integers(50, 300)), m=int(rng.integers(25, 50)), seed=int(rng.integers(1, 1 << 30)))[0].copy()
        sh = rng.uniform(-0.4, 0.4, 3).astype(np.float32)
        pts["x"] += sh[0]; pts["y"] += sh[1]; pts["z"] += sh[2]
    prm = pyoracle.default_params()
    prm.voxel_size = float(np.float32(rng.choice([0.8, 0.8, rng.uniform(0.2, 0.98)])))
    prm.max_layer = int(rng.choice([2, 2, 2, 1, 0]))
    prm.min_points = int(rng.choice([20, 20, rng.integers(4, 40)]))
    prm.cluster_min_points = int(rng.choice([20, 20, rng.integers(4, 40)]))
    prm.cluster_gap = float(rng.choice([0.05, 0.05, 10 ** rng.uniform(-4, -1)]))
    prm.planer_threshold = float(np.float32(rng.choice([0.01, 0.01, 10 ** rng.uniform(-4, -1)])))
    prm.min_plane_likeness = float(rng.choice([0.1, 0.1, rng.uniform(0.0, 0.6)]))
    s_ref, id_ref, st = pyoracle.extract_surfels(pts, prm)
    want = set(helpers.id_tuples(id_ref))
    ctx.set_params(prm); ctx.params = prm
    n_cfg += 1
    if exact_too:  # the exact arithmetic (the path every fall-back ends on): the oracle's bytes, in the oracle's order
        ctx.set_exact_sums(True)
        for rep in range(2):
            s, i = ctx.extract_surfels(pts)
            n_exact += 1
            if not (i.tobytes() == id_ref.tobytes() and all(np.array_equal(s[f], s_ref[f], equal_nan=True) for f in ("t", "center", "cov", "normal", "sigma", "resolution"))):
                n_bad += 1
                print("EXACT MISMATCH seed", seed, "kind", kind, "rep", rep, "n", len(pts), "vs", prm.voxel_size, "layers", prm.max_layer, "min", prm.min_points, prm.cluster_min_points,
                      "gap", prm.cluster_gap, "surfels", len(s), len(s_ref), "ids equal", i.tobytes() == id_ref.tobytes())
        ctx.set_exact_sums(False)
    for form in (0, 1):
        ctx.set_dev_option("fx_split", form)
        for rep in range(3):
            try:
                s, i = ctx.extract_surfels(pts)
            except Exception as e:
                n_bad += 1
                print("EXCEPTION seed", seed, "kind", kind, "form", form, repr(e)[:200])
                continue
            n_runs += 1
            info = ctx.extract_path_info()
            n_fast += bool(info["fast"])
            got = set(helpers.id_tuples(i))
            ok = got == want and len(s) == len(s_ref)
            if ok and len(s_ref):
                try:
                    helpers.check_surfels(s, i, s_ref, id_ref, tol=1e-6, t_tol=1e-5)
                except AssertionError as e:
                    ok = False
                    import traceback
                    tb = traceback.extract_tb(e.__traceback__)[-1]
                    print("GEOMETRY seed", seed, "helpers.py:%d" % tb.lineno, tb.line[:120], repr(e)[:200])
            if not ok:
                n_bad += 1
                print("MISMATCH seed", seed, "kind", kind, "form", form, "rep", rep, "n", len(pts), "vs", prm.voxel_size, "layers", prm.max_layer, "min", prm.min_points, prm.cluster_min_points,
                      "gap", prm.cluster_gap, "thr", prm.planer_threshold, "like", prm.min_plane_likeness, "surfels", len(s), len(s_ref), "fast", info["fast"],
                      "missing", sorted(want - got)[:3], "extra", sorted(got - want)[:3])
ctx.set_dev_option("fx_split", -1)
print("configurations %d, runs %d (%d completed by the default path), exact-arithmetic runs %d, mismatches %d, last seed %d" % (n_cfg, n_runs, n_fast, n_exact, n_bad, seed))
