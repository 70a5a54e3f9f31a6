"""a few LARGE clouds (above the 2 M-point switch to the two-kernel node stage and the dynamic hand-out) against the oracle, default
arithmetic, two repetitions each: rooms in firing order, lattices shifted against the grid (layer 2 in use), odd voxel sizes.
python profiles/stress_large.py"""
import os, sys, time
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R_ + "/wildcat-slam_amd/python", R_ + "/oracle", R_ + "/tests"]
import numpy as np
import pyoracle, helpers
from wildcat_slam_amd import lib, synth
ctx = lib.Context(0)
bad = 0
rng = np.random.default_rng(8)
cases = []
for i in range(8):
    kind = i % 4
    vs = float(np.float32([0.8, 0.95, 0.6, 0.8][kind]))
    if kind in (0, 2):
        pts = synth.g1_room(int(rng.integers(2_200_000, 5_000_000)), seed=int(rng.integers(1, 1 << 30)))
    else:
        pts = synth.g2_lattice(int(rng.integers(9_000, 16_000)), m=int(rng.integers(24, 40)), seed=int(rng.integers(1, 1 << 30)))[0].copy()
        pts["x"] += np.float32(0.31); pts["y"] -= np.float32(0.17)
    prm = pyoracle.default_params(); prm.voxel_size = vs
    t0 = time.time()
    s_ref, id_ref, _ = pyoracle.extract_surfels(pts, prm)
    t_or = time.time() - t0
    ctx.set_params(prm); ctx.params = prm
    want = set(helpers.id_tuples(id_ref))
    for rep in range(2):
        s, ids = ctx.extract_surfels(pts)
        info = ctx.extract_path_info()
        got = set(helpers.id_tuples(ids))
        ok = got == want
        if ok:
            try:
                helpers.check_surfels(s, ids, s_ref, id_ref, tol=1e-6, t_tol=1e-5)
            except AssertionError as e:
                ok = False
        l2 = sum(1 for t in want if (t[3] & 3) == 2)
        print("case %d kind %d vs %.2f points %d surfels %d (layer 2: %d) rep %d fast %s: %s (oracle %.1f s)" % (i, kind, vs, len(pts), len(s_ref), l2, rep, info["fast"], "ok" if ok else "MISMATCH missing %s extra %s" % (sorted(want - got)[:3], sorted(got - want)[:3]), t_or))
        bad += not ok
print("mismatches", bad)
