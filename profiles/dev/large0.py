import os, sys
R_ = "/root/repo"
sys.path[:0] = [R_ + "/" + os.environ.get("TREE", ".") + "/wildcat-slam_amd/python", R_ + "/oracle", R_ + "/tests"]
import numpy as np
import pyoracle, helpers
from wildcat_slam_amd import lib, synth
ctx = lib.Context(0)
rng = np.random.default_rng(8)
n = int(rng.integers(2_200_000, 5_000_000)); seed = int(rng.integers(1, 1 << 30))
pts = synth.g1_room(n, seed=seed)
prm = pyoracle.default_params(); prm.voxel_size = float(np.float32(0.8))
s_ref, id_ref, _ = pyoracle.extract_surfels(pts, prm)
ctx.set_params(prm); ctx.params = prm
for exact in (False, True):
    ctx.set_exact_sums(exact)
    s, ids = ctx.extract_surfels(pts)
    try:
        print(exact, helpers.check_surfels(s, ids, s_ref, id_ref, tol=1e-6, t_tol=1e-5))
    except AssertionError as e:
        print(exact, "ASSERT", str(e)[:600])
