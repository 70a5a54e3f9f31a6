"""default-arithmetic extraction at a given voxel size against the oracle, with the library of <tree>: python profiles/dev/vs_check.py <tree> <vs>"""
import os, sys
tree = os.path.abspath(sys.argv[1]); vs = float(sys.argv[2])
sys.path[:0] = [os.path.join(tree, "wildcat-slam_amd", "python"), "/root/repo/oracle", "/root/repo/tests"]
import numpy as np
import pyoracle, helpers
from wildcat_slam_amd import lib, synth
ctx = lib.Context(0)
prm = pyoracle.default_params(); prm.voxel_size = vs
ctx.set_params(prm); ctx.params = prm
for name, pts in (("g2", synth.g2_lattice(120, m=40)[0]), ("room", synth.g1_room(90_000, seed=13))):
    s_ref, id_ref, st = pyoracle.extract_surfels(pts, prm)
    for exact in (True, False):
        ctx.set_exact_sums(exact)
        s, i = ctx.extract_surfels(pts)
        info = ctx.extract_path_info()
        a, b = set(helpers.id_tuples(i)), set(helpers.id_tuples(id_ref))
        print(os.path.basename(tree) or "repo", name, "exact" if exact else "default", len(s), len(s_ref), "fast" if info["fast"] else "exact-path", "missing", sorted(b - a)[:3], "extra", sorted(a - b)[:3])
