import os, sys
sys.path[:0] = ["/root/repo/wildcat-slam_amd/python", "/root/repo/oracle", "/root/repo/tests"]
import numpy as np
import pyoracle, helpers
from wildcat_slam_amd import lib, synth
ctx = lib.Context(0)
prm = pyoracle.default_params(); prm.voxel_size = 0.95
ctx.set_params(prm); ctx.params = prm
pts = synth.g2_lattice(120, m=40)[0]
s_ref, id_ref, st = pyoracle.extract_surfels(pts, prm)
l2_ref = [t for t in helpers.id_tuples(id_ref) if (t[3] & 3) == 2]
print("oracle layer-2 surfels", len(l2_ref))
for split in (0, 1):
    ctx.set_dev_option("fx_split", split)
    for rep in range(3):
        s, i = ctx.extract_surfels(pts)
        got = set(helpers.id_tuples(i))
        l2 = [t for t in got if (t[3] & 3) == 2]
        print("split", split, "rep", rep, len(s), "layer-2 found", len(l2), "missing", sorted(set(helpers.id_tuples(id_ref)) - got), ctx.extract_path_info())
ctx.set_dev_option("fx_split", -1)
# sub-clouds: tile 20 alone, and the cloud from tile 20 on
for lo, hi in ((20 * 1024, 21 * 1024), (20 * 1024, len(pts)), (0, 21 * 1024), (21280 - 200, 21360 + 200)):
    sub = pts[lo:hi].copy()
    r, ir, _ = pyoracle.extract_surfels(sub, prm)
    s, i = ctx.extract_surfels(sub)
    print("points", lo, hi, "oracle", len(r), "gpu", len(s), "missing", sorted(set(helpers.id_tuples(ir)) - set(helpers.id_tuples(i))))
