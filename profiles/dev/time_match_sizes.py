# the two searches of a step-like window (8 sweeps, a quarter as many fixed-window surfels) at several sizes: python time_match_sizes.py
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R + "/wildcat-slam_amd/python"]
import numpy as np
from wildcat_slam_amd import lib, synth
ctx = lib.Context(0)
for patches in (2000, 4000, 8000, 16000, 31248, 62496):
    w = synth.surfel_window(8, patches, seed=synth.SEED + 7, fixed_patches=2 * patches)
    n_s, n_f = len(w["surf"]), len(w["fix_surf"])
    d_s, d_p, d_fs, d_fp = ctx.to_device(w["surf"]), ctx.to_device(w["pose"]), ctx.to_device(w["fix_surf"]), ctx.to_device(w["fix_pose"])
    d_b, d_u = ctx.alloc(8 * n_s), ctx.alloc(8 * n_s)
    res = {}
    for which in ("same", "fixed", "pair"):
        ts = []
        for rep in range(5):
            ctx.sync(); t0 = time.perf_counter()
            if which == "same":
                ctx.match_device(d_s, d_p, n_s, d_s, d_p, n_s, True, d_b, n_s)
            elif which == "fixed":
                ctx.match_device(d_s, d_p, n_s, d_fs, d_fp, n_f, False, d_u, n_s)
            else:
                ctx.match_pair_device(d_s, d_p, n_s, d_fs, d_fp, n_f, d_b, n_s, d_u, n_s)
            ts.append(time.perf_counter() - t0)
        res[which] = round(min(ts[1:]) * 1e3, 3)
    print("%7d queries, %6d fixed: %s" % (n_s, n_f, res), flush=True)
