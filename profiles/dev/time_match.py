# matcher timing on the bench windows: python profiles/dev/time_match.py [same|fixed|pair]   (with an argument: the step-like window only)
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R + "/wildcat-slam_amd/python"]
import numpy as np
from wildcat_slam_amd import lib, synth
ctx = lib.Context(0)
only = sys.argv[1] if len(sys.argv) > 1 else None
for name, (scans, patches, fixed) in ((("C4 1M/50k", (20, 50000, 50000)),) if not only else ()) + (("step-like 250k/62k", (8, 31248, 62496)),):
    w = synth.surfel_window(scans, patches, seed=synth.SEED + 7, fixed_patches=fixed)
    n_s, n_f = len(w["surf"]), len(w["fix_surf"])
    d_s, d_p, d_fs, d_fp = ctx.to_device(w["surf"]), ctx.to_device(w["pose"]), ctx.to_device(w["fix_surf"]), ctx.to_device(w["fix_pose"])
    d_b, d_u = ctx.alloc(8 * n_s), ctx.alloc(8 * n_s)
    res = {}
    for which in (("same", "fixed", "pair") if not only else (only,)):
        ts = []
        for rep in range(6):
            ctx.sync(); t0 = time.perf_counter()
            if which == "same":
                n = ctx.match_device(d_s, d_p, n_s, d_s, d_p, n_s, True, d_b, n_s)
            elif which == "fixed":
                n = ctx.match_device(d_s, d_p, n_s, d_fs, d_fp, n_f, False, d_u, n_s)
            else:
                n = ctx.match_pair_device(d_s, d_p, n_s, d_fs, d_fp, n_f, d_b, n_s, d_u, n_s)
            ts.append(time.perf_counter() - t0)
        res[which] = (round(min(ts[1:]) * 1e3, 3), n)
    print(name, res)
