# does the leaf-order sort of a two-set search's queries still pay with the early bound?  The step-like window's fixed-window search with the
# queries in their own order and in a RANDOM order, sorted (knn_sort=1) and not (0): python profiles/dev/ab_sort_random.py
import os, sys, time
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R_ + "/wildcat-slam_amd/python"]
import numpy as np
from wildcat_slam_amd import lib, synth
ctx = lib.Context(0)
for name, (scans, patches, fixed) in (("step-like 250k/62k", (8, 31248, 62496)), ("C4 1M/50k", (20, 50000, 50000)), ("window 8x8000/16000", (8, 8000, 16000))):
    w = synth.surfel_window(scans, patches, seed=synth.SEED + 7, fixed_patches=fixed)
    n_s, n_f = len(w["surf"]), len(w["fix_surf"])
    perm = np.random.default_rng(1).permutation(n_s)
    d_fs, d_fp = ctx.to_device(w["fix_surf"]), ctx.to_device(w["fix_pose"])
    d_u = ctx.alloc(8 * n_s)
    for order, (S, P) in (("own order", (w["surf"], w["pose"])), ("random order", (w["surf"][perm], w["pose"][perm]))):
        d_s, d_p = ctx.to_device(S), ctx.to_device(P)
        res = []
        for srt in (1, 0):
            ctx.set_dev_option("knn_sort", srt)
            ts = []
            for rep in range(7):
                ctx.sync(); t0 = time.perf_counter()
                n = ctx.match_device(d_s, d_p, n_s, d_fs, d_fp, n_f, False, d_u, n_s)
                ts.append(time.perf_counter() - t0)
            ts = sorted(ts[1:])
            res.append("sort=%d %.3f ms" % (srt, ts[len(ts) // 2] * 1e3))
        ctx.set_dev_option("knn_sort", -1)
        print("%-22s %-12s %s  (%d pairs)" % (name, order, "  ".join(res), n), flush=True)
