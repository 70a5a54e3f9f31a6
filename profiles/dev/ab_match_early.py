# A/B of the two-set search's early bound (round 6, development option knn_early): time and pair lists byte for byte, on the bench's windows,
# the odometry step's scan window and a room window.  python profiles/dev/ab_match_early.py [quick]
import os, sys, time, zlib
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R + "/wildcat-slam_amd/python"]
import numpy as np
from wildcat_slam_amd import lib, synth

ctx = lib.Context(0)
quick = len(sys.argv) > 1
cases = [("step-like 250k/62k", (8, 31248, 62496)), ("window 8x3000 / 3000", (8, 3000, 3000)), ("window 4x500 / 100", (4, 500, 100))]
if not quick:
    cases.append(("C4 1M/50k", (20, 50000, 50000)))
for name, (scans, patches, fixed) in cases:
    w = synth.surfel_window(scans, patches, seed=synth.SEED + 7, fixed_patches=fixed)
    n_s, n_f = len(w["surf"]), len(w["fix_surf"])
    d_s, d_p, d_fs, d_fp = ctx.to_device(w["surf"]), ctx.to_device(w["pose"]), ctx.to_device(w["fix_surf"]), ctx.to_device(w["fix_pose"])
    d_b, d_u = ctx.alloc(8 * n_s), ctx.alloc(8 * n_s)
    out = {}
    for early in (0, 1):
        ctx.set_dev_option("knn_early", early)
        for which in ("fixed", "pair"):
            ts = []
            for rep in range(6):
                ctx.sync(); t0 = time.perf_counter()
                if which == "fixed":
                    n = ctx.match_device(d_s, d_p, n_s, d_fs, d_fp, n_f, False, d_u, n_s)
                else:
                    n = ctx.match_pair_device(d_s, d_p, n_s, d_fs, d_fp, n_f, d_b, n_s, d_u, n_s)
                ts.append(time.perf_counter() - t0)
            nu = n if which == "fixed" else n[1]
            pairs = d_u.download(np.uint8, 8 * int(nu)) if nu else np.zeros(0, np.uint8)
            crc = zlib.crc32(pairs.tobytes())
            if which == "pair" and n[0]:
                crc = zlib.crc32(d_b.download(np.uint8, 8 * int(n[0])).tobytes(), crc)
            out[(early, which)] = (min(ts[1:]) * 1e3, n, crc)
        st = ctx.match_stats() if hasattr(ctx, "match_stats") else None
        out[(early, "stats")] = st
    ctx.set_dev_option("knn_early", 1)
    for which in ("fixed", "pair"):
        a, b = out[(0, which)], out[(1, which)]
        print("%-22s %-5s plain %.3f ms  early %.3f ms  pairs %s / %s  crc %08x / %08x  %s" % (name, which, a[0], b[0], a[1], b[1], a[2], b[2], "SAME" if a[1:] == b[1:] else "DIFFERENT"), flush=True)
    print("   walk stats (plain / early):", out[(0, "stats")], out[(1, "stats")])
