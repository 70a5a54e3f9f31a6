# timing of the step-like pair of searches under development options: python profiles/dev/ab_match_opt.py "knn_sort=0" "knn_sort=1" ...
import os, sys, time, zlib
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R + "/wildcat-slam_amd/python"]
import numpy as np
from wildcat_slam_amd import lib, synth
ctx = lib.Context(0)
DEF = {"knn_sort": -1, "knn_early": 1, "knn_group": -1, "knn_park": -1}
for name, (scans, patches, fixed) in (("step-like 250k/62k", (8, 31248, 62496)), ("window 8x3000 / 3000", (8, 3000, 3000)), ("C4 1M/50k", (20, 50000, 50000))):
    w = synth.surfel_window(scans, patches, seed=synth.SEED + 7, fixed_patches=fixed)
    n_s, n_f = len(w["surf"]), len(w["fix_surf"])
    d_s, d_p, d_fs, d_fp = ctx.to_device(w["surf"]), ctx.to_device(w["pose"]), ctx.to_device(w["fix_surf"]), ctx.to_device(w["fix_pose"])
    d_b, d_u = ctx.alloc(8 * n_s), ctx.alloc(8 * n_s)
    for spec in (sys.argv[1:] or [""]):
        kv = [s.split("=") for s in spec.split(",") if s]
        for k, v in kv: ctx.set_dev_option(k, int(v))
        res = []
        for which in ("same", "fixed", "pair"):
            ts = []
            for rep in range(7):
                ctx.sync(); t0 = time.perf_counter()
                if which == "same": n = ctx.match_device(d_s, d_p, n_s, d_s, d_p, n_s, True, d_b, n_s)
                elif which == "fixed": n = ctx.match_device(d_s, d_p, n_s, d_fs, d_fp, n_f, False, d_u, n_s)
                else: n = ctx.match_pair_device(d_s, d_p, n_s, d_fs, d_fp, n_f, d_b, n_s, d_u, n_s)
                ts.append(time.perf_counter() - t0)
            ts = sorted(ts[1:])
            st = ctx.match_stats()
            res.append("%s %.3f" % (which, ts[len(ts) // 2] * 1e3) + (" (nodes %.1f leaves %.1f exact %.1f)" % (st["nodes_per_query"], st["leaves_per_query"], st["exact_per_query"]) if which != "pair" else ""))
        crc = zlib.crc32(d_u.download(np.uint8, 8 * int(n[1])).tobytes(), zlib.crc32(d_b.download(np.uint8, 8 * int(n[0])).tobytes()))
        for k, v in kv: ctx.set_dev_option(k, DEF.get(k, 0))
        print("%-22s [%-24s] %s ms  crc %08x" % (name, spec, "  ".join(res), crc), flush=True)
