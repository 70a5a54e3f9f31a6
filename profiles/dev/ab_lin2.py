"""A/B of one linearisation at C4 and at the odometry step's size under development options (round 6): device time per linearisation
(wc_window_linearize_timed), pieces, max |H - H_first| / max |H|.  python profiles/dev/ab_lin2.py "lin_pair=0" "" """
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", os.environ.get("AB_LM_TREE", "."), "wildcat-slam_amd", "python"))
import numpy as np
from wildcat_slam_amd import lib, synth

settings = sys.argv[1:] or [""]
ctx = lib.Context(0)
for name, (scans, patches) in (("C4 20x50000", (20, 50000)), ("10x31248", (10, 31248)), ("8x3000", (8, 3000))):
    w = synth.surfel_window(scans, patches, seed=synth.SEED + 7, fixed_patches=patches)
    n_s = len(w["surf"])
    d_surf, d_pose = ctx.to_device(w["surf"]), ctx.to_device(w["pose"])
    d_fs, d_fp = ctx.to_device(w["fix_surf"]), ctx.to_device(w["fix_pose"])
    d_pairs, d_pf = ctx.alloc(8 * n_s), ctx.alloc(8 * n_s)
    n_b, n_u = ctx.match_pair_device(d_surf, d_pose, n_s, d_fs, d_fp, len(w["fix_surf"]), d_pairs, n_s, d_pf, n_s)
    ns = len(w["sample_times"])
    x0 = float(os.environ.get("AB_X0", "0")) * np.random.default_rng(3).normal(size=12 * ns)
    first = None
    for spec in settings:
        for kv in [s for s in spec.split(",") if s]:
            ctx.set_dev_option(kv.split("=")[0], int(kv.split("=")[1]))
        ctx.window_build(d_surf, d_pose, d_pairs, n_b, w["imu"], w["sample_times"], w["grav"], False, d_fs, d_fp, d_pf, n_u)
        H, g, cost = ctx.window_linearize(x0)
        tms = [ctx.window_linearize_timed(x0, 40) for _ in range(12)]
        ms = min(tms)
        if os.environ.get("AB_VERBOSE"): print(" ".join("%.3f" % t for t in tms))
        for kv in [s for s in spec.split(",") if s]:
            ctx.set_dev_option(kv.split("=")[0], {"lin_pair": 1}.get(kv.split("=")[0], 0))
        if first is None:
            first = (H, g, cost)
        print("%-12s [%-24s] %.4f ms per linearisation, pieces %d, |dH| %.1e |dg| %.1e |dcost| %.1e" % (
            name, spec, ms, ctx.window_counts()[3], np.abs(H - first[0]).max() / np.abs(first[0]).max(), np.abs(g - first[1]).max() / np.abs(first[1]).max(),
            abs(cost - first[2]) / abs(first[2])), flush=True)
