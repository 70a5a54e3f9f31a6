"""A/B of the assembly on one box: python profiles/dev/ab_lin.py <tree> [imu_apart] - ms per linearisation (HIP events on the ctx stream,
50 linearisations) of the C4 window (20 x 50 000 surfels) and of an odometry-step-sized window (10 x 25 000), LM iterations / s of the
C4 solve, and a checksum of H / g / cost, with the library of <tree>"""
import hashlib, os, sys, time
tree = os.path.abspath(sys.argv[1])
sys.path.insert(0, os.path.join(tree, "wildcat-slam_amd", "python"))
import numpy as np
from wildcat_slam_amd import lib, synth

ctx = lib.Context(0)
for opt in sys.argv[2:]:
    k, v = opt.split("=")
    ctx.set_dev_option(k, int(v))
out = []
for scans, patches in (((20, 50000), (10, 25000)) if not os.environ.get("SIZES") else [tuple(int(a) for a in z.split("x")) for z in os.environ["SIZES"].split(",")]):
    w = synth.surfel_window(scans, patches, seed=synth.SEED + 7, fixed_patches=patches)
    n_s = len(w["surf"])
    d_surf, d_pose = ctx.to_device(w["surf"]), ctx.to_device(w["pose"])
    d_fs, d_fp = ctx.to_device(w["fix_surf"]), ctx.to_device(w["fix_pose"])
    d_pairs, d_pf = ctx.alloc(8 * n_s), ctx.alloc(8 * n_s)
    n_b, n_u = ctx.match_pair_device(d_surf, d_pose, n_s, d_fs, d_fp, len(w["fix_surf"]), d_pairs, n_s, d_pf, n_s)
    ctx.window_build(d_surf, d_pose, d_pairs, n_b, w["imu"], w["sample_times"], w["grav"], False, d_fs, d_fp, d_pf, n_u)
    ns = len(w["sample_times"])
    x0 = np.zeros(12 * ns)
    for _ in range(5):
        ctx.window_linearize_only(x0)
    best = 1e9
    for rep in range(3):
        ctx.timer_start()
        for _ in range(50):
            ctx.window_linearize_only(x0)
        best = min(best, ctx.timer_stop_ms() / 50)
    rng = np.random.default_rng(5)
    x1 = 1e-3 * rng.standard_normal(12 * ns)
    res = ctx.window_linearize(x1)
    h = hashlib.sha256()
    for a in res:
        h.update(np.ascontiguousarray(a).tobytes())
    norms = "|H| %.15e |g| %.15e c %.15e" % (np.linalg.norm(res[0]), np.linalg.norm(res[1]), res[2])
    if os.environ.get("NOSOLVE"):  # (knock-out builds: the numbers are wrong on purpose)
        out.append("%dx%d: lin %.4f ms" % (scans, patches, best))
        continue
    ctx.window_solve(x0)
    ctx.sync()
    t0 = time.perf_counter()
    x, summ, _ = ctx.window_solve(x0)
    ctx.sync()
    t_solve = time.perf_counter() - t0
    out.append("%dx%d: lin %.4f ms, solve %.3f ms / %d it = %.0f it/s, cost %.12e, %s sha %s" % (
        scans, patches, best, t_solve * 1e3, summ.iterations, summ.iterations / t_solve, summ.final_cost, norms, h.hexdigest()[:10]))
print(os.path.basename(tree) or tree, " ".join(sys.argv[2:]), "|", " | ".join(out))
