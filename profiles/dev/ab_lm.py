"""A/B of the LM solve under development options (round 6): the C4 window (bench.py's), the odometry step's window and a small one; per
setting: iterations, termination, final cost, ms per LM iteration (median of the solves), crc32 of x, max |x - x_first_setting|.
python profiles/dev/ab_lm.py "lm_back_chunks=1" "lm_back_chunks=0" [...]   (each argument: comma-separated name=value pairs; "" = defaults)
env AB_LM_CASES=c4,small,step: which windows"""
import os, sys, time, zlib
ROOT_ = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT_, os.environ.get("AB_LM_TREE", "."), "wildcat-slam_amd", "python"))  # (AB_LM_TREE=ab_var/<name>: a compile-time variant, profiles/dev/mk_var.sh)
import numpy as np
from wildcat_slam_amd import lib, synth
from wildcat_slam_amd.step import StepWindow

settings = sys.argv[1:] or [""]
ctx = lib.Context(0)
ctx.warmup() if hasattr(ctx, "warmup") else None


def apply(spec, on):
    for kv in [s for s in spec.split(",") if s]:
        k, v = kv.split("=")
        ctx.set_dev_option(k, int(v) if on else {"lin_pair": 1, "lm_side_stream": 1, "pcr_ahead": 1, "fx_split": -1, "knn_group": -1}.get(k, 0))


def c4_like(scans, patches, seed):
    w = synth.surfel_window(scans, patches, seed=seed, fixed_patches=patches)
    n_s = len(w["surf"])
    d_surf, d_pose = ctx.to_device(w["surf"]), ctx.to_device(w["pose"])
    d_fs, d_fp = ctx.to_device(w["fix_surf"]), ctx.to_device(w["fix_pose"])
    d_pairs, d_pf = ctx.alloc(8 * n_s), ctx.alloc(8 * n_s)
    n_b, n_u = ctx.match_pair_device(d_surf, d_pose, n_s, d_fs, d_fp, len(w["fix_surf"]), d_pairs, n_s, d_pf, n_s)
    ctx.window_build(d_surf, d_pose, d_pairs, n_b, w["imu"], w["sample_times"], w["grav"], False, d_fs, d_fp, d_pf, n_u)
    ns = len(w["sample_times"])
    return np.zeros(12 * ns), (d_surf, d_pose, d_fs, d_fp, d_pairs, d_pf)


def solve_stats(x0, reps=5):
    ctx.window_solve(x0)
    ctx.sync()
    ts, x, summ = [], None, None
    for _ in range(reps):
        t0 = time.perf_counter()
        x, summ, _ = ctx.window_solve(x0)
        ctx.sync()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return x, summ, ts[len(ts) // 2]


cases = []
which = os.environ.get("AB_LM_CASES", "c4,small,step").split(",")
if "c4" in which and not os.environ.get("AB_LM_SKIP_C4"):
    cases.append(("C4 20x50000", lambda: c4_like(20, 50000, synth.SEED + 7)))
if "small" in which:
    cases.append(("window 8x3000", lambda: c4_like(8, 3000, synth.SEED + 3)))
for name, mk in cases:
    x0, keep = mk()
    first = None
    for spec in settings:
        apply(spec, True)
        x, summ, t = solve_stats(x0)
        apply(spec, False)
        if first is None:
            first = x
        it = max(1, summ.iterations)
        print("%-16s [%-28s] iters %3d ok/bad %d/%d term %d cost %.12e  %.4f ms/iter (%.1f it/s)  crc %08x  |dx| %.2e" % (
            name, spec, summ.iterations, summ.successful_steps, summ.unsuccessful_steps, summ.termination, summ.final_cost, t * 1e3 / it, it / t,
            zlib.crc32(x.tobytes()), float(np.abs(x - first).max())), flush=True)
    del keep

if "step" not in which:
    sys.exit(0)
w = synth.g2_scan_sequence(10, 3906, m=32, seed=synth.SEED + 21)
sw = StepWindow(ctx, w, rank=0, world=1)
first = None
for spec in settings:
    apply(spec, True)
    sw.step()
    runs = []
    for _ in range(7):
        T, info, x = sw.step()
        runs.append(T["solve"])
    runs.sort()
    apply(spec, False)
    if first is None:
        first = x
    print("%-16s [%-28s] iters %3d term %d cost %.12e  solve %.4f ms (%.4f ms/iter)  crc %08x  |dx| %.2e" % (
        "odometry step", spec, info["iters"], info["term"], float(np.ravel(info["cost"])[-1]), runs[3] * 1e3, runs[3] * 1e3 / max(1, info["iters"]), zlib.crc32(x.tobytes()),
        float(np.abs(x - first).max())), flush=True)
