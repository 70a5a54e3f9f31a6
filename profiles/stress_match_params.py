"""Randomised PARAMETER + window stress of the matcher's PAIRS (exact k-NN + gates + the order-dependent de-duplication,
knn_surfel_matcher.cc:16-89) against the CPU oracle: windows of re-observed patches with perturbed poses (synth.surfel_window) of
random size, random k / scales / gates, both walks of the tree pinned (development option knn_group), two repetitions each; the pair
lists must be the oracle's byte for byte.  python profiles/stress_match_params.py [seconds]"""
import os, sys, time
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R_ + "/wildcat-slam_amd/python", R_ + "/oracle"]
import numpy as np
import pyoracle
from wildcat_slam_amd import lib, synth

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
ctx = lib.Context(0)
t_end = time.time() + budget
rounds = bad = 0
seed = 0
while time.time() < t_end:
    seed += 1
    rng = np.random.default_rng(99_000 + seed)
    scans = int(rng.integers(2, 9))
    patches = int(10 ** rng.uniform(0.5, 3.6))
    w = synth.surfel_window(scans, patches, seed=int(rng.integers(1, 1 << 30)), fixed_patches=int(rng.choice([0, patches, 3 * patches])),
                            pose_err=(float(10 ** rng.uniform(-3, -1)), float(10 ** rng.uniform(-5, -2))))
    prm = pyoracle.default_params()
    prm.knn_k = int(rng.choice([1, 2, 3, 5, 8, 10, 10, 10, 12, 16]))
    prm.center_scale = float(rng.choice([1.0, 1.0, 10 ** rng.uniform(-1, 1)]))
    prm.angular_scale = float(rng.choice([5.0, 5.0, rng.uniform(1.0, 30.0)]) * np.pi / 180.0)
    prm.surfel_dist_max = float(rng.choice([0.1, 0.1, 10 ** rng.uniform(-2.5, 0)]))
    prm.time_diff_min = float(rng.choice([0.06, 0.06, rng.uniform(0.0, 1.0)]))
    ctx.set_params(prm)
    ref_b = pyoracle.match(w["surf"], w["pose"], w["surf"], w["pose"], True, prm)
    fix_s, fix_p = w.get("fix_surf"), w.get("fix_pose")
    ref_u = pyoracle.match(w["surf"], w["pose"], fix_s, fix_p, False, prm) if fix_s is not None and len(fix_s) else None
    for group in (0, 1):
        ctx.set_dev_option("knn_group", group)
        for rep in range(2):
            got_b = ctx.match(w["surf"], w["pose"], w["surf"], w["pose"], True)
            ok = got_b.tobytes() == ref_b.tobytes()
            if ref_u is not None:
                got_u = ctx.match(w["surf"], w["pose"], fix_s, fix_p, False)
                ok = ok and got_u.tobytes() == ref_u.tobytes()
            rounds += 1
            if not ok:
                bad += 1
                print("MISMATCH seed", seed, "group", group, "rep", rep, "surfels", len(w["surf"]), "fixed", 0 if fix_s is None else len(fix_s), "k", prm.knn_k, "cs", prm.center_scale,
                      "as", prm.angular_scale, "dmax", prm.surfel_dist_max, "tmin", prm.time_diff_min, "pairs", len(got_b), len(ref_b))
ctx.set_dev_option("knn_group", -1)
print("rounds %d, mismatches %d, last seed %d" % (rounds, bad, seed))
