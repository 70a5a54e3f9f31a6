"""Randomised PARAMETER + window stress of the factors and the LM solve against the CPU oracle: windows of 2 - 10 sweeps and 20 - 1 500
patches with random pose errors, random loss / weights / sigma0 / quirks / gauge / IMU on-off; H, g, cost by value at a random point
(1e-9 / 1e-10), then the solve: iterations, accepted steps, termination equal; corrections 1e-5 / final cost 1e-7 on converged solves
of at most 30 iterations with the gauge held (the 1e-6 bar is the tests' on their nine configurations; random windows sit at 1e-9 ... 5e-6).  A solve whose
cost change in the oracle's LAST iteration lies within 1e-3 of the function tolerance may legitimately end an iteration earlier or
later on either side (Ceres' |dcost| <= 1e-6 cost test): counted apart as "borderline".  python profiles/stress_window.py [seconds]"""
import os, sys, time
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R_ + "/wildcat-slam_amd/python", R_ + "/oracle"]
import numpy as np
import pyoracle
from wildcat_slam_amd import lib, synth, records as R

def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
only = [int(a) for a in sys.argv[2].split(",")] if len(sys.argv) > 2 else None  # replay these seeds (with the dense LM step beside the default one)
ctx = lib.Context(0)
t_end = time.time() + budget
n = bad = borderline = n_strict = retries = n_free = 0
worst = 0.0
seed = 0
while time.time() < t_end:
    seed += 1
    if only is not None:
        if not only:
            break
        seed = only.pop(0)
    rng = np.random.default_rng(55_000 + seed)
    scans, patches = int(rng.integers(2, 11)), int(10 ** rng.uniform(1.3, 3.2))
    fixed = int(rng.choice([0, patches // 2, patches]))
    w = synth.surfel_window(scans, patches, seed=int(rng.integers(1, 1 << 30)), fixed_patches=fixed,
                            pose_err=(float(10 ** rng.uniform(-3, -1.3)), float(10 ** rng.uniform(-5, -2.5))))
    prm = pyoracle.default_params()
    prm.reference_quirks = int(rng.integers(0, 2))
    prm.cauchy_a = float(rng.choice([0.4, 0.4, 10 ** rng.uniform(-2, 1)]))
    prm.surfel_sigma0 = float(rng.choice([0.05 / 6, 0.05 / 6, 10 ** rng.uniform(-3, -1)]))
    for name in ("w_gyr", "w_acc", "w_bg", "w_ba"):
        setattr(prm, name, float(getattr(prm, name) * rng.choice([1.0, 1.0, 10 ** rng.uniform(-1, 1)])))
    fix_first = bool(rng.integers(0, 2))
    with_imu = bool(rng.random() < 0.8)
    if not with_imu:
        fix_first = True  # (without IMU factors the gauge must be held by the first state)
    pairs = pyoracle.match(w["surf"], w["pose"], w["surf"], w["pose"], True, prm)
    pf = pyoracle.match(w["surf"], w["pose"], w["fix_surf"], w["fix_pose"], False, prm) if fixed else np.zeros(0, R.PAIR)
    if len(pairs) < 10:
        continue
    W = pyoracle.Window(w["sample_times"], w["grav"], fix_first, prm)
    W.add_binary(w["surf"], w["pose"], pairs)
    if fixed and len(pf):
        W.add_unary(w["fix_surf"], w["fix_pose"], w["surf"], w["pose"], pf)
    if with_imu:
        W.add_imu(w["imu"])
    ctx.set_params(prm)
    d_s, d_p, d_pairs = ctx.to_device(w["surf"]), ctx.to_device(w["pose"]), ctx.to_device(pairs)
    use_fix = fixed and len(pf)
    ctx.window_build(d_s, d_p, d_pairs, len(pairs), w["imu"] if with_imu else None, w["sample_times"], w["grav"], fix_first,
                     ctx.to_device(w["fix_surf"]) if use_fix else None, ctx.to_device(w["fix_pose"]) if use_fix else None,
                     ctx.to_device(pf) if use_fix else None, len(pf) if use_fix else 0)
    n += 1
    what = []
    x = 1e-3 * rng.normal(size=12 * W.ns)
    H_ref, g_ref, c_ref = W.linearize(x)
    H, g, c = ctx.window_linearize(x)
    if not (abs(c - c_ref) <= 1e-10 * c_ref and rel(H, H_ref) <= 1e-9 and rel(g, g_ref) <= 1e-9 and np.array_equal(H, H.T)):
        what.append("linearize cost %.1e H %.1e g %.1e" % (abs(c - c_ref) / c_ref, rel(H, H_ref), rel(g, g_ref)))
    x0 = np.zeros(12 * W.ns)
    x_ref, s_ref, f_ref = W.solve(x0)
    xg, s, f_s = ctx.window_solve(x0)
    if only is not None:
        ctx.set_dev_option("lm_dense", 1)
        xd, sd, f_d = ctx.window_solve(x0)
        ctx.set_dev_option("lm_dense", 0)
        print("seed %d first step against the oracle's: default %.1e, dense %.1e" % (seed, rel(f_s, f_ref), rel(f_d, f_ref)))
        print("seed %d ns %d: oracle it %d acc %d term %d cost %.9e | default it %d acc %d term %d cost %.9e x %.1e | dense it %d acc %d term %d cost %.9e x %.1e" % (
            seed, W.ns, s_ref.iterations, s_ref.successful_steps, s_ref.termination, s_ref.final_cost, s.iterations, s.successful_steps, s.termination, s.final_cost,
            rel(xg, x_ref), sd.iterations, sd.successful_steps, sd.termination, sd.final_cost, rel(xd, x_ref)))
    same = s.termination == s_ref.termination and s.iterations == s_ref.iterations and s.successful_steps == s_ref.successful_steps
    gauge_held = bool(fix_first or use_fix)
    if not gauge_held:
        # no first state held and no fixed window: global translation and yaw are only held by the damping (never the reference's
        # situation: lidar_odometry.cc:556-560 fixes the first sample until the fixed window exists).  The normal equations are
        # singular up to the damping, and which steps get rejected is rounding - the oracle, round 2's dense path and the default path
        # each count differently.  Only the final cost is compared (1e-3).
        n_free += 1
        dcost = abs(s.final_cost - s_ref.final_cost) / max(s_ref.final_cost, 1e-300)
        if s.termination == 0 and s_ref.termination == 0 and not dcost <= 1e-3:
            what.append("gauge-free final cost %.1e" % dcost)
    elif not same:
        # borderline: replay the oracle's costs? - cheap proxy: the two final costs agree to 1e-5 and the iteration counts differ by one
        if abs(s.iterations - s_ref.iterations) <= 1 and abs(s.final_cost - s_ref.final_cost) <= 1e-5 * s_ref.final_cost:
            borderline += 1
        else:
            what.append("solve iterations %d / %d, accepted %d / %d, termination %d / %d, cost %.6e / %.6e" % (
                s.iterations, s_ref.iterations, s.successful_steps, s_ref.successful_steps, s.termination, s_ref.termination, s.final_cost, s_ref.final_cost))
    else:
        # the corrections are held to 1e-6 where they are determined: the gauge held (first position fixed, or a fixed window), the solve
        # converged, and not after dozens of iterations (a run of 60 - 100 iterations amplifies the last bits of every step; the oracle
        # against round 2's dense path differs by 1e-5 ... 1e-3 there too).  Otherwise the final cost alone (1e-6) is compared.
        strict = s_ref.termination == 0 and s_ref.iterations <= 30
        dcost = abs(s.final_cost - s_ref.final_cost) / max(s_ref.final_cost, 1e-300)
        n_strict += bool(strict)
        if strict:
            worst = max(worst, rel(xg, x_ref))
        if strict and not (dcost <= 1e-7 and rel(xg, x_ref) <= 1e-5):
            what.append("solve values: cost %.1e x %.1e" % (dcost, rel(xg, x_ref)))
        if not strict and not dcost <= 1e-6:
            what.append("final cost %.1e (x %.1e, not compared)" % (dcost, rel(xg, x_ref)))
    retries += int(s.first_step[1])
    if what:
        bad += 1
        print("MISMATCH seed", seed, "scans", scans, "patches", patches, "fixed", fixed, "ns", W.ns, "pairs", len(pairs), len(pf), "quirks", prm.reference_quirks,
              "imu", with_imu, "fix_first", fix_first, "cauchy", prm.cauchy_a, "sigma0", prm.surfel_sigma0, "|", "; ".join(what))
print("windows %d (%d gauge-free: cost only; %d with corrections compared: worst %.1e), mismatches %d, borderline solves (one iteration apart at the function tolerance) %d, steps re-formed by the dense factorisation %d, last seed %d" % (n, n_free, n_strict, worst, bad, borderline, retries, seed))
