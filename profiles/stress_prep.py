"""Randomised stress of the per-point / per-surfel stages around the hot path against the oracle: the pre-filter (random extrinsic,
ranges, blind box; lidar_odometry.cc:489-496), UndistortSweep (random IMU tables: rates, epochs, perturbed poses; :143-158) and
UpdateSurfelPoses (:160-170) on random surfel sets, twice in a row (world -> body conversion happens once).
python profiles/stress_prep.py [seconds]"""
import os, sys, time
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R_ + "/wildcat-slam_amd/python", R_ + "/oracle"]
import numpy as np
import pyoracle
from wildcat_slam_amd import lib, synth, records as R

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
ctx = lib.Context(0)
t_end = time.time() + budget
n = bad = 0
seed = 0
def note(*a):
    global bad
    bad += 1
    print("MISMATCH", *a)
while time.time() < t_end:
    seed += 1
    rng = np.random.default_rng(12_000 + seed)
    t0 = float(rng.choice([0.0, 1000.0, 1.6e9, 1.7e9 + rng.uniform(0, 1e6)]))
    npts = int(10 ** rng.uniform(1, 5.3))
    pts = synth.g1_room(npts, seed=int(rng.integers(1, 1 << 30)), t_start=t0)
    # ---- pre-filter ----
    ax = rng.normal(size=3); ax /= np.linalg.norm(ax); ang = rng.uniform(0, np.pi)
    q = np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * ax])
    tvec = rng.uniform(-0.5, 0.5, 3)
    lo, hi = np.sort(rng.uniform(-2, 2, (2, 3)), axis=0)
    args = (q, tvec, float(rng.uniform(0.1, 3.0)), float(rng.uniform(20, 150)), lo, hi)
    ref = pyoracle.prefilter_points(pts, *args)
    got = ctx.prefilter_points(pts, *args)
    if len(got) != len(ref) or not (np.array_equal(got["time"], ref["time"]) and all(np.array_equal(got[f], ref[f]) for f in ("x", "y", "z"))):
        note("prefilter seed", seed, len(pts), len(got), len(ref))
    # ---- undistort ----
    rate = float(rng.choice([100.0, 200.0, 400.0]))
    ep, er = float(10 ** rng.uniform(-3, -1)), float(10 ** rng.uniform(-4, -2))
    ph = rng.random(6) * 2 * np.pi
    def perturb(tr):
        return (ep * np.stack([np.sin(0.9 * tr + ph[0]), np.sin(1.3 * tr + ph[1]), np.sin(0.7 * tr + ph[2])], -1),
                er * np.stack([np.sin(1.1 * tr + ph[3]), np.sin(0.8 * tr + ph[4]), np.sin(1.7 * tr + ph[5])], -1))
    imu, _ = synth.imu_states(t0 - 0.0031, t0 + 0.51, rate=rate, t_origin=t0, perturb=perturb)
    rc, uref = pyoracle.undistort_sweep(pts, imu)
    if rc == 0:
        ug = ctx.undistort_sweep(pts, imu)
        ok = np.array_equal(ug["time"], uref["time"])
        for f in ("x", "y", "z"):
            d = np.abs(ug[f].astype(np.float64) - uref[f].astype(np.float64))
            ok = ok and d.max() <= np.spacing(np.abs(uref[f]).max().astype(np.float32)) and (d > 0).mean() < 2e-3
        xyz, tt = ctx.undistort_sweep_packed(pts, imu)
        ok = ok and np.array_equal(tt, ug["time"]) and np.array_equal(xyz[:, 0], ug["x"]) and np.array_equal(xyz[:, 2], ug["z"])
        if not ok:
            note("undistort seed", seed, len(pts), rate)
    # ---- surfel poses ----
    ns = int(10 ** rng.uniform(0.5, 4.5))
    S = np.zeros(ns, R.SURFEL)
    S["t"] = np.sort(rng.uniform(t0 + 0.001, t0 + 0.5, ns))
    S["center"] = rng.uniform(-30, 30, (ns, 3))
    nr = rng.normal(size=(ns, 3)); S["normal"] = nr / np.linalg.norm(nr, axis=1, keepdims=True)
    a = rng.normal(size=(ns, 3, 3)); S["cov"] = (a @ a.transpose(0, 2, 1)).reshape(ns, 9) * 1e-3
    P = np.zeros(ns, R.POSE); B = np.zeros(ns, np.uint8)
    B[rng.random(ns) < 0.3] = 1  # some are in the body frame already
    Sr, Pr, Br = S.copy(), P.copy(), B.copy()
    d_imu, d_s, d_p, d_b = ctx.to_device(imu), ctx.to_device(S), ctx.to_device(P), ctx.to_device(B)
    for rep in range(2):
        pyoracle.update_surfel_poses(imu, Sr, Pr, Br)
        ctx.update_surfel_poses(d_imu, len(imu), d_s, d_p, d_b, ns)
        Sg, Pg, Bg = d_s.download(R.SURFEL, ns), d_p.download(R.POSE, ns), d_b.download(np.uint8, ns)
        err = max(np.abs(Sg[f] - Sr[f]).max() / max(1.0, np.abs(Sr[f]).max()) for f in ("center", "normal", "cov"))
        perr = max(np.abs(Pg["pos"] - Pr["pos"]).max() / max(1.0, np.abs(Pr["pos"]).max()), np.abs(Pg["quat"] - Pr["quat"]).max())
        if not (np.array_equal(Bg, Br) and err <= 1e-11 and perr <= 1e-11):
            note("poses seed", seed, ns, "rep", rep, err, perr)
    n += 1
print("rounds %d, mismatches %d, last seed %d" % (n, bad, seed))
