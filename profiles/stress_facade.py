"""Randomised stress of the drop-in class against the orchestrated oracle, sweep by sweep with re-synchronisation (the comparison of
tests/test_facade_gpu.py::test_facade_each_sweep_against_the_oracle_resynchronised) on streams the test does not run: random point
rate, duration, stream epoch (up to 1.6e9 s: ROS stamps), gyro bias, range noise, beams, message length, seed, quirks, arithmetic.
python profiles/stress_facade.py [seconds]"""
import os, sys, time
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R_ + "/wildcat-slam_amd/python", R_ + "/oracle", R_ + "/tests"]
import numpy as np
import pyoracle
from wildcat_slam_amd import lib, synth
from test_facade_gpu import _feed, _state_diff, _pair_set_difference

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 200.0
only = [int(a) for a in sys.argv[2].split(",")] if len(sys.argv) > 2 else None  # replay these seeds
dense = len(sys.argv) > 3 and sys.argv[3] == "dense"  # ... with round 2's dense LM step (development option lm_dense)
t_end = time.time() + budget
streams = sweeps = bad = hard = 0
worst = 0.0
seed = 0
while time.time() < t_end:
    seed += 1
    if only is not None:
        if not only:
            break
        seed = only.pop(0)
    rng = np.random.default_rng(31_000 + seed)
    dur = float(rng.uniform(1.7, 8.4))
    pps = int(rng.choice([60_000, 100_000, 150_000, 250_000, 400_000]))
    t0 = float(rng.choice([0.0, 1000.0, 1.6e9, 1.7e9 + rng.uniform(0, 1e6)]))
    exact, quirks = int(rng.integers(0, 2)), int(rng.integers(0, 2))
    kw = dict(pts_per_s=pps, seed=int(rng.integers(1, 1 << 30)), t_start=t0, gyro_bias=tuple(rng.uniform(-0.03, 0.03, 3)),
              range_noise=float(rng.choice([0.01, 0.002, 0.03])), beams=int(rng.choice([16, 32, 64])), msg_dt=float(rng.choice([0.1, 0.05, 0.125])))
    try:
        msgs, imu, _ = synth.raw_stream(dur, **kw)
    except Exception as e:
        print("generator refused", kw, repr(e)[:100])
        continue
    odo, ref = lib.Odometry(0), pyoracle.Odometry()
    if dense:
        odo.set_dev_option("lm_dense", 1)
    if len(sys.argv) > 3 and sys.argv[3].startswith("radius"):
        odo.set_dev_option("lm_dense_radius", int(sys.argv[3][6:]))
    odo.set_exact_sums(exact); odo.set_quirks(quirks); ref.set_quirks(quirks); odo.set_keep_pair_stamps(True)
    notes = []

    def on_sweep(k):
        global sweeps, worst
        a, b = odo.samples(), ref.samples()
        sa, sb = odo.stats(), ref.stats()
        sweeps += 1
        if a.shape != b.shape or not np.array_equal(a[:, 0], b[:, 0]):
            notes.append((k, "sample states differ in number / stamps"))
            return
        for key in ("sld_surfels", "fix_surfels", "lm_iters", "termination"):
            if sa[key] != sb[key]:
                notes.append((k, key, sa[key], sb[key]))
        for key in ("binary", "unary"):
            if (sa[key] != sb[key]) if exact else abs(sa[key] - sb[key]) > 2:
                notes.append((k, key, sa[key], sb[key]))
        for which in (0, 1):
            only_f, only_o = _pair_set_difference(odo.pair_stamps(which), ref.pair_stamps(which), 0.0 if exact else 1e-5)
            if only_f + only_o > (0 if exact else 4):
                notes.append((k, "pairs", which, only_f, only_o))
        d = _state_diff(a, b)
        worst = max(worst, d)
        if not d <= 1e-6:
            notes.append((k, "states", float("%.2g" % d), "iters", sa["lm_iters"]))
        odo.import_state(*ref.export_state())

    try:
        _feed(odo, ref, msgs, imu, on_sweep)
    except AssertionError as e:
        notes.append(("assert", repr(e)[:160]))
    except Exception as e:
        notes.append(("exception", repr(e)[:160]))
    fast, ex = odo.extract_paths()
    streams += 1
    if notes:
        bad += 1
        # the gate of the driver's short run (tests/test_stress_gpu.py): anything but a drift of the sample states below 5e-5 in the
        # DEFAULT arithmetic (36 of 630 sweeps sit at 1e-6 ... 3.9e-5 there, on sparse streams or epoch-sized stamps)
        if any(not (len(nt) >= 3 and nt[1] == "states" and not exact and nt[2] <= 5e-5) for nt in notes):
            hard += 1
        print("MISMATCH seed", seed, "dur %.1f" % dur, kw, "exact", exact, "quirks", quirks, "fast/exact sweeps", fast, ex, "|", notes[:6])
    odo.close(); ref.close()
print("streams %d, sweeps compared %d, worst sample-state difference %.2g, streams with a note %d (other than default-arithmetic drift below 5e-5: %d), last seed %d" % (streams, sweeps, worst, bad, hard, seed))
