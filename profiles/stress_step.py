"""Randomised stress of the whole odometry step (wildcat_slam_amd/step.py: routed extraction + gather, sharded matcher, sharded window,
solve) for 2 ... 4 thread-ranks against the one-rank step AND against the orchestrated oracle on random scan sequences (3 - 8 sweeps of
100 - 1 500 roots, 21 - 40 points per patch): the ranks bitwise equal among themselves; counts, iterations, termination of the one-rank
step, corrections 1e-6; the one-rank step's surfel counts, pair counts, iterations and termination are the oracle's.
python profiles/stress_step.py [seconds]"""
import os, sys, time, threading
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R_ + "/wildcat-slam_amd/python", R_ + "/oracle"]
import numpy as np
import pyoracle
from wildcat_slam_amd import lib, synth
from wildcat_slam_amd import dist as wdist
from wildcat_slam_amd.step import StepWindow

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 200.0
one = lib.Context(0)
t_end = time.time() + budget
n = bad = 0
seed = 0
while time.time() < t_end:
    seed += 1
    rng = np.random.default_rng(123_000 + seed)
    scans, roots, m = int(rng.integers(3, 9)), int(10 ** rng.uniform(2.0, 3.2)), int(rng.integers(21, 41))
    w = synth.g2_scan_sequence(scans, roots, m=m, seed=int(rng.integers(1, 1 << 30)))
    what = []
    try:
        _, info1, x1 = StepWindow(one, w).step()
        ref = pyoracle.odometry_step(w)
        rs = ref["summary"]
        if (info1["new_surfels"], info1["sld"], info1["fix"], info1["binary"], info1["unary"], info1["iters"], info1["term"]) != (
                ref["new"], len(ref["sld_surf"]), ref["n_fix"], len(ref["pairs_sld"]), len(ref["pairs_fix"]), rs.iterations, rs.termination):
            what.append("one rank against the oracle: %s / %s" % ((info1["new_surfels"], info1["sld"], info1["fix"], info1["binary"], info1["unary"], info1["iters"], info1["term"]),
                                                                   (ref["new"], len(ref["sld_surf"]), ref["n_fix"], len(ref["pairs_sld"]), len(ref["pairs_fix"]), rs.iterations, rs.termination)))
        world = int(rng.integers(2, 5))
        ctxs = [lib.Context(0) for _ in range(world)]
        shared = wdist.ThreadComm.shared(world)
        res, errors = [None] * world, []

        def run(r):
            try:
                c = ctxs[r]
                c.set_comm(wdist.ThreadComm(shared, r, c))
                res[r] = StepWindow(c, w, rank=r, world=world).step()
            except Exception as e:
                errors.append(repr(e)[:200])
                shared["bar"].abort()

        th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=300)
        if errors:
            what.append("errors %s" % errors[:2])
        else:
            for r in range(world):
                _, info, x = res[r]
                if not np.array_equal(x, res[0][2]):
                    what.append("rank %d diverged" % r)
                for key in ("new_surfels", "sld", "fix", "binary", "unary", "iters", "term"):
                    if info[key] != info1[key]:
                        what.append("rank %d %s %s / %s" % (r, key, info[key], info1[key]))
            d = np.abs(res[0][2] - x1).max() / max(np.abs(x1).max(), 1e-300)
            if info1["term"] == 0 and info1["iters"] <= 30 and not d <= 1e-6:
                what.append("corrections %.1e" % d)
        for c in ctxs:
            c.close()
    except Exception as e:
        what.append("exception " + repr(e)[:200])
    n += 1
    if what:
        bad += 1
        print("MISMATCH seed", seed, "scans", scans, "roots", roots, "m", m, "|", "; ".join(what[:4]))
print("steps %d, mismatches %d, last seed %d" % (n, bad, seed))
