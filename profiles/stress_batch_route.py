"""Randomised stress of the two multi-sweep / multi-rank forms of the extraction against the plain call:
  batch   wc_extract_surfels_batch_* on K = 2 ... 10 random sweeps (run-structured, firing order, fuzz clouds, short, empty), three
          rounds each: every sweep's surfels and ids byte for byte what K wc_extract_surfels calls give
  route   wc_extract_surfels_sharded + wc_gather_surfels on 2 ... 5 thread-ranks (dist.ThreadComm) over one random cloud: the merged
          list against the unsharded call - byte for byte in the exact arithmetic; ids as a set and geometry 1e-6 in the default one
python profiles/stress_batch_route.py [seconds]"""
import os, sys, time
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R_ + "/wildcat-slam_amd/python", R_ + "/oracle", R_ + "/tests"]
import numpy as np
import helpers
from wildcat_slam_amd import lib, synth, records as R
from test_fuzz_gpu import _cloud
from test_route_gpu import _run_ranks

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 200.0
t_end = time.time() + budget
gpu = lib.Context(0)
nb = nr = bad = soft = 0
seed = 0

def sweep(rng):
    k = int(rng.integers(0, 6))
    if k == 0:
        return synth.g2_lattice(int(rng.integers(4, 600)), m=int(rng.integers(21, 48)), seed=int(rng.integers(1, 1 << 30)))[0]
    if k == 1:
        return synth.g1_room(int(rng.integers(5_000, 200_000)), seed=int(rng.integers(1, 1 << 30)))
    if k == 2:
        return _cloud(int(rng.integers(0, 1 << 20)))
    if k == 3:
        return synth.g2_lattice(4, m=32, seed=3)[0][: int(rng.integers(0, 100))]
    if k == 4:
        p = synth.g2_lattice(int(rng.integers(30, 300)), m=int(rng.integers(25, 50)), seed=int(rng.integers(1, 1 << 30)))[0].copy()
        p["x"] += np.float32(rng.uniform(-0.4, 0.4)); p["z"] += np.float32(rng.uniform(-0.4, 0.4))
        return p
    return synth.g1_room(int(rng.integers(60_000, 120_000)), seed=int(rng.integers(1, 1 << 30)))

while time.time() < t_end:
    seed += 1
    rng = np.random.default_rng(41_000 + seed)
    if seed % 3:  # ---- batch ----
        K = int(rng.integers(2, 11))
        sweeps = [sweep(rng) for _ in range(K)]
        gpu.set_exact_sums(False)  # (resets the context's adaptive path choice: every reference call starts on the default path)
        single = [gpu.extract_surfels(p) if len(p) >= 1 else (np.zeros(0, R.SURFEL), np.zeros(0, R.SURFEL_ID)) for p in sweeps]
        ctx = lib.Context(0)
        try:
            jobs, keep = [], []
            for p in sweeps:
                n = len(p)
                cap = max(1024, (3 * n) // 20 + 1)
                d_p = ctx.to_device(p) if n else ctx.alloc(48)
                d_o, d_i = ctx.alloc(144 * cap), ctx.alloc(16 * cap)
                keep.append((d_p, d_o, d_i))
                t_lo, t_hi = (float(p["time"][0]), float(p["time"][-1])) if n else (1.0, 0.0)
                jobs.append((ctx.points_desc(d_p, n), d_o, d_i, cap, t_lo, t_hi))
            enq, fin = ctx.extract_batch_prepare(jobs)
            for rnd in range(3):
                enq()
                counts = fin()
                for k, (s_ref, id_ref) in enumerate(single):
                    ok = counts[k] == len(s_ref)
                    if ok and counts[k]:
                        s_b, id_b = keep[k][1].download(R.SURFEL, counts[k]), keep[k][2].download(R.SURFEL_ID, counts[k])
                        if not (s_b.tobytes() == s_ref.tobytes() and id_b.tobytes() == id_ref.tobytes()):
                            # not the same bytes: one of the two calls repeated the sweep on the exact path (a gate in the noise band, the
                            # adaptive back-off of a context after a fall-back) and the other did not - then ids as a set + geometry 1e-6
                            soft += 1
                            try:
                                ok = set(helpers.id_tuples(id_b)) == set(helpers.id_tuples(id_ref))
                                if ok:
                                    helpers.check_surfels(s_b, id_b, s_ref, id_ref, tol=1e-6, t_tol=1e-5)
                            except AssertionError:
                                ok = False
                    if not ok:
                        bad += 1
                        print("BATCH MISMATCH seed", seed, "K", K, "round", rnd, "sweep", k, "points", len(sweeps[k]), "surfels", counts[k], len(s_ref))
            nb += 1
        except Exception as e:
            bad += 1
            print("BATCH EXCEPTION seed", seed, repr(e)[:200])
        finally:
            ctx.close()
    else:  # ---- route ----
        world, exact = int(rng.integers(2, 6)), bool(rng.integers(0, 2))
        pts = sweep(rng)
        while len(pts) < 2000:
            pts = sweep(rng)
        gpu.set_exact_sums(exact)
        s_ref, id_ref = gpu.extract_surfels(pts)
        gpu.set_exact_sums(False)
        try:
            out = _run_ranks(world, pts, want_gather=True, exact=exact)
            for r in range(world):
                s_m, id_m = out[r][1]
                ok = s_m.tobytes() == s_ref.tobytes() and id_m.tobytes() == id_ref.tobytes()
                if not ok and not exact:
                    soft += 1  # a rank handed its share to the exact path: the two arithmetics meet in one list
                    ok = len(s_m) == len(s_ref) and set(helpers.id_tuples(id_m)) == set(helpers.id_tuples(id_ref))
                    if ok and len(s_ref):
                        try:
                            helpers.check_surfels(s_m, id_m, s_ref, id_ref, tol=1e-6, t_tol=1e-5)
                        except AssertionError:
                            ok = False
                if not ok:
                    bad += 1
                    print("ROUTE MISMATCH seed", seed, "world", world, "exact", exact, "rank", r, "points", len(pts), "surfels", len(s_m), len(s_ref))
            nr += 1
        except Exception as e:
            bad += 1
            print("ROUTE EXCEPTION seed", seed, "world", world, "exact", exact, repr(e)[:300])
print("batches %d, routed clouds %d, mismatches %d (comparisons by ids + 1e-6 instead of bytes, the two calls on different arithmetics: %d), last seed %d" % (nb, nr, bad, soft, seed))
