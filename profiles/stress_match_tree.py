"""Randomised stress of the matcher's kd-tree (csrc/match_tree.inc) against the CPU oracle's kd-tree: target sets of 1 ... 300 k
surfels drawn from different shapes - uniform boxes of very different extents, tight clusters, surfaces with coherent normals,
coincident copies, a line, huge and tiny coordinates -, queried by themselves and by another set; the k-NN tables (indices and
distances) must be identical, bit for bit.  Run on the GPU box: python profiles/stress_match_tree.py [seconds]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wildcat-slam_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle  # noqa: E402
from wildcat_slam_amd import lib, records as R  # noqa: E402

AS = 5.0 * np.pi / 180.0


def feat(s):
    return np.concatenate([s["center"], s["normal"] / AS], 1)


def unit(v):
    return v / np.linalg.norm(v, axis=1, keepdims=True)


def make(rng, n, shape):
    s = np.zeros(n, R.SURFEL)
    if shape == "uniform":
        ext = 10.0 ** rng.uniform(-2, 3)
        c, nr = rng.uniform(-ext, ext, (n, 3)), unit(rng.normal(size=(n, 3)))
    elif shape == "clusters":
        k = max(1, n // int(rng.integers(2, 40)))
        cc, cn = rng.uniform(-30, 30, (k, 3)), unit(rng.normal(size=(k, 3)))
        m = rng.integers(0, k, n)
        c, nr = cc[m] + 0.002 * rng.normal(size=(n, 3)), unit(cn[m] + 0.002 * rng.normal(size=(n, 3)))
    elif shape == "walls":
        axis = rng.integers(0, 3, n)
        c = rng.uniform(-20, 20, (n, 3))
        c[np.arange(n), axis] = rng.choice([-20.0, 20.0], n) + 0.01 * rng.normal(size=n)
        nr = np.zeros((n, 3))
        nr[np.arange(n), axis] = 1.0
        nr = unit(nr + 0.01 * rng.normal(size=(n, 3)))
    elif shape == "copies":
        k = max(1, n // 50)
        base_c, base_n = rng.uniform(-5, 5, (k, 3)), unit(rng.normal(size=(k, 3)))
        m = rng.integers(0, k, n)
        c, nr = base_c[m], base_n[m]  # exact duplicates: every distance ties, the index decides
    elif shape == "line":
        t = rng.uniform(-50, 50, n)
        c = np.stack([t, 0.3 * t, -0.1 * t], 1)
        nr = unit(np.tile([[0.0, 0.0, 1.0]], (n, 1)) + 1e-4 * rng.normal(size=(n, 3)))
    else:  # "far": large offsets, small spread (fp32 cannot tell the members apart)
        c = np.array([900.0, -700.0, 400.0]) + 1e-4 * rng.normal(size=(n, 3))
        nr = unit(np.tile([[0.6, 0.0, 0.8]], (n, 1)) + 1e-6 * rng.normal(size=(n, 3)))
    s["center"], s["normal"] = c, nr
    s["t"] = np.sort(rng.uniform(0, 5, n))
    p = np.zeros(n, R.POSE)
    p["quat"][:, 0] = 1.0
    return s, p


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.default_rng(20260929)
    ctx = lib.Context(0)
    t_end, rounds = time.time() + budget, 0
    shapes = ("uniform", "clusters", "walls", "copies", "line", "far")
    while time.time() < t_end:
        shape = shapes[rounds % len(shapes)]
        nt = int(10 ** rng.uniform(0, 5.48))
        nq = int(10 ** rng.uniform(0, 3.7))
        t, tp = make(rng, nt, shape)
        q, qp = make(rng, nq, shapes[int(rng.integers(0, len(shapes)))])
        q["t"] += 10.0  # (the fixed-window surfel of a pair must be the older one, lidar_odometry.cc:301)
        _, idx, d2 = ctx.match(q, qp, t, tp, False, want_knn=True)
        ridx, rd2 = pyoracle.knn6(feat(t), feat(q), 10)
        assert np.array_equal(d2, rd2), (rounds, shape, nt, nq, "distances, other set")
        assert np.array_equal(idx.astype(np.int64), ridx.astype(np.int64)), (rounds, shape, nt, nq, "indices, other set")
        if nt <= 60000:
            _, idx, d2 = ctx.match(t, tp, t, tp, True, want_knn=True)
            ridx, rd2 = pyoracle.knn6(feat(t), feat(t), 10)
            assert np.array_equal(d2, rd2) and np.array_equal(idx.astype(np.int64), ridx.astype(np.int64)), (rounds, shape, nt, "same set")
        rounds += 1
        print("round %3d: %-8s %6d targets, %5d queries: identical" % (rounds, shape, nt, nq), flush=True)
    print("%d rounds agree with the oracle" % rounds)


if __name__ == "__main__":
    main()
