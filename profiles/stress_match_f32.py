"""Random windows through both precisions of k_knn_gate's first look and both orders of the candidate halves (WC_KNN_F32 /
WC_KNN_ORDER are read per call), against the CPU oracle: k-NN tables (indices and distances) and pair lists must be identical.
Extents from 2 m to 180 m, offsets up to 90 m from the origin, clustered and uniform centres, random and coherent normals.
python profiles/stress_match_f32.py [cases]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wildcat-slam_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle  # noqa: E402
from wildcat_slam_amd import lib  # noqa: E402
from wildcat_slam_amd import records as R  # noqa: E402

ctx = lib.Context(0)
rng = np.random.default_rng(31337)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24


def surfels(n, extent, offset, clustered, coherent, t0):
    s = np.zeros(n, R.SURFEL)
    if clustered:
        nc = max(4, n // 40)
        c = rng.uniform(-extent / 2, extent / 2, size=(nc, 3))
        s["center"] = c[rng.integers(0, nc, n)] + rng.normal(scale=0.02, size=(n, 3)) + offset
    else:
        s["center"] = rng.uniform(-extent / 2, extent / 2, size=(n, 3)) + offset
    if coherent:
        base = np.eye(3)[rng.integers(0, 3, n)] * rng.choice([-1.0, 1.0], size=(n, 1))
        nrm = base + rng.normal(scale=0.02, size=(n, 3))
    else:
        nrm = rng.normal(size=(n, 3))
    s["normal"] = nrm / np.linalg.norm(nrm, axis=1, keepdims=True)
    s["t"] = t0 + np.sort(rng.uniform(0, 5, size=n))
    p = np.zeros(n, R.POSE)
    p["quat"][:, 0] = 1.0
    return s, p


def feat(s):
    return np.concatenate([s["center"] / 1.0, s["normal"] / (5.0 * np.pi / 180.0)], 1)


for case in range(cases):
    extent = float(rng.choice([2.0, 8.0, 30.0, 180.0]))
    offset = rng.uniform(-90, 90, size=3) * (case % 3 != 0)
    nt, nq = int(rng.integers(200, 30000)), int(rng.integers(200, 20000))
    clustered, coherent = bool(case & 1), bool(case & 2)
    t, tp = surfels(nt, extent, offset, clustered, coherent, 0.0)
    q, qp = surfels(nq, extent * 1.2, offset, clustered, coherent, 10.0)
    ridx, rd2 = pyoracle.knn6(feat(t), feat(q), 10)
    rpairs = pyoracle.match(q, qp, t, tp, False)
    rself = pyoracle.match(t, tp, t, tp, True)
    for f32 in ("0", "1"):
        for order in ("centre", "normal"):
            os.environ["WC_KNN_F32"], os.environ["WC_KNN_ORDER"] = f32, order
            pairs, idx, d2 = ctx.match(q, qp, t, tp, False, want_knn=True)
            assert np.array_equal(idx.astype(np.int64), ridx.astype(np.int64)) and np.array_equal(d2, rd2), (case, f32, order, "knn")
            assert np.array_equal(pairs, rpairs), (case, f32, order, "pairs")
            assert np.array_equal(ctx.match(t, tp, t, tp, True), rself), (case, f32, order, "self")
    print(f"case {case:2d}: extent {extent:6.1f} m, offset {np.abs(offset).max():5.1f} m, {nq:6d} queries / {nt:6d} targets, clustered={clustered}, coherent={coherent}: "
          f"{len(rpairs)} + {len(rself)} pairs identical in all four variants")
print("all cases agree with the oracle")
